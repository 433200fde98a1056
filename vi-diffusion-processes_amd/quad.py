"""
Thin wrappers over the tensor-product Gauss-Hermite kernels (csrc/mfgm_quad.h, include/mfgm.h `mfgm_quad_*`): the local CVI-DP / VDP
quantities of drifts that couple the state dimensions (Van der Pol), have no polynomial form (the ReLU network drift) or come with a
full diffusion matrix -- the reference's own quadrature formulation (markovflow/sde/sde.py:92-131, 359-518; sde_utils.py:119-359,
473-547; vi_sde.py:205-287, 422-470), natural-layout device tensors, state dimension <= 3.
"""
import ctypes

import torch

from . import _lib
from .packed import _ptr, _stream


def _info(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def _check_info(info, what):
    if int(info.item()) != 0:
        raise ArithmeticError(f"{what}: a marginal or conditional covariance is not positive definite")


def n_params(prm):
    return 3 * prm.nh + 1 if prm.kind == 11 else 2


def linearize(prm, mean, cov, check=True):
    """A [N.., d, d] = I + dt E_q[df/dx] and b [N.., d] = dt (E_q f - E_q[df/dx] m) on N(mean, cov) (10-point rule per dimension),
    clipped when prm.clip_lo < prm.clip_hi."""
    d = mean.shape[-1]
    m, c = mean.contiguous(), cov.contiguous()
    N = m.numel() // d
    A = torch.empty_like(c)
    b = torch.empty_like(m)
    info = _info(m.device)
    _lib.check(_lib.load().mfgm_quad_linearize(ctypes.byref(prm), N, _ptr(m), _ptr(c), _ptr(A), _ptr(b), _ptr(info), _stream()),
               "mfgm_quad_linearize")
    if check:
        _check_info(info, "quad.linearize")
    return A, b


def kl(prm, mu, Sig, Sub, grad=False, param_grad=False, check=True):
    """KL[q || p_SDE] per chain [B] along the Gaussian path mu [B,T,d], Sig [B,T,d,d], Sub [B,T-1,d,d] = Cov(x_{t+1}, x_t); with `grad`
    also (g1, gd, gs) = d KL / d (eta_lin, eta_diag, eta_sub), with `param_grad` d KL / d theta [B, np]."""
    lib = _lib.load()
    B, T, d = mu.shape
    mu, Sig, Sub = mu.contiguous(), Sig.contiguous(), Sub.contiguous()
    dev = mu.device
    out = torch.empty(B, dtype=torch.float64, device=dev)
    scratch = torch.empty(int(lib.mfgm_quad_kl_scratch_doubles(B, T, d, prm.nh)), dtype=torch.float64, device=dev)
    g1 = gd = gs = gth = None
    if grad or param_grad:
        g1, gd, gs = torch.empty_like(mu), torch.empty_like(Sig), torch.empty_like(Sub)
        if param_grad:
            gth = torch.empty((B, n_params(prm)), dtype=torch.float64, device=dev)
    info = _info(dev)
    _lib.check(lib.mfgm_quad_kl(ctypes.byref(prm), B, T, _ptr(mu), _ptr(Sig), _ptr(Sub), _ptr(out), _ptr(g1), _ptr(gd), _ptr(gs), _ptr(gth),
                                _ptr(scratch), _ptr(info), _stream()), "mfgm_quad_kl")
    if check:
        _check_info(info, "quad.kl")
    res = [out]
    if grad:
        res.append((g1, gd, gs))
    if param_grad:
        res.append(gth)
    return res[0] if len(res) == 1 else tuple(res)


def esde(prm, mean, cov, A, b, grads=True, param_grad=False, check=True):
    """E [N..] = 1/2 E_{N(mean, cov)} |f(x) + A x - b|^2_{q^-1} per node (no Riemann factor dt) and, with `grads`, its gradients with
    respect to (m, S, A, b); with `param_grad` the per-node gradient with respect to the drift parameters [N.., np]."""
    d = mean.shape[-1]
    m, c, A, b = mean.contiguous(), cov.contiguous(), A.contiguous(), b.contiguous()
    N = m.numel() // d
    E = torch.empty(m.shape[:-1], dtype=torch.float64, device=m.device)
    dm = dS = dA = db = gth = None
    if grads:
        dm, dS, dA, db = torch.empty_like(m), torch.empty_like(c), torch.empty_like(A), torch.empty_like(b)
    if param_grad:
        gth = torch.empty(m.shape[:-1] + (n_params(prm),), dtype=torch.float64, device=m.device)
    info = _info(m.device)
    _lib.check(_lib.load().mfgm_quad_esde(ctypes.byref(prm), N, _ptr(m), _ptr(c), _ptr(A), _ptr(b), _ptr(E), _ptr(dm), _ptr(dS), _ptr(dA),
                                          _ptr(db), _ptr(gth), _ptr(info), _stream()), "mfgm_quad_esde")
    if check:
        _check_info(info, "quad.esde")
    return E, (dm, dS, dA, db), gth


def vdp_lagrange(A, dEdm, dEdS, dobsm, dobsS, dt, clip=0.0):
    """(psi [B, N, d, d], lam [B, N, d]) of the VDP Lagrange sweep with jump conditions (vi_sde.py:289-347)."""
    B, N, d = dEdm.shape
    A, dEdm, dEdS, dobsm, dobsS = (x.contiguous() for x in (A, dEdm, dEdS, dobsm, dobsS))
    psi, lam = torch.empty_like(dEdS), torch.empty_like(dEdm)
    _lib.check(_lib.load().mfgm_quad_vdp_lagrange(B, N, d, float(dt), float(clip), _ptr(A), _ptr(dEdm), _ptr(dEdS), _ptr(dobsm), _ptr(dobsS),
                                                  _ptr(psi), _ptr(lam), _stream()), "mfgm_quad_vdp_lagrange")
    return psi, lam
