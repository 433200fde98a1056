"""
Host-side mirror of markovflow/ssm_gaussian_transformations.py: the six bijections between a
`StateSpaceModel`, its expectation parameters eta and its natural parameters theta.

The sequential work (block Cholesky, selected inverse, solves) runs in the HIP sweeps; what is left per time
step is independent d x d algebra on the selected-inverse blocks.
"""
import os

import torch

from . import linalg
from ._lib import FULL, SYM, VEC
from .packed import Plan
from .state_space_model import StateSpaceModel, _flat


def _T(x):
    return x.transpose(-1, -2)


def _chol_solve(L, B):
    return linalg.cholesky_solve(B, L)


def ssm_to_expectations(ssm: StateSpaceModel):
    """ssm_gaussian_transformations.py:32-89: eta_lin = mu, eta_diag = S + mu mu^T, eta_sub = A S + mu_+ mu^T."""
    pl = ssm.plan
    s = ssm._posterior_packed()["s"]
    mu = pl.unpack(VEC, s["x"])
    cov = pl.unpack(SYM, s["Sig"])
    sub = pl.unpack(FULL, s["Sub"], ssm.T - 1)
    eta_diag = cov + mu[..., :, None] * mu[..., None, :]
    eta_sub = sub + mu[:, 1:, :, None] * mu[:, :-1, None, :]
    return ssm._unflat(mu), ssm._unflat(eta_diag), ssm._unflat(eta_sub)


def expectations_to_ssm_params(eta_linear, eta_diag, eta_subdiag):
    """ssm_gaussian_transformations.py:93-178 (per-step independent algebra)."""
    m = eta_linear[..., None]
    cov = eta_diag - m @ _T(m)
    cov_sub = _T(eta_subdiag) - m[..., :-1, :, :] @ _T(m[..., 1:, :, :])
    chols = linalg.cholesky(cov)
    As = _T(_chol_solve(chols[..., :-1, :, :], cov_sub))
    offsets = (m[..., 1:, :, :] - As @ m[..., :-1, :, :])[..., 0]
    cond = cov[..., 1:, :, :] - As @ (cov[..., :-1, :, :] @ _T(As))
    return As, offsets, chols[..., 0, :, :], linalg.cholesky(cond), m[..., 0, :, 0]


def ssm_to_naturals(ssm: StateSpaceModel):
    """ssm_gaussian_transformations.py:182-253 (HIP kernel k_ssm_to_naturals)."""
    pk, pl = ssm.packed, ssm.plan
    nat = pl.ssm_to_naturals(pk.A, pk.off, pk.chol, precision=False)
    return (ssm._unflat(pl.unpack(VEC, nat["lin"])), ssm._unflat(pl.unpack(SYM, nat["diag"])),
            ssm._unflat(pl.unpack(FULL, nat["sub"], ssm.T - 1)))


def ssm_to_naturals_no_smoothing(ssm: StateSpaceModel):
    """ssm_gaussian_transformations.py:257-329 (per-step independent)."""
    chols = ssm.concatenated_cholesky_process_covariance
    theta_sub = _chol_solve(chols[..., 1:, :, :], ssm.state_transitions)
    theta_lin = _chol_solve(chols, ssm.concatenated_state_offsets[..., None])[..., 0]
    eye = torch.eye(ssm.d, dtype=chols.dtype, device=chols.device).expand(chols.shape)
    return theta_lin, -0.5 * _chol_solve(chols, eye), theta_sub


def _ssm_params_from_blocks(theta_linear, pd, ps, mu, cov, cov_sub):
    """Per-step algebra of naturals_to_ssm_params (ssm_gaussian_transformations.py:459-511) given the selected inverse."""
    # A_k = (S_kk^{-1} S_{k,k+1})^T
    As = _T(linalg.cholesky_solve(_T(cov_sub), linalg.cholesky(cov[:, :-1])))
    low = torch.tril(pd)
    pdsym = low + _T(torch.tril(pd, -1))
    cond_prec = pdsym.clone()
    cond_prec[:, :-1] += _T(As) @ ps      # Q_k^{-1} = P_kk + A_{k+1}^T P_{k+1,k}
    cond_prec = 0.5 * (cond_prec + _T(cond_prec))
    chols = linalg.cholesky(linalg.spd_inverse(cond_prec))
    # offsets: Q (A^{-T})^{-1} theta == mu_{k+1} - A_k mu_k (same quantity, no second sequential sweep)
    offsets = mu[:, 1:] - (As @ mu[:, :-1, :, None])[..., 0]
    return As, offsets, chols[:, 0], chols[:, 1:], mu[:, 0]


def naturals_to_ssm_params_packed(plan: Plan, lin, diag, sub):
    """theta (packed) -> StateSpaceModel sharing `plan`."""
    f = plan.factor(diag, sub, lin, aD=-2.0, aS=-1.0, aR=1.0, want_logdet=False)
    s = plan.selinv(f["L"], f["G"], f["y"], want_sub=True)
    T = plan.T
    if plan.d <= 8 and os.environ.get("VIDP_FUSED_NAT2SSM", "1") != "0":
        # the per-step algebra in one pass over the packed selected inverse (mfgm_packed_naturals_to_ssm); the natural-layout
        # parameter tensors are unpacked on first use
        from . import _lib
        from .packed import _ptr, _stream
        from ._lib import TRI
        A, off, chol = plan.empty(FULL), plan.empty(VEC), plan.empty(TRI)
        _lib.check(plan.lib.mfgm_packed_naturals_to_ssm(plan.h, _ptr(s["Sig"]), _ptr(s["Sub"]), _ptr(s["x"]), _ptr(diag), _ptr(sub), _ptr(A),
                                                        _ptr(off), _ptr(chol), _ptr(plan.info), _stream()), "mfgm_packed_naturals_to_ssm")
        plan.check_info()
        from .variational_cvi_sde import _ssm_from_packed
        ssm = _ssm_from_packed(plan, A, off, chol)
        ssm._post = dict(f=f, s=s)        # the marginals of this model are the ones it was built from
        return ssm
    plan.check_info()
    pd = -2.0 * plan.unpack(SYM, diag)
    ps = -plan.unpack(FULL, sub, T - 1)
    mu, cov, cov_sub = plan.unpack(VEC, s["x"]), plan.unpack(SYM, s["Sig"]), plan.unpack(FULL, s["Sub"], T - 1)
    As, off, cP0, cQ, mu0 = _ssm_params_from_blocks(None, pd, ps, mu, cov, cov_sub)
    ssm = StateSpaceModel(mu0, cP0, As, off, cQ, plan=plan)
    return ssm


def naturals_to_ssm_params(theta_linear, theta_diag, theta_subdiag, plan=None):
    """
    ssm_gaussian_transformations.py:333-511.  Returns (As, offsets, chol_P0, chol_Qs, mu0).
    `plan` (optional) fixes the time partition of the sweeps; `Plan(B, T, d, R0=T)` is the sequential elimination order
    of the reference ("numerics of the partitioned sweeps").
    """
    tl, bs = _flat(theta_linear, 2)
    td, _ = _flat(theta_diag, 3)
    ts, _ = _flat(theta_subdiag, 3)
    B, T, d = tl.shape
    plan = Plan(B, T, d, device=tl.device) if plan is None else plan
    ssm = naturals_to_ssm_params_packed(plan, plan.pack(VEC, tl), plan.pack(SYM, td), plan.pack(FULL, ts))
    return (ssm.state_transitions.reshape(bs + (T - 1, d, d)), ssm.state_offsets.reshape(bs + (T - 1, d)),
            ssm.cholesky_initial_covariance.reshape(bs + (d, d)), ssm.cholesky_process_covariances.reshape(bs + (T - 1, d, d)),
            ssm.initial_mean.reshape(bs + (d,)))


def naturals_to_ssm_params_no_smoothing(theta_linear, theta_diag, theta_subdiag):
    """ssm_gaussian_transformations.py:515-593 (per-step independent)."""
    c = linalg.cholesky(-2.0 * theta_diag)
    As = _chol_solve(c[..., 1:, :, :], theta_subdiag)
    off = _chol_solve(c, theta_linear[..., None])[..., 0]
    chols = linalg.cholesky(linalg.spd_inverse(chol=c))
    return As, off[..., 1:, :], chols[..., 0, :, :], chols[..., 1:, :, :], off[..., 0, :]
