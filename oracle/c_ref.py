"""
ctypes binding of oracle/csrc/libbtdref.so, the plain-C CPU port used as `cpu_baseline` (kind "port") and as
an independent cross-check of the NumPy oracle.  Test infrastructure only (see oracle/__init__.py).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "csrc", "libbtdref.so")
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(LIB)
        lib.ref_btd_cholesky.restype = ctypes.c_int
        lib.ref_btd_cholesky.argtypes = [_dp, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int]
        lib.ref_btd_solve.restype = None
        lib.ref_btd_solve.argtypes = [_dp, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.ref_btd_logdet.restype = ctypes.c_double
        lib.ref_btd_logdet.argtypes = [_dp, ctypes.c_int, ctypes.c_int]
        lib.ref_btd_inverse_blocks.restype = None
        lib.ref_btd_inverse_blocks.argtypes = [_dp, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int]
        lib.ref_ssm_to_naturals.restype = None
        lib.ref_ssm_to_naturals.argtypes = [_dp] * 6 + [ctypes.c_int, ctypes.c_int]
        lib.ref_cvi_step_work_doubles.restype = ctypes.c_size_t
        lib.ref_cvi_step_work_doubles.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.ref_cvi_step.restype = ctypes.c_double
        lib.ref_cvi_step.argtypes = ([ctypes.c_int] * 4 + [_ip, _dp, _dp, ctypes.c_double] + [_dp] * 10
                                     + [ctypes.c_double, ctypes.c_double, _dp, _dp])
        lib.ref_sde_kl.restype = ctypes.c_double
        lib.ref_sde_kl.argtypes = [_dp] * 6 + [ctypes.c_double, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]
        lib.ref_cvi_dp_step_work_doubles.restype = ctypes.c_size_t
        lib.ref_cvi_dp_step_work_doubles.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.ref_cvi_dp_step.restype = ctypes.c_double
        lib.ref_cvi_dp_step.argtypes = ([ctypes.c_int] * 4 + [_ip, _dp, _dp, ctypes.c_double] + [_dp] * 6 + [ctypes.c_double]
                                        + [_dp] * 7 + [ctypes.c_double, ctypes.c_double, _dp, _dp])
        lib.ref_vdp_work_doubles.restype = ctypes.c_size_t
        lib.ref_vdp_work_doubles.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.ref_vdp_forward.restype = None
        lib.ref_vdp_forward.argtypes = [ctypes.c_int] * 3 + [_dp] * 3 + [ctypes.c_double, _dp, _dp, ctypes.c_int, _dp, _dp]
        lib.ref_vdp_step.restype = ctypes.c_double
        lib.ref_vdp_step.argtypes = ([ctypes.c_int] * 4 + [_ip, _dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp,
                                     ctypes.c_double] + [_dp] * 8 + [ctypes.c_double, ctypes.c_int, _dp, _dp])
        lib.ref_cvigp_step_work_doubles.restype = ctypes.c_size_t
        lib.ref_cvigp_step_work_doubles.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.ref_cvigp_step.restype = ctypes.c_double
        lib.ref_cvigp_step.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double, _dp, _dp, ctypes.c_double, ctypes.c_double,
                                       _dp, _dp, _dp]
        lib.ref_sparse_cvi_step_work_doubles.restype = ctypes.c_size_t
        lib.ref_sparse_cvi_step_work_doubles.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.ref_sparse_cvi_step.restype = ctypes.c_double
        lib.ref_sparse_cvi_step.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _dp, _dp, _dp, ctypes.c_double, ctypes.c_double,
                                            _dp, _dp, ctypes.c_double, _dp, _dp, _dp, _dp]
        lib.ref_num_threads.restype = ctypes.c_int
        lib.ref_set_num_threads.argtypes = [ctypes.c_int]
        lib.ref_set_num_threads.restype = None
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def btd_cholesky(diag, sub):
    lib = load()
    diag, sub = c64(diag), c64(sub)
    T, d = diag.shape[0], diag.shape[-1]
    Ld, Ls = np.zeros_like(diag), np.zeros_like(sub)
    bad = lib.ref_btd_cholesky(_p(diag), _p(sub), _p(Ld), _p(Ls), T, d)
    if bad:
        raise np.linalg.LinAlgError("not positive definite")
    return Ld, Ls


def btd_solve(Ld, Ls, rhs, transpose=False):
    lib = load()
    Ld, Ls, rhs = c64(Ld), c64(Ls), c64(rhs)
    out = np.zeros_like(rhs)
    lib.ref_btd_solve(_p(Ld), _p(Ls), _p(rhs), _p(out), Ld.shape[0], Ld.shape[-1], int(transpose))
    return out


def btd_inverse_blocks(Ld, Ls):
    lib = load()
    Ld, Ls = c64(Ld), c64(Ls)
    Sd, Ss = np.zeros_like(Ld), np.zeros_like(Ls)
    lib.ref_btd_inverse_blocks(_p(Ld), _p(Ls), _p(Sd), _p(Ss), Ld.shape[0], Ld.shape[-1])
    return Sd, Ss


def ssm_to_naturals(A, off, chol):
    lib = load()
    A, off, chol = c64(A), c64(off), c64(chol)
    T, d = off.shape
    lin, diag, sub = np.zeros((T, d)), np.zeros((T, d, d)), np.zeros((T - 1, d, d))
    lib.ref_ssm_to_naturals(_p(A), _p(off), _p(chol), _p(lin), _p(diag), _p(sub), T, d)
    return lin, diag, sub


class CviStepState:
    """Host arrays for ref_cvi_step (B chains)."""

    def __init__(self, p1, pd, ps, pmu, pslc, idx, y, Rinv, logdetR):
        self.p1, self.pd, self.ps, self.pmu, self.pslc = c64(p1), c64(pd), c64(ps), c64(pmu), c64(pslc)
        self.B, self.T, self.d = self.p1.shape
        self.idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.y, self.Rinv, self.logdetR = c64(y), c64(Rinv), float(logdetR)
        self.n = self.idx.shape[0]
        B, T, d, n = self.B, self.T, self.d, self.n
        self.g1 = np.zeros((B, T, d))
        self.g2d = -1e-10 * np.ones((B, T, d, d))
        self.g2s = -1e-10 * np.ones((B, T - 1, d, d))
        self.d1 = np.zeros((B, n, d))
        self.d2 = 1e-10 * np.broadcast_to(np.eye(d), (B, n, d, d)).copy()
        lib = load()
        self.work = np.zeros(B * lib.ref_cvi_step_work_doubles(T, d))
        self.elbo = np.zeros(B)

    def step(self, lr_d, lr_g):
        lib = load()
        return lib.ref_cvi_step(self.B, self.T, self.d, self.n, self.idx.ctypes.data_as(_ip), _p(self.y), _p(self.Rinv),
                                self.logdetR, _p(self.p1), _p(self.pd), _p(self.ps), _p(self.pmu), _p(self.pslc),
                                _p(self.g1), _p(self.g2d), _p(self.g2s), _p(self.d1), _p(self.d2), lr_d, lr_g,
                                _p(self.work), _p(self.elbo))


def sde_kl(mu, Sig, Sub, alpha, beta, qdiag, dt, init_mu, init_cov, want_grads=True):
    lib = load()
    mu, Sig, Sub = c64(mu), c64(Sig), c64(Sub)
    T, d = mu.shape
    al, be, qd = c64(np.broadcast_to(alpha, (d,))), c64(np.broadcast_to(beta, (d,))), c64(qdiag)
    g1, gd, gs = np.zeros((T, d)), np.zeros((T, d, d)), np.zeros((T - 1, d, d))
    kl = lib.ref_sde_kl(_p(mu), _p(Sig), _p(Sub), _p(al), _p(be), _p(qd), float(dt), _p(c64(init_mu)), _p(c64(init_cov)), T, d,
                        int(want_grads), _p(g1), _p(gd), _p(gs))
    return (kl, (g1, gd, gs)) if want_grads else kl


class CviDpStepState(CviStepState):
    """Host arrays for ref_cvi_dp_step (B chains, fixed linearised prior naturals)."""

    def __init__(self, p1, pd, ps, idx, y, Rinv, logdetR, alpha, beta, qdiag, dt, init_mu, init_cov):
        B, T, d = np.asarray(p1).shape
        super().__init__(p1, pd, ps, np.zeros((B, T, d)), np.zeros(B), idx, y, Rinv, logdetR)
        self.alpha, self.beta = c64(np.broadcast_to(alpha, (d,))), c64(np.broadcast_to(beta, (d,)))
        self.qdiag, self.dt = c64(qdiag), float(dt)
        self.init_mu, self.init_cov = c64(init_mu), c64(init_cov)
        lib = load()
        self.work = np.zeros(B * lib.ref_cvi_dp_step_work_doubles(T, d))

    def step(self, lr_d, lr_g):
        lib = load()
        return lib.ref_cvi_dp_step(self.B, self.T, self.d, self.n, self.idx.ctypes.data_as(_ip), _p(self.y), _p(self.Rinv),
                                   self.logdetR, _p(self.p1), _p(self.pd), _p(self.ps), _p(self.alpha), _p(self.beta),
                                   _p(self.qdiag), self.dt, _p(self.init_mu), _p(self.init_cov), _p(self.g1), _p(self.g2d),
                                   _p(self.g2s), _p(self.d1), _p(self.d2), lr_d, lr_g, _p(self.work), _p(self.elbo))


class VdpStepState:
    """Host arrays for ref_vdp_step: B trajectories of the VDP model (oracle/np_models.VariationalMarkovGP, closed_form=True) with a
    per-dimension cubic drift af x - bf x^3, diagonal q, Gaussian likelihood; q(x0) fixed at the prior's initial state."""

    def __init__(self, A, b, idx, y, Rinv, logdetR, af, bf, qdiag, dt, p0_mu, p0_cov, stabilize=True):
        self.A, self.b = c64(A).copy(), c64(b).copy()
        self.B, N, self.d = self.b.shape
        self.T = N + 1
        self.idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.n = self.idx.shape[0]
        self.y, self.Rinv, self.logdetR = c64(y), c64(Rinv), float(logdetR)
        self.af, self.bf, self.qdiag, self.dt = float(af), float(bf), c64(qdiag), float(dt)
        self.p0_mu, self.p0_cov = c64(p0_mu), c64(p0_cov)
        self.q0_mu, self.q0_chol = self.p0_mu.copy(), np.linalg.cholesky(self.p0_cov)
        self.stabilize = int(bool(stabilize))
        lib = load()
        self.m, self.S = np.zeros((self.B, self.T, self.d)), np.zeros((self.B, self.T, self.d, self.d))
        self.work = np.zeros(self.B * lib.ref_vdp_work_doubles(self.T, self.d))
        self.elbo = np.zeros(self.B)
        lib.ref_vdp_forward(self.B, self.T, self.d, _p(self.A), _p(self.b), _p(self.qdiag), self.dt, _p(self.q0_mu), _p(self.q0_chol),
                            self.stabilize, _p(self.m), _p(self.S))

    def step(self, lr):
        lib = load()
        return lib.ref_vdp_step(self.B, self.T, self.d, self.n, self.idx.ctypes.data_as(_ip), _p(self.y), _p(self.Rinv), self.logdetR,
                                self.af, self.bf, _p(self.qdiag), self.dt, _p(self.q0_mu), _p(self.q0_chol), _p(self.p0_mu),
                                _p(self.p0_cov), _p(self.A), _p(self.b), _p(self.m), _p(self.S), float(lr), self.stabilize,
                                _p(self.work), _p(self.elbo))



class CviGpStepState:
    """Host arrays for ref_cvigp_step: CVI for GP regression with a state-space kernel (oracle/np_models.CVIGaussianProcess), one chain,
    scalar Gaussian likelihood.  `ssm` is the oracle's prior StateSpaceModel on the data's time points (zero mean), `h` the emission row."""

    def __init__(self, ssm, h, y, noise_variance, learning_rate):
        pd, ps = ssm.precision()
        self.pd, self.ps = c64(pd), c64(ps)
        self.T, self.d = self.pd.shape[0], self.pd.shape[-1]
        self.half_logdet_prior = 0.5 * float(ssm.log_det_precision())
        self.h, self.y = c64(h).reshape(-1), c64(y).reshape(-1)
        assert self.h.shape[0] == self.d and self.y.shape[0] == self.T
        self.s2, self.lr = float(noise_variance), float(learning_rate)
        self.nat1, self.nat2 = np.zeros(self.T), -1e-10 * np.ones(self.T)
        self.work = np.zeros(load().ref_cvigp_step_work_doubles(self.T, self.d))

    def step(self):
        """update_sites(); elbo() -> the ELBO."""
        return load().ref_cvigp_step(self.T, self.d, _p(self.pd), _p(self.ps), self.half_logdet_prior, _p(self.h), _p(self.y), self.s2,
                                     self.lr, _p(self.nat1), _p(self.nat2), _p(self.work))


class SparseCviStepState:
    """Host arrays for ref_sparse_cvi_step: sparse / inducing-state CVI (oracle/np_conditionals.SparseCVIGaussianProcess), one chain, scalar
    Gaussian likelihood.  The data-dependent constants -- interval index, h^T P_n, h^T T_n h of every data point -- and the prior's
    precision blocks come from the NumPy oracle's kernel, once (the GPU model computes its own once as well)."""

    def __init__(self, kernel, z, t, y, noise_variance, learning_rate):
        from . import np_conditionals as npc
        z, t = np.asarray(z, dtype=np.float64), np.asarray(t, dtype=np.float64)
        ssm = kernel.state_space_model(z)
        pd, ps = ssm.precision()
        self.pd, self.ps = c64(pd), c64(ps)
        self.M, self.d = self.pd.shape[0], self.pd.shape[-1]
        self.pslc = -0.5 * float(ssm.log_det_precision())
        P, T, idx = npc.conditional_statistics(t, z, kernel)
        h = kernel.emission_vector()[0]
        self.w = c64(np.einsum("i,nij->nj", h, P))
        self.c = c64(np.einsum("i,nij,j->n", h, T, h))
        self.idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.N = self.idx.shape[0]
        self.y = c64(y).reshape(-1)
        self.pinf = c64(kernel.steady_state_covariance() + kernel.jitter * np.eye(self.d))
        self.s2, self.lr = float(noise_variance), float(learning_rate)
        self.nat1 = np.zeros((self.M + 1, 2 * self.d))
        self.nat2 = np.zeros((self.M + 1, 2 * self.d, 2 * self.d))
        self.work = np.zeros(load().ref_sparse_cvi_step_work_doubles(self.M, self.d))

    def step(self):
        """update_sites(data); classic_elbo(data) -> the ELBO."""
        return load().ref_sparse_cvi_step(self.M, self.d, self.N, self.idx.ctypes.data_as(_ip), _p(self.w), _p(self.c), _p(self.y), self.s2,
                                          self.lr, _p(self.pd), _p(self.ps), self.pslc, _p(self.pinf), _p(self.nat1), _p(self.nat2),
                                          _p(self.work))
