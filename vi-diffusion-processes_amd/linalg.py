"""
Batched small dense SPD algebra on [..., d, d] device tensors through libmfgm (`mfgm_batched_cholesky`,
`mfgm_batched_trsm`): the per-time-step tf.linalg.cholesky / cholesky_solve / triangular_solve calls of the reference
(ssm_gaussian_transformations.py:93-178, 459-511, 515-593; conditionals.py:207-256; kalman_filter.py:298-345).
torch only owns the memory; nothing here goes through a vendor batched LAPACK.
"""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_dev(*ts):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float64):
            raise ValueError("vidp_amd.linalg works on fp64 device tensors (there is no CPU fallback for the product path)")


def cholesky(A, check=True):
    """Lower Cholesky factors of SPD blocks A [..., d, d]; raises ArithmeticError on a non-positive pivot."""
    _check_dev(A)
    lib = _lib.load()
    d = A.shape[-1]
    a = A.contiguous()
    N = a.numel() // (d * d) if a.numel() else 0
    L = torch.empty_like(a)
    info = torch.zeros(1, dtype=torch.int32, device=a.device)
    _lib.check(lib.mfgm_batched_cholesky(N, d, _ptr(a), _ptr(L), _ptr(info), _stream()), "mfgm_batched_cholesky")
    if check and int(info.item()) != 0:
        raise ArithmeticError("cholesky: a block is not positive definite")
    return L


def _trsm(L, B, mode):
    _check_dev(L, B)
    lib = _lib.load()
    d = L.shape[-1]
    vec = (B.dim() == L.dim() - 1)
    b = B[..., None] if vec else B
    m = b.shape[-1]
    if b.shape[-2] != d or L.shape[-2] != d:
        raise ValueError(f"triangular solve: shapes {tuple(L.shape)} and {tuple(B.shape)} do not match")
    batch = torch.broadcast_shapes(L.shape[:-2], b.shape[:-2])
    shared = (L.dim() == 2) or all(s == 1 for s in L.shape[:-2])
    bb = b.expand(batch + (d, m)).contiguous()
    N = bb.numel() // (d * m) if bb.numel() else 0
    if shared:
        ll, lbatch = L.reshape(d, d).contiguous(), 1
    else:
        ll, lbatch = L.expand(batch + (d, d)).contiguous(), N
    X = torch.empty_like(bb)
    _lib.check(lib.mfgm_batched_trsm(N, d, m, lbatch if N else 1, _ptr(ll), _ptr(bb), _ptr(X), mode, _stream()), "mfgm_batched_trsm")
    return X[..., 0] if vec else X


def cholesky_solve(B, L):
    """(L L^T)^{-1} B, argument order of torch.cholesky_solve.  B: [..., d, m] (or [..., d]); batch dims broadcast."""
    return _trsm(L, B, 3)


def solve_lower(L, B):
    """L^{-1} B."""
    return _trsm(L, B, 1)


def solve_lower_t(L, B):
    """L^{-T} B."""
    return _trsm(L, B, 2)


def spd_inverse(A=None, chol=None):
    """A^{-1} for SPD blocks, from A or from its Cholesky factor."""
    L = cholesky(A) if chol is None else chol
    d = L.shape[-1]
    eye = torch.eye(d, dtype=L.dtype, device=L.device).expand(L.shape)
    return _trsm(L, eye, 3)


def logdet_spd(A=None, chol=None):
    """log det A for SPD blocks."""
    L = cholesky(A) if chol is None else chol
    return 2.0 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)


def small_inverse(M):
    """Inverse of GENERAL (not necessarily symmetric) blocks [..., d, d], d <= 3, by cofactors (element-wise torch arithmetic; the
    reference's `tf.linalg.inv` on the 2 psi(0) + P0^{-1} of update_initial_statistics, vi_sde.py:241-260, whose psi is not symmetric)."""
    d = M.shape[-1]
    if d == 1:
        return 1.0 / M
    if d == 2:
        a, b, c, e = M[..., 0, 0], M[..., 0, 1], M[..., 1, 0], M[..., 1, 1]
        det = a * e - b * c
        return torch.stack([torch.stack([e, -b], -1), torch.stack([-c, a], -1)], -2) / det[..., None, None]
    if d != 3:
        raise ValueError("small_inverse covers d <= 3")
    m = [[M[..., i, j] for j in range(3)] for i in range(3)]
    cof = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(3):
            r = [k for k in range(3) if k != i]
            c = [k for k in range(3) if k != j]
            minor = m[r[0]][c[0]] * m[r[1]][c[1]] - m[r[0]][c[1]] * m[r[1]][c[0]]
            cof[i][j] = minor if (i + j) % 2 == 0 else -minor
    det = m[0][0] * cof[0][0] + m[0][1] * cof[0][1] + m[0][2] * cof[0][2]
    adj = torch.stack([torch.stack([cof[j][i] for j in range(3)], -1) for i in range(3)], -2)
    return adj / det[..., None, None]


def general_inverse(M):
    """Inverse of GENERAL square blocks [..., d, d] by Gauss-Jordan elimination with partial pivoting, written with element-wise /
    gather torch operations (batched over the leading axes; d steps).  For the handful of small non-symmetric systems of the models --
    `tf.linalg.inv(P0^{-1} + 2 psi(0))` of VDP's update_initial_statistics (vi_sde.py:241-260): psi is not symmetric, the Lagrange sweep
    adds psi A + psi A -- where the SPD kernels (mfgm_batched_cholesky) do not apply."""
    d = M.shape[-1]
    if d <= 3:
        return small_inverse(M)
    shape = M.shape
    A = M.reshape(-1, d, d).clone()
    n = A.shape[0]
    X = torch.eye(d, dtype=A.dtype, device=A.device).expand(n, d, d).clone()
    rows = torch.arange(n, device=A.device)
    for k in range(d):
        piv = A[:, k:, k].abs().argmax(dim=1) + k                     # [n]
        # swap rows k and piv
        for T_ in (A, X):
            rk, rp = T_[rows, k].clone(), T_[rows, piv].clone()
            T_[rows, k], T_[rows, piv] = rp, rk
        inv = 1.0 / A[:, k, k]
        A[:, k] = A[:, k] * inv[:, None]
        X[:, k] = X[:, k] * inv[:, None]
        f = A[:, :, k].clone()
        f[:, k] = 0.0
        A = A - f[:, :, None] * A[:, k][:, None, :]
        X = X - f[:, :, None] * X[:, k][:, None, :]
    return X.reshape(shape)

