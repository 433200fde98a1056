"""
Oracle (test infrastructure, see oracle/__init__.py): block-tri-diagonal matrices in NumPy.

Restates, on ``[..., T, d, d]`` / ``[..., T-1, d, d]`` block arrays, what the reference obtains from
``markovflow/block_tri_diag.py`` + the un-vendored native dependency ``banded-matrices==0.0.6``
(C++ TF custom ops: cholesky_band, solve_triang_mat, product_band_mat,
inverse_from_cholesky_band; call sites block_tri_diag.py:158,189,233,330-331,350,440,562).
The banded package is absent from /root/reference, so the published algorithms are restated
(banded Cholesky == block recursion L_kk = chol(D_k - L_{k,k-1} L_{k,k-1}^T),
L_{k+1,k} = S_k L_kk^{-T}; Takahashi recursion for the band of the inverse) and pinned against
dense ``numpy.linalg`` exactly as tests/unit/test_block_tri_diag.py does.
"""
import numpy as np


def _T(x):
    return np.swapaxes(x, -1, -2)


def solve_lower(L, B):
    """L^{-1} B for batched lower-triangular L (plain dense solve; L is triangular)."""
    return np.linalg.solve(L, B)


def solve_upper_from_lower(L, B):
    """L^{-T} B."""
    return np.linalg.solve(_T(L), B)


def to_dense(diag, sub=None, symmetric=True):
    """block_tri_diag.py:150-158 (`to_dense`): dense [..., T*d, T*d]."""
    diag = np.asarray(diag)
    T, d = diag.shape[-3], diag.shape[-1]
    out = np.zeros(diag.shape[:-3] + (T * d, T * d), dtype=diag.dtype)
    for k in range(T):
        out[..., k * d:(k + 1) * d, k * d:(k + 1) * d] = diag[..., k, :, :]
    if sub is not None:
        for k in range(T - 1):
            out[..., (k + 1) * d:(k + 2) * d, k * d:(k + 1) * d] = sub[..., k, :, :]
            if symmetric:
                out[..., k * d:(k + 1) * d, (k + 1) * d:(k + 2) * d] = _T(sub[..., k, :, :])
    return out


def cholesky(diag, sub=None):
    """
    SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:428-440 -> cholesky_band).
    Only the lower triangle of each diagonal block is read (cholesky_band operates on the
    lower band).  Returns (L_diag [..., T, d, d] lower-triangular, L_sub [..., T-1, d, d] or None).
    Raises np.linalg.LinAlgError when a pivot block is not positive definite.
    """
    diag = np.asarray(diag, dtype=np.float64)
    T = diag.shape[-3]
    low = np.tril(diag)
    symd = low + _T(np.tril(diag, -1))
    Ld = np.empty_like(diag)
    Ls = None if sub is None else np.empty_like(np.asarray(sub, dtype=np.float64))
    carry = np.zeros_like(diag[..., 0, :, :])
    for k in range(T):
        Lk = np.linalg.cholesky(symd[..., k, :, :] - carry)
        Ld[..., k, :, :] = Lk
        if sub is not None and k < T - 1:
            # L_{k+1,k} = S_k L_kk^{-T}
            Lsk = _T(solve_lower(Lk, _T(sub[..., k, :, :])))
            Ls[..., k, :, :] = Lsk
            carry = Lsk @ _T(Lsk)
        else:
            carry = np.zeros_like(carry)
    return Ld, Ls


def solve(Ld, Ls, rhs, transpose_left=False):
    """
    LowerTriangularBlockTriDiagonal.solve (block_tri_diag.py:339-351 -> solve_triang_mat):
    L^{-1} rhs or L^{-T} rhs with rhs [..., T, d].
    """
    Ld = np.asarray(Ld, dtype=np.float64)
    rhs = np.asarray(rhs, dtype=np.float64)
    T = Ld.shape[-3]
    out = np.empty(np.broadcast_shapes(rhs.shape, Ld.shape[:-1]), dtype=np.float64)
    if not transpose_left:
        prev = None
        for k in range(T):
            r = rhs[..., k, :]
            if Ls is not None and k > 0:
                r = r - (Ls[..., k - 1, :, :] @ prev[..., None])[..., 0]
            prev = solve_lower(Ld[..., k, :, :], r[..., None])[..., 0]
            out[..., k, :] = prev
    else:
        nxt = None
        for k in range(T - 1, -1, -1):
            r = rhs[..., k, :]
            if Ls is not None and k < T - 1:
                r = r - (_T(Ls[..., k, :, :]) @ nxt[..., None])[..., 0]
            nxt = solve_upper_from_lower(Ld[..., k, :, :], r[..., None])[..., 0]
            out[..., k, :] = nxt
    return out


def abs_log_det(Ld):
    """LowerTriangularBlockTriDiagonal.abs_log_det (block_tri_diag.py:353-366)."""
    dg = np.diagonal(Ld, axis1=-2, axis2=-1)
    return 0.5 * np.sum(np.log(np.square(dg)), axis=(-1, -2))


def inverse_blocks(Ld, Ls):
    """
    Diagonal and sub-diagonal blocks of (L L^T)^{-1} (block_tri_diag.py:318-337 and
    ssm_gaussian_transformations.py:443-458 -> inverse_from_cholesky_band), by the backward
    Takahashi recursion:
        S_TT = L_TT^{-T} L_TT^{-1}
        S_{k+1,k} = -S_{k+1,k+1} L_{k+1,k} L_kk^{-1}
        S_kk = L_kk^{-T} L_kk^{-1} - S_{k+1,k}^T L_{k+1,k} L_kk^{-1}
    Returns (S_diag [..., T, d, d], S_sub [..., T-1, d, d] or None), S_sub[k] = S_{k+1,k}.
    """
    Ld = np.asarray(Ld, dtype=np.float64)
    T, d = Ld.shape[-3], Ld.shape[-1]
    eye = np.eye(d)
    Sd = np.empty_like(Ld)
    Ss = None if Ls is None else np.empty_like(Ls)
    for k in range(T - 1, -1, -1):
        Linv = solve_lower(Ld[..., k, :, :], np.broadcast_to(eye, Ld[..., k, :, :].shape))
        base = _T(Linv) @ Linv
        if Ls is not None and k < T - 1:
            G = Ls[..., k, :, :] @ Linv  # L_{k+1,k} L_kk^{-1}
            Ssk = -Sd[..., k + 1, :, :] @ G
            Ss[..., k, :, :] = Ssk
            base = base - _T(Ssk) @ G
        Sd[..., k, :, :] = 0.5 * (base + _T(base))
    return Sd, Ss


def block_diagonal_of_inverse(Ld, Ls):
    """LowerTriangularBlockTriDiagonal.block_diagonal_of_inverse (block_tri_diag.py:318-337)."""
    return inverse_blocks(Ld, Ls)[0]


def dense_mult(diag, sub, x, symmetric, transpose_left=False):
    """
    BlockTriDiagonal.dense_mult (block_tri_diag.py:175-199 -> product_band_mat): M x, M^T x for a
    lower block-bidiagonal M, or (M + M^T - diag) x for the symmetric flavour.  x: [..., T, d].
    """
    diag = np.asarray(diag, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    if symmetric:
        low = np.tril(diag)
        dsym = low + _T(np.tril(diag, -1))
        out = (dsym @ x[..., None])[..., 0]
        if sub is not None:
            out[..., 1:, :] += (sub @ x[..., :-1, :, None])[..., 0]
            out[..., :-1, :] += (_T(sub) @ x[..., 1:, :, None])[..., 0]
        return out
    if not transpose_left:
        out = (diag @ x[..., None])[..., 0]
        if sub is not None:
            out[..., 1:, :] += (sub @ x[..., :-1, :, None])[..., 0]
    else:
        out = (_T(diag) @ x[..., None])[..., 0]
        if sub is not None:
            out[..., :-1, :] += (_T(sub) @ x[..., 1:, :, None])[..., 0]
    return out


def upper_diagonal_lower(diag, sub):
    """
    SymmetricBlockTriDiagonal.upper_diagonal_lower (block_tri_diag.py:442-549): backward UDU^T.
        D_n = K_nn ; D_k = K_kk - K_{k+1,k}^T D_{k+1}^{-1} K_{k+1,k} ; U_k^T = D_k^{-1} K_{k,k-1}
    Returns (u_sub [..., T-1, d, d] (the sub-diagonal of U^T whose diagonal is identity),
             chol_D [..., T, d, d]).
    """
    diag = np.asarray(diag, dtype=np.float64)
    T = diag.shape[-3]
    low = np.tril(diag)
    symd = low + _T(np.tril(diag, -1))
    cholD = np.empty_like(diag)
    cholD[..., T - 1, :, :] = np.linalg.cholesky(symd[..., T - 1, :, :])
    for k in range(T - 2, -1, -1):
        Sk = sub[..., k, :, :]
        c = cholD[..., k + 1, :, :]
        dinv_s = solve_upper_from_lower(c, solve_lower(c, Sk))
        cholD[..., k, :, :] = np.linalg.cholesky(symd[..., k, :, :] - _T(Sk) @ dinv_s)
    c = cholD[..., 1:, :, :]
    u_s = solve_upper_from_lower(c, solve_lower(c, sub))
    return u_s, cholD
