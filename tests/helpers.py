"""Random problem generators shared by the oracle and GPU parity tests."""
import numpy as np


def random_spd_btd(rng, batch, T, d, with_sub=True):
    """
    Positive-definite block-tri-diagonal (diag, sub) as in the reference's
    tests/unit/test_block_tri_diag.py:274-296: build a random lower block-bidiagonal L and form L L^T.
    """
    Ld = np.tril(0.5 * rng.normal(size=batch + (T, d, d)))
    idx = np.arange(d)
    Ld[..., idx, idx] = np.abs(Ld[..., idx, idx]) + 1.0
    Ls = 0.5 * rng.normal(size=batch + (T - 1, d, d)) if (with_sub and T > 1) else None
    diag = Ld @ np.swapaxes(Ld, -1, -2)
    sub = None
    if Ls is not None:
        diag[..., 1:, :, :] += Ls @ np.swapaxes(Ls, -1, -2)
        sub = Ls @ np.swapaxes(Ld[..., :-1, :, :], -1, -2)
    return diag, sub, Ld, Ls


def random_dominant_btd(rng, batch, T, d):
    """
    Well-conditioned SPD block-tri-diagonal (cond ~ 10 whatever T): D_t = M M^T / d + 2.5 I,
    |S_t| ~ 0.8 in spectral norm, so block Gershgorin keeps the smallest eigenvalue near 1.
    (The L L^T generator above becomes exponentially ill-conditioned in T and is kept for tiny T only.)
    """
    M = rng.normal(size=batch + (T, d, d))
    diag = M @ np.swapaxes(M, -1, -2) / d + 2.5 * np.eye(d)
    sub = 0.4 * rng.normal(size=batch + (max(T - 1, 0), d, d)) / np.sqrt(d)
    return diag, (sub if T > 1 else None)


def random_ssm_params(rng, batch, T, d, scale_A=0.6):
    """Random stable SSM parameters (mu0, cholP0, A, b, cholQ)."""
    A = scale_A * rng.normal(size=batch + (T - 1, d, d)) / np.sqrt(d)
    b = rng.normal(size=batch + (T - 1, d))
    idx = np.arange(d)
    cholQ = np.tril(0.3 * rng.normal(size=batch + (T - 1, d, d)))
    cholQ[..., idx, idx] = np.abs(cholQ[..., idx, idx]) + 0.5
    cholP0 = np.tril(0.3 * rng.normal(size=batch + (d, d)))
    cholP0[..., idx, idx] = np.abs(cholP0[..., idx, idx]) + 0.5
    mu0 = rng.normal(size=batch + (d,))
    return mu0, cholP0, A, b, cholQ


def assert_close(actual, desired, rtol=1e-6, scale_atol=1e-8):
    """Element-wise rtol plus an absolute floor tied to the tensor's own magnitude (fp64 parity; the
    north-star bound is 1e-5 relative)."""
    desired = np.asarray(desired)
    atol = scale_atol * max(1.0, float(np.max(np.abs(desired))) if desired.size else 1.0)
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol)
