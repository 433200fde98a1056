"""
CPU tests pinning the NumPy oracle: dense numpy.linalg identities (KA1, KA4, KA5 of SURVEY.md 8c,
mirroring the reference's tests/unit/test_block_tri_diag.py, test_state_space_model.py,
test_ssm_gaussian_transformations.py) and the golden vectors produced by the reference's own
NumPy Kalman filter / expm kernels (KA2, KA3, KA6).
"""
import numpy as np
import pytest

from oracle import np_btd, np_kalman, np_kernels, np_ssm, np_transforms
from tests.conftest import golden
from tests.helpers import random_spd_btd, random_ssm_params


@pytest.mark.parametrize("d,T", [(1, 1), (1, 4), (3, 1), (3, 4), (2, 4), (3, 5)])
@pytest.mark.parametrize("with_sub", [True, False])
def test_btd_against_dense(rng, batch_shape, d, T, with_sub):
    diag, sub, _, _ = random_spd_btd(rng, batch_shape, T, d, with_sub)
    dense = np_btd.to_dense(diag, sub, symmetric=True)
    Ld, Ls = np_btd.cholesky(diag, sub)
    Ldense = np_btd.to_dense(Ld, Ls, symmetric=False)
    np.testing.assert_allclose(Ldense, np.linalg.cholesky(dense), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(np_btd.abs_log_det(Ld), 0.5 * np.linalg.slogdet(dense)[1], rtol=1e-10)
    x = rng.normal(size=batch_shape + (T, d))
    xf = x.reshape(batch_shape + (T * d, 1))
    for tr in (False, True):
        M = np.swapaxes(Ldense, -1, -2) if tr else Ldense
        np.testing.assert_allclose(np_btd.solve(Ld, Ls, x, tr).reshape(xf.shape), np.linalg.solve(M, xf), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(np_btd.dense_mult(Ld, Ls, x, False, tr).reshape(xf.shape), M @ xf, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np_btd.dense_mult(diag, sub, x, True).reshape(xf.shape), dense @ xf, rtol=1e-10, atol=1e-12)
    inv = np.linalg.inv(dense)
    Sd, Ss = np_btd.inverse_blocks(Ld, Ls)
    for k in range(T):
        np.testing.assert_allclose(Sd[..., k, :, :], inv[..., k * d:(k + 1) * d, k * d:(k + 1) * d], rtol=1e-7, atol=1e-10)
        if Ss is not None and k < T - 1:
            np.testing.assert_allclose(Ss[..., k, :, :], inv[..., (k + 1) * d:(k + 2) * d, k * d:(k + 1) * d], rtol=1e-7, atol=1e-10)
    if with_sub and T > 1:
        u_s, cholD = np_btd.upper_diagonal_lower(diag, sub)
        Ut = np_btd.to_dense(np.broadcast_to(np.eye(d), diag.shape), u_s, symmetric=False)
        D = np_btd.to_dense(cholD @ np.swapaxes(cholD, -1, -2), None)
        np.testing.assert_allclose(np.swapaxes(Ut, -1, -2) @ D @ Ut, dense, rtol=1e-6, atol=1e-9)


def test_cholesky_not_pd_raises():
    diag = -np.eye(2)[None].repeat(3, 0)
    with pytest.raises(np.linalg.LinAlgError):
        np_btd.cholesky(diag, None)


@pytest.mark.parametrize("d,T", [(1, 2), (3, 4), (5, 6)])
def test_ssm_identities(rng, batch_shape, d, T):
    ssm = np_ssm.StateSpaceModel(*random_ssm_params(rng, batch_shape, T, d))
    diag, sub = ssm.precision()
    dense = np_btd.to_dense(diag, sub)
    np.testing.assert_allclose(ssm.log_det_precision(), np.linalg.slogdet(dense)[1], rtol=1e-9)
    # explicit forward recursions (reference tests/unit/test_state_space_model.py:63-101)
    P0 = ssm.cholP0 @ np.swapaxes(ssm.cholP0, -1, -2)
    Q = ssm.cholQ @ np.swapaxes(ssm.cholQ, -1, -2)
    mu, cov = ssm.marginals
    m, P = ssm.mu0, P0
    for k in range(T):
        np.testing.assert_allclose(mu[..., k, :], m, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(cov[..., k, :, :], P, rtol=1e-7, atol=1e-9)
        if k < T - 1:
            A = ssm.A[..., k, :, :]
            np.testing.assert_allclose(ssm.subsequent_covariances(cov)[..., k, :, :], A @ P, rtol=1e-7, atol=1e-9)
            m = (A @ m[..., None])[..., 0] + ssm.b[..., k, :]
            P = A @ P @ np.swapaxes(A, -1, -2) + Q[..., k, :, :]
    # the dense covariance's inverse is the precision
    np.testing.assert_allclose(np.linalg.inv(dense)[..., :d, :d], P0, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(ssm.kl_divergence(ssm), 0.0, atol=1e-6)
    # KL against the dense formula
    other = np_ssm.StateSpaceModel(*random_ssm_params(rng, batch_shape, T, d))
    d2, s2 = other.precision()
    P2 = np_btd.to_dense(d2, s2)
    S1 = np.linalg.inv(dense)
    dm = (other.marginal_means - ssm.marginal_means).reshape(batch_shape + (T * d, 1))
    kl = 0.5 * (np.trace(P2 @ S1, axis1=-2, axis2=-1) + (np.swapaxes(dm, -1, -2) @ P2 @ dm)[..., 0, 0] - T * d
                - np.linalg.slogdet(P2)[1] + np.linalg.slogdet(dense)[1])
    np.testing.assert_allclose(ssm.kl_divergence(other), kl, rtol=1e-7, atol=1e-8)
    # log pdf against the dense MVN
    x = ssm.sample((2,), rng)
    xf = x.reshape(x.shape[:-2] + (T * d,)) - mu.reshape(batch_shape + (T * d,))
    ref = -0.5 * np.einsum("...i,...ij,...j->...", xf, np.broadcast_to(dense, xf.shape[:-1] + dense.shape[-2:]), xf) \
        + 0.5 * np.linalg.slogdet(dense)[1] - 0.5 * T * d * np.log(2 * np.pi)
    np.testing.assert_allclose(ssm.log_pdf(x), ref, rtol=1e-8, atol=1e-8)


def test_zero_transitions_raises(rng):
    with pytest.raises(ValueError):
        np_ssm.StateSpaceModel(np.zeros(2), np.eye(2), np.zeros((0, 2, 2)), np.zeros((0, 2)), np.zeros((0, 2, 2)))


def _ssm_from_golden(g, batch_shape, T):
    d = g["A"].shape[-1]
    return np_ssm.StateSpaceModel(
        np.broadcast_to(g["mu0"], batch_shape + (d,)), np.broadcast_to(g["cholP0"], batch_shape + (d, d)),
        np.broadcast_to(g["A"], batch_shape + (T - 1, d, d)), np.broadcast_to(g["b"], batch_shape + (T - 1, d)),
        np.broadcast_to(g["cholQ"], batch_shape + (T - 1, d, d)))


@pytest.mark.parametrize("tag,bs", [("b0", ()), ("b3", (3,)), ("b21", (2, 1))])
def test_kalman_filter_golden(tag, bs):
    """KA2: reference tests/integration/test_kalman_filter.py:105-139 against its NumPy filter."""
    g = golden(f"kalman_filter_{tag}.npz")
    T = g["y"].shape[-2]
    ssm = _ssm_from_golden(g, bs, T)
    H = np.broadcast_to(g["H"], bs + (T,) + g["H"].shape)
    kf = np_kalman.KalmanFilter(ssm, H, g["y"], np.linalg.cholesky(g["R"]))
    np.testing.assert_allclose(kf.log_likelihood(), g["log_lik_total"], rtol=1e-7)
    post = kf.posterior_state_space_model()
    np.testing.assert_allclose(post.marginal_means, g["smooth_means"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(post.marginal_covariances, np.broadcast_to(g["smooth_covs"], bs + g["smooth_covs"].shape), rtol=1e-6, atol=1e-8)


def test_kalman_filter_sites_golden():
    """KA3: reference tests/integration/test_kalman_filter_with_sites.py (dead test there, live here)."""
    g = golden("kalman_filter_sites.npz")
    T = g["site_means"].shape[0]
    ssm = _ssm_from_golden(g, (), T)
    H = np.broadcast_to(g["H"], (T,) + g["H"].shape)
    sites = np_kalman.GaussianSitesNat(g["site_means"] / g["site_covs"][..., 0], -0.5 / g["site_covs"])
    kf = np_kalman.KalmanFilterWithSites(ssm, H, sites)
    np.testing.assert_allclose(kf.log_likelihood(), g["log_lik_total"], rtol=1e-7)
    post = kf.posterior_state_space_model()
    np.testing.assert_allclose(post.marginal_means, g["smooth_means"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(post.marginal_covariances, g["smooth_covs"], rtol=1e-6, atol=1e-8)


def test_matern_golden():
    """KA6: reference tests/unit/test_matern.py:78-124 (expm reference, atol 1e-8)."""
    g = golden("matern_expm.npz")
    dts = g["time_deltas"]
    for name, cls in (("m12", np_kernels.Matern12), ("m32", np_kernels.Matern32), ("m52", np_kernels.Matern52)):
        for i in range(2):
            var, ls = g[f"{name}_{i}_params"]
            k = cls(lengthscale=ls, variance=var)
            A, Q = k.transition_statistics(dts)
            np.testing.assert_allclose(A, g[f"{name}_{i}_A"], atol=1e-8)
            np.testing.assert_allclose(Q, g[f"{name}_{i}_Q"], atol=1e-8)
            np.testing.assert_allclose(k.steady_state_covariance(), g[f"{name}_{i}_Pinf"], atol=1e-10)


def test_transform_round_trips():
    """KA5: reference tests/unit/test_ssm_gaussian_transformations.py:36-105, same setup
    (Sum of 10 Matern52(0.01, 0.01), linspace(0,1,1001), d=30) and tolerances (rtol 1e-7 / atol 1e-6)."""
    kern = np_kernels.Sum([np_kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(10)])
    ssm = kern.state_space_model(np.linspace(0, 1, 1001))
    ref = (ssm.A, ssm.b, ssm.cholP0, ssm.cholQ, ssm.mu0)
    for fwd, bwd in ((np_transforms.ssm_to_expectations, np_transforms.expectations_to_ssm_params),
                     (np_transforms.ssm_to_naturals, np_transforms.naturals_to_ssm_params),
                     (np_transforms.ssm_to_naturals_no_smoothing, np_transforms.naturals_to_ssm_params_no_smoothing)):
        back = bwd(*fwd(ssm))
        for a, b in zip(back, ref):
            np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-6)


def test_transform_round_trips_random(rng, batch_shape):
    ssm = np_ssm.StateSpaceModel(*random_ssm_params(rng, batch_shape, 9, 3))
    ref = (ssm.A, ssm.b, ssm.cholP0, ssm.cholQ, ssm.mu0)
    for fwd, bwd in ((np_transforms.ssm_to_expectations, np_transforms.expectations_to_ssm_params),
                     (np_transforms.ssm_to_naturals, np_transforms.naturals_to_ssm_params),
                     (np_transforms.ssm_to_naturals_no_smoothing, np_transforms.naturals_to_ssm_params_no_smoothing)):
        back = bwd(*fwd(ssm))
        for a, b in zip(back, ref):
            np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-8)
