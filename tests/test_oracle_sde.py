"""
CPU tests of the oracle's SDE layer: the reference's quadrature formulation (mvnquad) against the closed-form
Gaussian moments that the HIP kernels use, and the hand-derived d KL / d eta against finite differences of the
quadrature KL (the stand-in for the reference's GradientTape).  KA10 (OU linearisation) as in
tests/unit/test_sde.py:66-103.
"""
import numpy as np
import pytest

from oracle import np_sde, np_ssm, np_transforms
from tests.helpers import random_ssm_params


def test_mvnquad_moments(rng):
    m = rng.normal(size=(4, 2))
    L = np.tril(rng.normal(size=(4, 2, 2))) * 0.3 + np.eye(2)
    S = L @ np.swapaxes(L, -1, -2)
    got = np_sde.mvnquad(lambda x: x[:, :1] * x[:, 1:], m, S, 5, 2, (1,))[:, 0]
    np.testing.assert_allclose(got, S[:, 0, 1] + m[:, 0] * m[:, 1], rtol=1e-12)


def test_ou_linearisation_analytic(rng):
    """KA10: linearize_sde of an OU process is exact: A = 1 - decay dt, b = 0, Q = q dt (test_sde.py:66-103)."""
    decay, q, dt, N = 0.7, 1.3, 0.01, 30
    sde = np_sde.OrnsteinUhlenbeckSDE(decay, q * np.eye(1))
    t = np.arange(N + 1) * dt
    ssm = np_sde.linearize_sde(sde, t, rng.normal(size=(N, 1)), 0.5 + rng.random((N, 1, 1)), np.zeros(1), np.eye(1))
    np.testing.assert_allclose(ssm.A, (1 - decay * dt) * np.ones((N, 1, 1)), atol=1e-12)
    np.testing.assert_allclose(ssm.b, 0.0, atol=1e-12)
    np.testing.assert_allclose(ssm.cholQ, np.sqrt(q * dt) * np.ones((N, 1, 1)), atol=1e-12)


@pytest.mark.parametrize("d", [1, 2])
@pytest.mark.parametrize("kind", ["ou", "dw"])
def test_closed_form_kl_and_gradient(rng, d, kind):
    T, dt = 5, 0.05
    qd = 0.5 + rng.random(d)
    sde = np_sde.OrnsteinUhlenbeckSDE(0.8, np.diag(qd)) if kind == "ou" else np_sde.DoubleWellSDE(np.diag(qd))
    q = np_ssm.StateSpaceModel(*random_ssm_params(rng, (), T, d, scale_A=0.8))
    eta = np_transforms.ssm_to_expectations(q)
    init_mu, init_cov = 0.1 * rng.normal(size=d), np.eye(d) * 0.7
    kl_quad = np_sde.sde_ssm_kl_from_expectations(*eta, sde, dt, init_mu, init_cov)
    mu, Sig = q.marginals
    Sub = q.subsequent_covariances(Sig)
    alpha, beta = sde.cubic(dt)
    kl_cf, grads = np_sde.sde_ssm_kl_closed_form(mu, Sig, Sub, alpha, beta, qd, dt, init_mu, init_cov)
    np.testing.assert_allclose(kl_cf, kl_quad, rtol=1e-9)
    fd = np_sde.sde_ssm_kl_grads_fd(*eta, sde, dt, init_mu, init_cov, eps=1e-5)
    for a, b in zip(grads, fd):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * max(1.0, np.max(np.abs(b))))


def test_linear_prior_gradient_is_natural_parameter_difference(rng):
    """For an OU (linear) prior the gradient equals theta_q - theta_p of the Euler SSM (the identity KA11 relies on)."""
    T, dt, d = 6, 0.02, 1
    sde = np_sde.OrnsteinUhlenbeckSDE(1.2, np.eye(1))
    q = np_ssm.StateSpaceModel(*random_ssm_params(rng, (), T, d, scale_A=0.8))
    mu, Sig = q.marginals
    init_mu, init_cov = np.zeros(1), np.eye(1) * 0.4
    _, grads = np_sde.sde_ssm_kl_closed_form(mu, Sig, q.subsequent_covariances(Sig), *sde.cubic(dt), np.ones(1), dt, init_mu, init_cov)
    p = np_ssm.StateSpaceModel(init_mu, np.linalg.cholesky(init_cov), np.full((T - 1, 1, 1), 1 - 1.2 * dt), np.zeros((T - 1, 1)),
                               np.full((T - 1, 1, 1), np.sqrt(dt)))
    for g, tq, tp in zip(grads, np_transforms.ssm_to_naturals(q), np_transforms.ssm_to_naturals(p)):
        np.testing.assert_allclose(g, tq - tp, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("d,kind", [(1, "dw"), (2, "dw"), (2, "ou")])
def test_e_sde_closed_form(rng, d, kind):
    """VDP drift-difference energy: the reference's quadrature (sde_utils.py:182-249) vs the closed form and its gradient."""
    N, dt = 4, 0.05
    qd = 0.5 + rng.random(d)
    sde = np_sde.OrnsteinUhlenbeckSDE(0.8, np.diag(qd)) if kind == "ou" else np_sde.DoubleWellSDE(np.diag(qd))
    A = 0.5 * rng.normal(size=(N, d, d))
    b = rng.normal(size=(N, d))
    m = rng.normal(size=(N, d))
    L = np.tril(0.3 * rng.normal(size=(N, d, d))) + 0.8 * np.eye(d)
    S = L @ np.swapaxes(L, -1, -2)
    ref = np_sde.squared_drift_difference_along_gaussian_path(sde, A, b, m, S, dt)
    af, bf = np_sde.drift_cubic(sde)
    E, dm, dS = np_sde.e_sde_closed_form(af, bf, qd, A, b, m, S, dt)
    np.testing.assert_allclose(E, ref, rtol=1e-10)
    f = lambda mm, SS: np_sde.squared_drift_difference_along_gaussian_path(sde, A, b, mm, SS, dt)
    eps = 1e-6
    for idx in np.ndindex(m.shape):
        e = np.zeros_like(m); e[idx] = eps
        np.testing.assert_allclose(dm[idx], (f(m + e, S) - f(m - e, S)) / (2 * eps), rtol=1e-5, atol=1e-7)
    for t, i, j in np.ndindex(S.shape):
        if j > i:
            continue
        e = np.zeros_like(S); e[t, i, j] = eps; e[t, j, i] = eps
        fd = (f(m, S + e) - f(m, S - e)) / (2 * eps)
        np.testing.assert_allclose(dS[t, i, j] * (1.0 if i == j else 2.0), fd, rtol=1e-5, atol=1e-7)


def test_non_polynomial_drifts_against_adaptive_quadrature(rng):
    """BenesSDE / SineDiffusionSDE / SqrtDiffusionSDE (sde.py:227-356): the 10-point rule of expected_drift and
    expected_gradient_drift (sde.py:92-131) against scipy's adaptive quadrature of the same Gaussian integrals."""
    from scipy import integrate
    for sde, m, v in ((np_sde.BenesSDE(1.3), 0.4, 0.3), (np_sde.SineDiffusionSDE(0.4), -0.7, 0.5), (np_sde.SqrtDiffusionSDE(1.5), 4.0, 0.2)):
        pdf = lambda x: np.exp(-0.5 * (x - m) ** 2 / v) / np.sqrt(2 * np.pi * v)
        Ef = integrate.quad(lambda x: float(sde.drift(np.array([x]))[0]) * pdf(x), m - 12 * np.sqrt(v), m + 12 * np.sqrt(v))[0]
        Eg = integrate.quad(lambda x: float(sde.gradient_drift(np.array([x]))[0]) * pdf(x), m - 12 * np.sqrt(v), m + 12 * np.sqrt(v))[0]
        mm, vv = np.full((1, 1, 1), m), np.full((1, 1, 1, 1), v)
        # tolerance = accuracy of the reference's own 10-point rule on these integrands
        np.testing.assert_allclose(sde.expected_drift(mm, vv)[0, 0, 0], Ef, rtol=5e-6)
        np.testing.assert_allclose(sde.expected_gradient_drift(mm, vv)[0, 0, 0], Eg, rtol=5e-5)


def test_oracle_vanderpol_linearisation_matches_closed_form(rng):
    """The coupled-drift route of the oracle (full expected Jacobian by the 10-point rule, reference sde.py:484-518) against the closed
    form the Van der Pol drift admits: E f1 = tau a (m1 - (m1^3 + 3 m1 S11) / 3 - m2), E f2 = tau m1 / a,
    E df/dx = tau [[a (1 - m1^2 - S11), -a], [1 / a, 0]] -- the rule is exact for this cubic."""
    from oracle import np_sde
    a, tau = 1.3, 0.9
    sde = np_sde.VanderPolSDE(a, tau, 0.5 * np.eye(2))
    N = 7
    m = rng.normal(size=(N, 2))
    L = np.tril(rng.normal(size=(N, 2, 2))) + 1.5 * np.eye(2)
    S = L @ np.swapaxes(L, -1, -2)
    Ef = sde.expected_drift(m[None], S[None])[0]
    J = np_sde.mvnquad(lambda x: sde.jacobian_drift(x), m, S, 10, 2, (2, 2))
    m1, m2, s11 = m[:, 0], m[:, 1], S[:, 0, 0]
    np.testing.assert_allclose(Ef[:, 0], tau * a * (m1 - (m1 ** 3 + 3 * m1 * s11) / 3 - m2), rtol=1e-10)
    np.testing.assert_allclose(Ef[:, 1], tau * m1 / a, rtol=1e-10)
    np.testing.assert_allclose(J[:, 0, 0], tau * a * (1 - m1 ** 2 - s11), rtol=1e-10)
    np.testing.assert_allclose(J[:, 0, 1], -tau * a, rtol=1e-10)
    np.testing.assert_allclose(J[:, 1, 0], tau / a, rtol=1e-10)
    np.testing.assert_allclose(J[:, 1, 1], 0.0, atol=1e-12)
