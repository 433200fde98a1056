"""CPU tests of the oracle's model layer (CVISitesSSM with a linear prior)."""
import numpy as np
import pytest

from oracle import np_kalman, np_models, np_ssm
from tests.helpers import random_ssm_params


def ou_euler_ssm(T, dt, decay, q, d=1):
    """Euler discretisation LinearDrift.to_ssm of dx = -decay x dt + sqrt(q) dB (drift.py:102-108), P0 = q/(2 decay)."""
    A = np.broadcast_to((1.0 - decay * dt) * np.eye(d), (T - 1, d, d)).copy()
    b = np.zeros((T - 1, d))
    cholQ = np.broadcast_to(np.sqrt(q * dt) * np.eye(d), (T - 1, d, d)).copy()
    cholP0 = np.sqrt(q / (2 * decay)) * np.eye(d)
    return np_ssm.StateSpaceModel(np.zeros(d), cholP0, A, b, cholQ)


@pytest.mark.parametrize("d", [1, 2])
def test_conjugate_cvi_equals_kalman(rng, d):
    """KA11: Gaussian likelihood, one data-site update with lr=1 -> ELBO == log marginal likelihood."""
    T, dt = 80, 0.01
    ssm = ou_euler_ssm(T, dt, decay=1.2, q=1.0, d=d)
    idx = np.sort(rng.choice(T, size=9, replace=False))
    y = rng.normal(size=(9, d))
    chol_R = 0.3 * np.eye(d)
    m = np_models.CVISitesSSM(ssm, np.arange(T) * dt, idx, y, np_models.MultivariateGaussianLik(chol_R))
    m.update_data_sites(1.0)
    Rinv = np.linalg.inv(chol_R @ chol_R.T)
    np.testing.assert_allclose(m.d1, y @ Rinv, rtol=1e-12)
    np.testing.assert_allclose(m.d2, np.broadcast_to(-0.5 * Rinv, m.d2.shape), rtol=1e-12)
    # the log marginal likelihood from the sparse-site Kalman filter (d=1) / a dense Gaussian (any d)
    prec_d, prec_s = ssm.precision()
    from oracle import np_btd
    K = np.linalg.inv(np_btd.to_dense(prec_d, prec_s))
    sel = (idx[:, None] * d + np.arange(d)[None, :]).reshape(-1)
    Kyy = K[np.ix_(sel, sel)] + np.kron(np.eye(len(idx)), chol_R @ chol_R.T)
    yf = y.reshape(-1)
    loglik = -0.5 * yf @ np.linalg.solve(Kyy, yf) - 0.5 * np.linalg.slogdet(Kyy)[1] - 0.5 * len(yf) * np.log(2 * np.pi)
    np.testing.assert_allclose(m.classic_elbo(), loglik, rtol=1e-6, atol=1e-6)
    if d == 1:
        sites = np_kalman.GaussianSitesNat(m.d1, m.d2)
        kf = np_kalman.KalmanFilterWithSparseSites(ssm, np.ones((T, 1, 1)), sites, T, idx, y)
        np.testing.assert_allclose(kf.log_likelihood(), loglik, rtol=1e-8)
    # a Girsanov update with lr=1 against the same linear prior drives the Girsanov sites to zero
    m.update_girsanov_sites(1.0)
    np.testing.assert_allclose(m.g1, 0.0, atol=1e-9)
    np.testing.assert_allclose(m.g2d, 0.0, atol=1e-9)
    np.testing.assert_allclose(m.classic_elbo(), loglik, rtol=1e-6, atol=1e-6)


def test_elbo_increases_with_damped_updates(rng):
    T, d = 40, 2
    ssm = np_ssm.StateSpaceModel(*random_ssm_params(rng, (), T, d))
    idx = np.arange(0, T, 5)
    y = rng.normal(size=(len(idx), d))
    m = np_models.CVISitesSSM(ssm, np.arange(T) * 0.1, idx, y, np_models.MultivariateGaussianLik(0.5 * np.eye(d)))
    prev = -np.inf
    for _ in range(4):
        m.update_data_sites(0.5)
        m.update_girsanov_sites(0.5)
        e = m.classic_elbo()
        assert e > prev - 1e-9
        prev = e


def test_cvi_dp_ou_conjugate(rng):
    """KA11 for CVISitesSDE: OU prior + Gaussian likelihood; lr = 1 updates reach the exact posterior, so
    classic_elbo equals the log marginal likelihood of the Euler-discretised SSM."""
    from oracle import np_btd, np_sde
    T, dt, decay, qv = 60, 0.01, 1.2, 1.0
    sde = np_sde.OrnsteinUhlenbeckSDE(decay, qv * np.eye(1))
    idx = np.sort(rng.choice(np.arange(1, T), size=8, replace=False))
    y = rng.normal(size=(8, 1))
    cholR = 0.1 * np.eye(1)
    P0 = qv / (2 * decay) * np.eye(1)
    m = np_models.CVISitesSDE(sde, np.arange(T) * dt, idx, y, np_models.MultivariateGaussianLik(cholR), np.zeros(1), P0)
    np.testing.assert_allclose(m.dist_p.A, 1 - decay * dt, rtol=1e-12)
    m.update_data_sites(1.0)
    m.update_girsanov_sites(1.0)
    ssm = ou_euler_ssm(T, dt, decay, qv)
    pd, ps = ssm.precision()
    K = np.linalg.inv(np_btd.to_dense(pd, ps))
    Kyy = K[np.ix_(idx, idx)] + 0.01 * np.eye(8)
    yf = y[:, 0]
    loglik = -0.5 * yf @ np.linalg.solve(Kyy, yf) - 0.5 * np.linalg.slogdet(Kyy)[1] - 0.5 * 8 * np.log(2 * np.pi)
    np.testing.assert_allclose(m.classic_elbo(), loglik, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(m.g1, 0.0, atol=1e-7)


def test_cvi_dp_double_well_elbo_improves(rng):
    from oracle import np_sde
    T, dt, d = 40, 0.02, 1
    sde = np_sde.DoubleWellSDE(np.eye(1))
    idx = np.arange(4, T, 6)
    y = np.sign(rng.normal(size=(len(idx), d))) + 0.1 * rng.normal(size=(len(idx), d))
    m = np_models.CVISitesSDE(sde, np.arange(T) * dt, idx, y, np_models.MultivariateGaussianLik(0.3 * np.eye(d)), np.zeros(d), np.eye(d))
    e0 = m.classic_elbo()
    for _ in range(3):
        m.update_data_sites(0.5)
        m.update_girsanov_sites(0.3)
    assert m.classic_elbo() > e0


def test_cvi_gp_one_step_optimum(rng):
    """KA7 (reference tests/integration/models/test_variational_cvi.py:82-141): Matern12(l=2, var=2.25), 8 points,
    Gaussian noise 1: after one update_sites with lr=1 the ELBO is the GPR log-likelihood and the sites are (y, -1/2)/sigma^2."""
    from oracle import np_kernels
    t = np.sort(rng.uniform(0, 4, size=8))
    y = np.cos(3 * t)[:, None] + 0.1 * rng.normal(size=(8, 1))
    k = np_kernels.Matern12(lengthscale=2.0, variance=2.25)
    m = np_models.CVIGaussianProcess(t, y, k, np_models.GaussianLik(1.0), learning_rate=1.0)
    m.update_sites()
    np.testing.assert_allclose(m.nat1, y / 1.0, rtol=1e-9)
    np.testing.assert_allclose(m.nat2, -0.5 * np.ones_like(m.nat2), rtol=1e-9)
    gpr = np_models.gpr_log_likelihood(t, y, k, 1.0)
    # dense check of the GPR log-likelihood itself
    Kxx = 2.25 * np.exp(-np.abs(t[:, None] - t[None, :]) / 2.0) + np.eye(8)
    dense = -0.5 * y[:, 0] @ np.linalg.solve(Kxx, y[:, 0]) - 0.5 * np.linalg.slogdet(Kxx)[1] - 4 * np.log(2 * np.pi)
    np.testing.assert_allclose(gpr, dense, rtol=1e-9)
    np.testing.assert_allclose(m.elbo(), gpr, rtol=1e-8)
    np.testing.assert_allclose(m.classic_elbo(), gpr, rtol=1e-7)


def test_sparse_cvi_and_conditionals(rng):
    """KA8 (reference tests/integration/models/test_sparse_variational_cvi.py:88-155): with z = x and a Gaussian likelihood,
    one update_sites(lr=1) gives sites (y, -1/2)/sigma^2 in the second half-blocks and classic_elbo == GPR log-likelihood;
    predict_f between the points matches dense GP regression."""
    from oracle import np_conditionals as npc, np_kernels
    N = 12
    t = np.linspace(0, 1, N)
    y = (np.cos(20 * t) + rng.normal(size=N)).reshape(-1, 1)
    k = np_kernels.Matern12(lengthscale=0.3, variance=1.5)
    m = npc.SparseCVIGaussianProcess(k, t, np_models.GaussianLik(1.0), learning_rate=1.0)
    m.update_sites(t, y)
    sd = 1
    np.testing.assert_allclose(m.nat1[:-1, sd:], y, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(m.nat2[:-1, sd:, sd:], -0.5 * np.ones((N, 1, 1)), rtol=1e-9)
    gpr = np_models.gpr_log_likelihood(t, y, k, 1.0)
    np.testing.assert_allclose(m.classic_elbo(t, y), gpr, rtol=1e-7)
    # prediction at new points against dense GP regression with the Matern-1/2 covariance
    tn = np.sort(rng.uniform(-0.2, 1.2, size=7))
    kern = lambda a, b: 1.5 * np.exp(-np.abs(a[:, None] - b[None, :]) / 0.3)
    Kxx = kern(t, t) + np.eye(N)
    mu_ref = kern(tn, t) @ np.linalg.solve(Kxx, y[:, 0])
    var_ref = 1.5 - np.einsum("ij,ji->i", kern(tn, t), np.linalg.solve(Kxx, kern(t, tn)))
    mu, var = npc.predict_f(m.dist_q, k, t, tn)
    np.testing.assert_allclose(mu[:, 0], mu_ref, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var[:, 0], var_ref, rtol=1e-6, atol=1e-8)


# ---- round 2: closed-form oracle route (what makes d = 6 checkable) pinned to the quadrature route ------------------------------
@pytest.mark.parametrize("d", [1, 2])
def test_cvi_sites_sde_closed_form_equals_quadrature(rng, d):
    """The oracle CVISitesSDE with closed_form=True (Gaussian moments of the cubic drift) follows the quadrature-route oracle
    (the reference's mvnquad formulation, H = 10 / 20) through updates, re-linearisation and ELBO: both are exact for a cubic."""
    from oracle import np_sde
    T, dt = 24, 0.02
    sde = np_sde.DoubleWellSDE(np.diag(0.6 + 0.5 * rng.random(d)))
    idx = np.arange(3, T - 1, 5)
    y = np.sign(rng.normal(size=(len(idx), d))) + 0.2 * rng.normal(size=(len(idx), d))
    lik = np_models.MultivariateGaussianLik(0.3 * np.eye(d) + 0.05 * np.tril(np.ones((d, d)), -1))
    init = (0.1 * np.ones(d), 0.5 * np.eye(d) + 0.1)
    a, b = (np_models.CVISitesSDE(sde, np.arange(T) * dt, idx, y, lik, *init, closed_form=cf) for cf in (False, True))
    np.testing.assert_allclose(b.dist_p.A, a.dist_p.A, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(b.dist_p.b, a.dist_p.b, rtol=1e-11, atol=1e-13)
    for _ in range(2):
        for m in (a, b):
            m.update_data_sites(0.5)
            m.update_girsanov_sites(0.3)
        np.testing.assert_allclose(b.KL_q_p(), a.KL_q_p(), rtol=1e-10)
        np.testing.assert_allclose(b.classic_elbo(), a.classic_elbo(), rtol=1e-10)
        for m in (a, b):
            m.relinearize()
        np.testing.assert_allclose(b.dist_p.A, a.dist_p.A, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(b.classic_elbo(), a.classic_elbo(), rtol=1e-10)


@pytest.mark.parametrize("d", [1, 2])
def test_vdp_closed_form_equals_quadrature(rng, d):
    """The same for the VDP oracle: E_sde, E f, E f' from Gaussian moments vs the reference's 20^d / 10^d-point grids."""
    from oracle import np_sde
    T, dt = 30, 0.01
    sde = np_sde.DoubleWellSDE(np.diag(0.8 + 0.4 * rng.random(d)))
    idx = np.arange(4, T - 1, 6)
    y = np.sign(rng.normal(size=(len(idx), d))) + 0.1 * rng.normal(size=(len(idx), d))
    lik = np_models.MultivariateGaussianLik(0.5 * np.eye(d))
    init = (np.zeros(d), 0.5 * np.eye(d))
    a, b = (np_models.VariationalMarkovGP(idx, y, sde, np.arange(T) * dt, lik, *init, closed_form=cf) for cf in (False, True))
    for it in range(3):
        for m in (a, b):
            mm, S = m.forward_pass()
            m.update_lagrange(mm, S)
            m.update_param(mm, S, 0.05)
        np.testing.assert_allclose(b.A, a.A, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(b.b, a.b, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(b.elbo(), a.elbo(), rtol=1e-10)


def config1_closed_form_log_marginal(time_grid, obs_index, y, decay, q, sigma):
    """
    KA11 target for BASELINE config 1, independent of every block-tri-diagonal routine: the log marginal likelihood of the
    observations under the Euler-discretised OU chain x_{t+1} = a x_t + N(0, q dt), a = 1 - decay dt, x_0 ~ N(0, q / (2 decay))
    (cvi_dp.py:60-65), observed with noise sigma at the grid indices `obs_index`.  Cov(x_s, x_t) = a^{|t-s|} P_{min(s,t)} with
    P_t = a^2 P_{t-1} + q dt, so K_yy is written down directly and the answer is one dense 32 x 32 Gaussian.
    """
    dt = float(time_grid[1] - time_grid[0])
    a = 1.0 - decay * dt
    P = np.empty(len(time_grid))
    P[0] = q / (2.0 * decay)
    for t in range(1, len(P)):
        P[t] = a * a * P[t - 1] + q * dt
    i = np.asarray(obs_index)
    lo = np.minimum(i[:, None], i[None, :])
    Kyy = a ** np.abs(i[:, None] - i[None, :]) * P[lo] + sigma ** 2 * np.eye(len(i))
    yf = np.asarray(y).reshape(-1)
    return -0.5 * yf @ np.linalg.solve(Kyy, yf) - 0.5 * np.linalg.slogdet(Kyy)[1] - 0.5 * len(yf) * np.log(2 * np.pi)


def load_config1():
    """BASELINE config 1 exactly as SURVEY 8d / docs/diffusion_processes/README.md:43 run it: the reference's shipped
    data/linear/15/0.npz (T = 1001, dt = 0.01, 32 observations, sigma = 0.1, generated with decay 0.5) under an OU prior with
    decay 1.2, q = Q, p(x0) = N(0, Q / 2.4)."""
    import os
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "linear_15_0.npz"))
    tg, og = z["time_grid"], z["observation_grid"]
    idx = np.searchsorted(tg, og)
    assert np.array_equal(tg[idx], og) and tg.shape == (1001,) and og.shape == (32,)
    return dict(time_grid=tg, obs_index=idx, y=z["observations"], Q=float(z["Q"]), sigma=float(z["sigma"]), decay=1.2,
                test_grid=z["test_grid"], test_y=z["test_observations"])


def test_config1_recipe_oracle():
    """Config 1 with the reference's own data and schedule (configs/cvi_linear_process.yaml: both learning rates 1, one site
    iteration): the oracle's ELBO equals the closed-form log marginal likelihood (KA11)."""
    from oracle import np_sde
    c = load_config1()
    sde = np_sde.OrnsteinUhlenbeckSDE(c["decay"], c["Q"] * np.eye(1))
    P0 = c["Q"] / (2 * c["decay"]) * np.eye(1)
    m = np_models.CVISitesSDE(sde, c["time_grid"], c["obs_index"], c["y"], np_models.MultivariateGaussianLik(c["sigma"] * np.eye(1)),
                              np.zeros(1), P0)
    m.update_data_sites(1.0)
    m.update_girsanov_sites(1.0)
    target = config1_closed_form_log_marginal(c["time_grid"], c["obs_index"], c["y"], c["decay"], c["Q"], c["sigma"])
    np.testing.assert_allclose(m.classic_elbo(), target, rtol=1e-6)
    np.testing.assert_allclose(m.d1, c["y"] / c["sigma"] ** 2, rtol=1e-10)
    np.testing.assert_allclose(m.g1, 0.0, atol=1e-5)
