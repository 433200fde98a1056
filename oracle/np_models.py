"""
Oracle (test infrastructure, see oracle/__init__.py): the site-based CVI models of
markovflow/models/variational_cvi_sde.py (`CVISitesSSM`) and markovflow/models/variational_cvi.py
(`CVIGaussianProcess`), single trajectory, NumPy, following the reference's own op sequence
(every property rebuilt from the sites).

Parity note: the reference has no test for CVISitesSSM / CVISitesSDE; the only pin is the conjugate case
KA11 of SURVEY.md 8c (OU prior, Gaussian likelihood, lr = 1 => ELBO equals the Kalman log marginal
likelihood), which tests/test_oracle_models.py checks against the reference's NumPy Kalman filter.
"""
import numpy as np

from . import np_btd, np_kalman, np_transforms
from .np_ssm import StateSpaceModel, chol_solve


class MultivariateGaussianLik:
    """markovflow/likelihoods/multivariate_gaussian.py:80-115."""

    def __init__(self, chol_covariance):
        self.chol = np.asarray(chol_covariance, dtype=np.float64)
        self.d = self.chol.shape[-1]
        self.inv_cov = chol_solve(self.chol, np.eye(self.d))

    def variational_expectations(self, mu, cov, y):
        z = np.linalg.solve(self.chol, (y - mu)[..., None])[..., 0]
        logp = -0.5 * np.sum(z * z, -1) - np.sum(np.log(np.diag(self.chol))) - 0.5 * self.d * np.log(2 * np.pi)
        return -0.5 * np.sum(self.inv_cov * cov, axis=(-1, -2)) + logp

    def grads_expectation(self, mu, cov, y):
        """d sum VE / d(mu, S) mapped to d / d(mu, S + mu mu^T) (variational_cvi.py:448-462)."""
        dmu = (self.inv_cov @ (y - mu)[..., None])[..., 0]
        dS = np.broadcast_to(-0.5 * self.inv_cov, cov.shape)
        return dmu - 2.0 * (dS @ mu[..., None])[..., 0], dS


class CVISitesSSM:
    """variational_cvi_sde.py:49-366 with a linear prior SSM; KL by the exact Gauss-Markov formula."""

    def __init__(self, prior_ssm: StateSpaceModel, time_grid, obs_index, observations, likelihood):
        self.dist_p = prior_ssm
        self.time_grid = np.asarray(time_grid)
        self.obs_index = np.asarray(obs_index)
        self.y = np.asarray(observations, dtype=np.float64)
        self.lik = likelihood
        T, d = self.time_grid.shape[0], self.y.shape[-1]
        self.T, self.d = T, d
        self.g1 = np.zeros((T, d))
        self.g2d = -1e-10 * np.ones((T, d, d))
        self.g2s = -1e-10 * np.ones((T - 1, d, d))
        self.d1 = np.zeros((self.y.shape[0], d))
        self.d2 = 1e-10 * np.broadcast_to(np.eye(d), (self.y.shape[0], d, d)).copy()
        self.fx_mus = np.zeros((T, d))
        self.fx_covs = np.broadcast_to(np.eye(d), (T, d, d)).copy()

    def _scatter(self, vals, shape):
        out = np.zeros(shape)
        np.add.at(out, self.obs_index, vals)
        return out

    def full_sites(self):
        p1, pd, ps = np_transforms.ssm_to_naturals(self.dist_p)
        return (p1 + self.g1 + self._scatter(self.d1, self.g1.shape),
                pd + self.g2d + self._scatter(self.d2, self.g2d.shape), ps + self.g2s)

    @property
    def dist_q(self):
        return np_transforms.ssm_from_params(np_transforms.naturals_to_ssm_params(*self.full_sites()))

    def update_data_sites(self, lr):
        g1, g2 = self.lik.grads_expectation(self.fx_mus[self.obs_index], self.fx_covs[self.obs_index], self.y)
        self.d1 = (1 - lr) * self.d1 + lr * g1
        self.d2 = (1 - lr) * self.d2 + lr * g2
        self.fx_mus, self.fx_covs = self.dist_q.marginals

    def grad_kl_wrt_exp_param(self):
        q = self.full_sites()
        p = np_transforms.ssm_to_naturals(self.dist_p)
        return tuple(a - b for a, b in zip(q, p))

    def update_girsanov_sites(self, lr):
        gk = self.grad_kl_wrt_exp_param()
        self.g1 = self.g1 + lr * (self._scatter(self.d1, self.g1.shape) - gk[0])
        self.g2d = self.g2d + lr * (self._scatter(self.d2, self.g2d.shape) - gk[1])
        self.g2s = self.g2s - lr * gk[2]
        self.fx_mus, self.fx_covs = self.dist_q.marginals

    def variational_expectation(self):
        mu, cov = self.dist_q.marginals
        return np.sum(self.lik.variational_expectations(mu[self.obs_index], cov[self.obs_index], self.y))

    def KL_q_p(self):
        return self.dist_q.kl_divergence(self.dist_p)

    def classic_elbo(self):
        return self.variational_expectation() - self.KL_q_p()


class CVISitesSDE(CVISitesSSM):
    """
    CVI-DP, variational_cvi_sde.py:368-518, single trajectory.  KL_q_p follows the reference's quadrature route
    (SSM_KL_along_Gaussian_path, H = 20); grad_kl_wrt_exp_param uses the closed form of oracle/np_sde.py, which
    tests/test_oracle_sde.py pins against finite differences of that same quadrature (the reference uses a GradientTape).
    PARITY UNPINNED: the reference has no test for this class; the only anchor is the conjugate
    OU case KA11.
    """

    def __init__(self, prior_sde, time_grid, obs_index, observations, likelihood, init_mu, init_cov, stabilize_ssm=True,
                 clip=(-1.0, 1.0), closed_form=False, exact_q=False):
        """closed_form: every expectation under q (E f, E f', the Girsanov KL) from the cubic drift's Gaussian moments instead of
        the reference's H^d-point Gauss-Hermite grids -- the same numbers (both are exact for a cubic; pinned against each other at
        d <= 2 in tests/test_oracle_sde.py and tests/test_oracle_models.py), tractable at d = 6."""
        from . import np_sde
        self._np_sde = np_sde
        self.closed_form = bool(closed_form)
        self.exact_q = bool(exact_q)          # full diffusion matrices: np_sde.linearize_sde(exact_q=True)
        self.sde = prior_sde
        self.init_mu, self.init_cov = np.asarray(init_mu, dtype=np.float64), np.asarray(init_cov, dtype=np.float64)
        self.stabilize_ssm, self.clip = stabilize_ssm, clip
        self.dt = float(time_grid[1] - time_grid[0])
        super().__init__(None, time_grid, obs_index, observations, likelihood)
        self.set_linearized_prior()

    def set_linearized_prior(self):
        lin = self._np_sde.linearize_sde(self.sde, self.time_grid, self.fx_mus[1:], self.fx_covs[1:], self.init_mu, self.init_cov,
                                         closed_form=self.closed_form, exact_q=self.exact_q)
        self.dist_p_linearized = lin
        if self.stabilize_ssm:
            self.dist_p = StateSpaceModel(lin.mu0, lin.cholP0, np.clip(lin.A, *self.clip), np.clip(lin.b, *self.clip), lin.cholQ)
        else:
            self.dist_p = lin

    def relinearize(self):
        """dist_p_last = dist_p; set_linearized_prior(); tranform_girsanov_sites (cvi_dp_trainer.py:127-134, sde_utils.py:550-568)."""
        old = np_transforms.ssm_to_naturals(self.dist_p)
        self.set_linearized_prior()
        new = np_transforms.ssm_to_naturals(self.dist_p)
        self.g1 = self.g1 + old[0] - new[0]
        self.g2d = self.g2d + old[1] - new[1]
        self.g2s = self.g2s + old[2] - new[2]

    def KL_q_p(self):
        q = self.dist_q
        mu, cov = q.marginals
        if self.closed_form:
            alpha, beta = self.sde.cubic(self.dt)
            return self._np_sde.sde_ssm_kl_closed_form(mu, cov, q.subsequent_covariances(cov), alpha, beta, np.diag(self.sde.q),
                                                       self.dt, self.dist_p.mu0, self.dist_p.cholP0 @ self.dist_p.cholP0.T,
                                                       want_grads=False)
        Qq = q.cholQ @ np.swapaxes(q.cholQ, -1, -2)
        N, D = q.b.shape
        Qp = np.broadcast_to(self.dt * self.sde.q, (N, D, D))
        f_q = lambda x: (q.A[None] @ x[..., None])[..., 0] + q.b[None]
        f_p = lambda x: x + self.dt * self.sde.drift(x)
        kl = self._np_sde.ssm_kl_along_gaussian_path(f_q, f_p, Qq, Qp, mu, cov)
        p0 = self.dist_p.cholP0 @ self.dist_p.cholP0.T
        return kl + self._np_sde.gauss_kl(q.mu0, q.cholP0 @ q.cholP0.T, self.dist_p.mu0, p0)

    def grad_kl_wrt_exp_param(self):
        q = self.dist_q
        mu, cov = q.marginals
        qm = np.asarray(self.sde.q)
        if not hasattr(self.sde, "cubic") or np.abs(qm - np.diag(np.diag(qm))).max() > 0.0:
            # non-polynomial / coupled drifts, full diffusion matrices: fourth-order difference quotients of the reference's quadrature
            # KL stand in for its GradientTape
            sub = q.subsequent_covariances(cov)
            eta_d = cov + mu[..., :, None] * mu[..., None, :]
            eta_s = sub + mu[1:, :, None] * mu[:-1, None, :]
            return self._np_sde.sde_ssm_kl_grads_fd(mu, eta_d, eta_s, self.sde, self.dt, self.init_mu, self.init_cov, eps=2e-4,
                                                    richardson=True)
        alpha, beta = self.sde.cubic(self.dt)
        _, grads = self._np_sde.sde_ssm_kl_closed_form(mu, cov, q.subsequent_covariances(cov), alpha, beta, np.diag(self.sde.q),
                                                       self.dt, self.init_mu, self.init_cov)
        return grads


class GaussianLik:
    """gpflow.likelihoods.Gaussian (third-party, GPflow 2.2.1): scalar Gaussian with variance `variance`."""

    def __init__(self, variance):
        self.variance = float(variance)

    def variational_expectations(self, mu, var, y):
        v = self.variance
        return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(v) - 0.5 * ((y - mu) ** 2 + var) / v

    def grads_expectation(self, mu, var, y):
        v = self.variance
        dmu, dvar = (y - mu) / v, -0.5 / v * np.ones_like(var)
        return dmu - 2.0 * dvar * mu, dvar


class CVIGaussianProcess:
    """variational_cvi.py:225-421, single batch element: sites on f = H s, kernel prior."""

    def __init__(self, time_points, observations, kernel, likelihood, learning_rate=0.1):
        self.t, self.y = np.asarray(time_points, dtype=np.float64), np.asarray(observations, dtype=np.float64)
        self.kernel, self.lik, self.lr = kernel, likelihood, learning_rate
        self.nat1 = np.zeros_like(self.y)
        self.nat2 = -1e-10 * np.ones(self.y.shape + (1,))

    @property
    def dist_p(self):
        return self.kernel.state_space_model(self.t)

    @property
    def dist_q(self):
        pd, ps = self.dist_p.precision()
        H = self.kernel.emission_matrix(self.t)
        bp1 = np.sum(H * self.nat1[..., None], axis=-2)
        bp2 = np.sum(self.nat2[..., 0][..., None, None] * H[..., None] * H[..., None, :], axis=-3)
        return np_transforms.ssm_from_params(np_transforms.naturals_to_ssm_params(bp1, -0.5 * pd + bp2, -ps))

    def predict_f(self):
        mu, cov = self.dist_q.marginals
        H = self.kernel.emission_matrix(self.t)
        return np.einsum("...ij,...j->...i", H, mu), np.einsum("...ij,...jk,...ik->...i", H, cov, H)

    def update_sites(self):
        mu, var = self.predict_f()
        g1, g2 = self.lik.grads_expectation(mu, var, self.y)
        self.nat1 = (1 - self.lr) * self.nat1 + self.lr * g1
        self.nat2 = (1 - self.lr) * self.nat2 + self.lr * g2[..., None]

    def elbo(self):
        sites = np_kalman.GaussianSitesNat(self.nat1, self.nat2)
        return np_kalman.KalmanFilterWithSites(self.dist_p, self.kernel.emission_matrix(self.t), sites).log_likelihood()

    def classic_elbo(self):
        mu, var = self.predict_f()
        return np.sum(self.lik.variational_expectations(mu, var, self.y)) - np.sum(self.dist_q.kl_divergence(self.dist_p))


def gpr_log_likelihood(time_points, observations, kernel, noise_variance):
    """GaussianProcessRegression.log_likelihood (gaussian_process_regression.py:152) through the Kalman filter."""
    t = np.asarray(time_points, dtype=np.float64)
    kf = np_kalman.KalmanFilter(kernel.state_space_model(t), kernel.emission_matrix(t), observations,
                                np.sqrt(noise_variance) * np.eye(1))
    return kf.log_likelihood()


class VariationalMarkovGP:
    """
    VDP (Archambeau et al. 2007), restating markovflow/models/vi_sde.py:63-482 for a single trajectory.
    E_sde follows the reference's quadrature (squared_drift_difference_along_Gaussian_path); its gradients use the
    closed form of oracle/np_sde.py (pinned to finite differences of that quadrature in tests/test_oracle_sde.py) in
    place of the reference's GradientTape.  The Lagrange sweep reproduces the reference's Python loop exactly,
    including `psi @ A + psi @ A`, the t-1 write and the untouched last row (vi_sde.py:337-347).
    PARITY UNPINNED: the reference has no test for this class.
    """

    CLIP_MIN, CLIP_MAX = -5000.0, 5000.0          # vi_sde.py:59-60

    def __init__(self, obs_index, observations, sde, grid, likelihood, init_mu, init_cov, stabilize_system=False, closed_form=False):
        """closed_form: E_sde, E f and E f' from the cubic drift's Gaussian moments instead of the 20^d / 10^d-point grids (the same
        numbers, see CVISitesSDE; tractable at d = 6)."""
        from . import np_sde
        self._np_sde = np_sde
        self.closed_form = bool(closed_form)
        self.stabilize_system = stabilize_system
        self.obs_index, self.y = np.asarray(obs_index), np.asarray(observations, dtype=np.float64)
        self.sde, self.grid, self.lik = sde, np.asarray(grid, dtype=np.float64), likelihood
        self.d = sde.state_dim
        self.N = self.grid.shape[0] - 1
        self.dt = float(self.grid[1] - self.grid[0])
        self.A = np.zeros((self.N, self.d, self.d))
        self.b = np.zeros((self.N, self.d))
        self.p0_mu, self.p0_cov = np.asarray(init_mu, dtype=np.float64), np.asarray(init_cov, dtype=np.float64)
        self.q0_mu, self.q0_chol = self.p0_mu.copy(), np.linalg.cholesky(self.p0_cov)
        self.lam = np.zeros((self.N, self.d))
        self.psi = 1e-10 * np.broadcast_to(np.eye(self.d), (self.N, self.d, self.d)).copy()

    def forward_pass(self):
        """vi_sde.py:171-204: marginals of the SSM of the linear drift -A x + b."""
        q = np.broadcast_to(self.sde.q, (self.N, self.d, self.d))
        ssm = self._np_sde.linear_drift_to_ssm(-self.A, self.b, q, self.grid, self.q0_mu, self.q0_chol)
        if self.stabilize_system:       # vi_sde.py:186-200
            A = np.clip(np.where(np.isnan(ssm.A), 1e-8, ssm.A), -1.0, 1.0)
            b = np.clip(np.where(np.isnan(ssm.b), 1e-8, ssm.b), -1.0, 1.0)
            ssm = StateSpaceModel(ssm.mu0, ssm.cholP0, A, b, ssm.cholQ)
        return ssm.marginals

    def E_sde(self, m=None, S=None):
        if m is None:
            m, S = self.forward_pass()
            m, S = m[:-1], S[:-1]
        if self.closed_form:
            af, bf = self._np_sde.drift_cubic(self.sde)
            return self._np_sde.e_sde_closed_form(af, bf, np.diag(self.sde.q), -self.A, self.b, m, S, self.dt, want_grads=False)
        return self._np_sde.squared_drift_difference_along_gaussian_path(self.sde, -self.A, self.b, m, S, self.dt)

    def _general(self):
        """A drift the per-dimension cubic closed forms do not cover (coupled / network drifts) or a full diffusion matrix."""
        q = np.asarray(self.sde.q)
        cubic = isinstance(self.sde, (self._np_sde.OrnsteinUhlenbeckSDE, self._np_sde.DoubleWellSDE))
        return not cubic or np.abs(q - np.diag(np.diag(q))).max() > 0.0

    def _grad_E_sde(self, m, S):
        if self._general():
            # fourth-order difference quotients of the reference's quadrature stand in for its GradientTape (vi_sde.py:205-243)
            if hasattr(self.sde, "hessian_drift"):
                # a polynomial drift: exact through the Gaussian identities (np_sde.e_sde_grads_stein)
                return self._np_sde.e_sde_grads_stein(self.sde, -self.A, self.b, m[:-1], S[:-1])
            return self._np_sde.e_sde_grads_fd(self.sde, -self.A, self.b, m[:-1], S[:-1])
        af, bf = self._np_sde.drift_cubic(self.sde)
        _, dm, dS = self._np_sde.e_sde_closed_form(af, bf, np.diag(self.sde.q), -self.A, self.b, m[:-1], S[:-1], self.dt)
        return dm / self.dt, dS / self.dt

    def _jump_conditions(self, m, S):
        """Gradients of the variational expectations wrt (m, S) at the observation times, scattered on the grid (vi_sde.py:262-287)."""
        mu, cov = m[self.obs_index], S[self.obs_index]
        dmu = (self.lik.inv_cov @ (self.y - mu)[..., None])[..., 0]
        dS = np.broadcast_to(-0.5 * self.lik.inv_cov, cov.shape)
        d_obs_m, d_obs_S = np.zeros_like(m), np.zeros_like(S)
        np.add.at(d_obs_m, self.obs_index, dmu)
        np.add.at(d_obs_S, self.obs_index, dS)
        return d_obs_m, d_obs_S

    def update_lagrange(self, m, S):
        dEdm, dEdS = self._grad_E_sde(m, S)
        d_obs_m, d_obs_S = self._jump_conditions(m, S)
        if self.stabilize_system:       # vi_sde.py:312-323
            fix = lambda x: np.clip(np.where(np.isnan(x), 1e-8, x), self.CLIP_MIN, self.CLIP_MAX)
            dEdm, dEdS, d_obs_m, d_obs_S = fix(dEdm), fix(dEdS), fix(d_obs_m), fix(d_obs_S)
        self.lam = np.zeros_like(self.lam)
        self.psi = 1e-10 * np.broadcast_to(np.eye(self.d), (self.N, self.d, self.d)).copy()
        for t in range(self.N - 1, 0, -1):
            d_psi = self.psi[t] @ self.A[t] + self.psi[t] @ self.A[t] - dEdS[t]
            d_lam = self.A[t] @ self.lam[t] - dEdm[t]
            self.psi[t - 1] = self.psi[t] - self.dt * d_psi - d_obs_S[t]
            self.lam[t - 1] = self.lam[t] - self.dt * d_lam - d_obs_m[t]

    def update_param(self, m, S, lr):
        m, S = m[:-1], S[:-1]
        if self.stabilize_system:       # vi_sde.py:393-397
            self.psi = np.clip(np.where(np.isnan(self.psi), 1e-8, self.psi), self.CLIP_MIN, self.CLIP_MAX)
            self.lam = np.clip(np.where(np.isnan(self.lam), 1e-8, self.lam), self.CLIP_MIN, self.CLIP_MAX)
        q = self.sde.q
        if self.closed_form:
            Ef, Jf = self._np_sde.expected_drift_closed_form(self.sde, m, S)
            Egrad = -Jf
        else:
            Ef = self.sde.expected_drift(m[None], S[None])[0]
            if not hasattr(self.sde, "jacobian_drift"):
                Egrad = -self.sde.expected_gradient_drift(m[None], S[None])[0]
        if hasattr(self.sde, "jacobian_drift"):
            # drifts that couple the dimensions: the full expected Jacobian (sde.py:484-518)
            Egrad = -self._np_sde.mvnquad(lambda x: self.sde.jacobian_drift(x), m, S, 10, self.d, (self.d, self.d))
            A_tilde = Egrad + 2.0 * q[None] @ self.psi
        else:
            A_tilde = Egrad[:, :, None] * np.eye(self.d) + 2.0 * q[None] @ self.psi
        b_tilde = Ef + (A_tilde @ m[..., None])[..., 0] - (q[None] @ self.lam[..., None])[..., 0]
        self.A = (1 - lr) * self.A + lr * A_tilde
        self.b = (1 - lr) * self.b + lr * b_tilde

    def update_initial_statistics(self, lr):
        """vi_sde.py:241-260."""
        mean = self.p0_mu - self.p0_cov @ self.lam[0]
        cov = np.linalg.inv(np.linalg.inv(self.p0_cov) + 2.0 * self.psi[0])
        q0_cov = self.q0_chol @ self.q0_chol.T
        self.q0_mu = (1 - lr) * self.q0_mu + lr * mean
        self.q0_chol = np.linalg.cholesky((1 - lr) * q0_cov + lr * cov)

    def KL_initial_state(self):
        return self._np_sde.gauss_kl(self.q0_mu, self.q0_chol @ self.q0_chol.T, self.p0_mu, self.p0_cov)

    def elbo(self, m=None, S=None):
        """vi_sde.py:436-455: E_obs - E_sde - KL(q0 || p0); note E_sde() is re-evaluated from the current parameters."""
        if m is None:
            m, S = self.forward_pass()
        E_obs = np.sum(self.lik.variational_expectations(m[self.obs_index], S[self.obs_index], self.y))
        return E_obs - self.E_sde() - self.KL_initial_state()


def calculate_nlpd(m, S, chol_R, time_grid, test_times, test_obs):
    """Negative log predictive density at held-out grid points (docs/diffusion_processes/exp_dp_utils.py:189-206 on the output of
    likelihood.predict_mean_and_var, cvi_dp_trainer.py:189-196 / multivariate_gaussian.py: y* ~ N(m, S + R)): minus the MEAN over the
    test points of log N(y_i; m_i, S_i + R).  m [T, d], S [T, d, d]."""
    idx = np.array([int(np.argmin(np.abs(time_grid - t))) for t in test_times])
    R = chol_R @ chol_R.T
    tot = 0.0
    for i, y in zip(idx, test_obs):
        C = S[i] + R
        L = np.linalg.cholesky(C)
        z = np.linalg.solve(L, y - m[i])
        tot += -0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * len(y) * np.log(2 * np.pi)
    return -tot / len(idx)


def calculate_rmse(m, time_grid, test_times, test_obs):
    """Root of the mean squared error over all entries at the held-out grid points (exp_dp_utils.py:209-223)."""
    idx = np.array([int(np.argmin(np.abs(time_grid - t))) for t in test_times])
    return float(np.sqrt(np.mean((m[idx] - test_obs) ** 2)))

