"""
Oracle (test infrastructure, see oracle/__init__.py): the sparse-precision ("SpInGP") Kalman
filter of markovflow/kalman_filter.py restated in NumPy.
"""
import numpy as np

from . import np_btd
from .np_ssm import StateSpaceModel, chol_solve

_T = np_btd._T


class GaussianSitesNat:
    """kalman_filter.py:382-436."""

    def __init__(self, nat1, nat2, log_norm=None):
        self.nat1 = np.asarray(nat1, dtype=np.float64)
        self.nat2 = np.asarray(nat2, dtype=np.float64)
        self.log_norm = log_norm

    @property
    def means(self):
        return -0.5 * (np.linalg.inv(self.nat2) @ self.nat1[..., None])[..., 0]

    @property
    def precisions(self):
        return -2.0 * self.nat2


class _Base:
    """kalman_filter.py:32-271 (BaseKalmanFilter).  H: [..., T, o, d]."""

    def __init__(self, ssm: StateSpaceModel, emission_matrix):
        self.prior_ssm = ssm
        self.H = np.asarray(emission_matrix, dtype=np.float64)

    # to be provided: r_inv ([o,o] or [..., T, o, o]), observations [..., T, o], log_det_obs_precision
    def k_inv_post(self):
        """kalman_filter.py:86-101: prior precision + H^T R^{-1} H on the block diagonal."""
        d, s = self.prior_ssm.precision()
        hrh = np.einsum("...ji,...jk,...kl->...il", self.H, self.r_inv, self.H)
        return d + hrh, s

    def back_project(self, y):
        """kalman_filter.py:257-271: (G^T Sigma^{-1}) y."""
        bp = np.einsum("...ij,...ki->...kj", self.H, self.r_inv)
        return np.einsum("...ij,...i->...j", bp, y)

    def log_likelihood(self):
        """kalman_filter.py:184-255, summed over the batch."""
        d, s = self.k_inv_post()
        Ld, Ls = np_btd.cholesky(d, s)
        T = self.prior_ssm.num_transitions + 1
        o = self.H.shape[-2]
        marginal = np.einsum("...ij,...j->...i", self.H, self.prior_ssm.marginal_means)
        disp = self.observations - marginal
        cst = -0.5 * np.log(2 * np.pi) * o * self.num_data
        term1 = -0.5 * np.sum(np.einsum("...op,...p,...o->...o", self.r_inv_data, self.disp_data(disp), self.disp_data(disp)),
                              axis=(-1, -2))
        obs_proj = self.back_project(disp)
        term2 = 0.5 * np.sum(np.square(np_btd.solve(Ld, Ls, obs_proj)), axis=(-1, -2))
        term3 = 0.5 * self.prior_ssm.log_det_precision() - np_btd.abs_log_det(Ld) \
            + 0.5 * self.log_det_observation_precision
        return np.sum(cst + term1 + term2 + term3)

    # defaults for the dense variants
    @property
    def num_data(self):
        return self.prior_ssm.num_transitions + 1

    @property
    def r_inv_data(self):
        return self.r_inv

    def disp_data(self, disp):
        return disp

    def posterior_state_space_model(self):
        """kalman_filter.py:109-182."""
        d, s = self.k_inv_post()
        u_s, chol_d = np_btd.upper_diagonal_lower(d, s)
        eye = np.broadcast_to(np.eye(d.shape[-1]), d.shape)
        obs_proj = self.back_project(self.observations)
        dp, sp = self.prior_ssm.precision()
        k_inv_mu = np_btd.dense_mult(dp, sp, self.prior_ssm.marginal_means, symmetric=True)
        # a_inv_post = (identities, u_s); chol_q_inv_post = (chol_d, None)
        tmp = np_btd.solve(eye, u_s, obs_proj + k_inv_mu, transpose_left=True)
        tmp = np_btd.solve(chol_d, None, tmp, transpose_left=False)
        m_post = np_btd.solve(chol_d, None, tmp, transpose_left=True)
        qs = np.linalg.cholesky(chol_solve(chol_d, eye))
        return StateSpaceModel(m_post[..., 0, :], qs[..., 0, :, :], -u_s, m_post[..., 1:, :], qs[..., 1:, :, :])


class KalmanFilter(_Base):
    """kalman_filter.py:275-345: one observation-noise Cholesky [o,o] shared by all steps."""

    def __init__(self, ssm, emission_matrix, observations, chol_obs_covariance):
        super().__init__(ssm, emission_matrix)
        self.observations = np.asarray(observations, dtype=np.float64)
        c = np.asarray(chol_obs_covariance, dtype=np.float64)
        self.r_inv = chol_solve(c, np.eye(c.shape[-1]))

    @property
    def log_det_observation_precision(self):
        return self.num_data * np.linalg.slogdet(self.r_inv)[1]


class KalmanFilterWithSites(_Base):
    """kalman_filter.py:440-500: per-step Gaussian sites in natural form."""

    def __init__(self, ssm, emission_matrix, sites: GaussianSitesNat):
        super().__init__(ssm, emission_matrix)
        self.sites = sites

    @property
    def r_inv(self):
        return self.sites.precisions

    @property
    def observations(self):
        return self.sites.means

    @property
    def log_det_observation_precision(self):
        return np.sum(np.linalg.slogdet(self.r_inv)[1], axis=-1)


class KalmanFilterWithSparseSites(_Base):
    """kalman_filter.py:504-639: sites only at `observations_index` of a finer grid (o = 1, no batch)."""

    def __init__(self, ssm, emission_matrix, sites: GaussianSitesNat, num_grid_points, observations_index,
                 observations):
        super().__init__(ssm, emission_matrix)
        self.sites = sites
        self.idx = np.asarray(observations_index).reshape(-1)
        obs = np.asarray(observations, dtype=np.float64)
        if obs.ndim == 3:
            if obs.shape[0] != 1:
                raise ValueError("KalmanFilterWithSparseSites doesn't support batches")
            obs = obs[0]
        self.sparse_observations = obs
        self.grid = num_grid_points

    def _scatter(self, vals, tail):
        out = np.zeros((self.grid,) + tail)
        np.add.at(out, self.idx, vals)  # tf.scatter_nd accumulates duplicates
        return out

    @property
    def r_inv(self):
        return self._scatter(self.sites.precisions, (1, 1))

    @property
    def r_inv_data(self):
        return self.sites.precisions

    @property
    def observations(self):
        return self._scatter(self.sparse_observations, (1,))

    @property
    def num_data(self):
        return self.idx.shape[0]

    def disp_data(self, disp):
        # kalman_filter.py:608: sparse observations minus gathered marginal; disp on the grid is
        # obs_grid - marginal, so gather and restore the sparse observation exactly.
        d = np.squeeze(disp, axis=0) if disp.ndim == 3 else disp
        marginal = self.observations - d
        return self.sparse_observations - marginal[self.idx]

    @property
    def log_det_observation_precision(self):
        return np.sum(np.linalg.slogdet(self.r_inv_data)[1], axis=-1)
