"""
Oracle (test infrastructure, see oracle/__init__.py): the state-space-model parameterisation of a
Gauss-Markov chain, restating markovflow/state_space_model.py in NumPy.

x_0 ~ N(mu0, P0), x_{k+1} = A_k x_k + b_k + q_k, q_k ~ N(0, Q_k); parameters carry Cholesky factors.
"""
import numpy as np

from . import np_btd

_T = np_btd._T


def chol_solve(L, B):
    """(L L^T)^{-1} B."""
    return np_btd.solve_upper_from_lower(L, np_btd.solve_lower(L, B))


class StateSpaceModel:
    """state_space_model.py:35-130 (constructor argument order kept)."""

    def __init__(self, initial_mean, chol_initial_covariance, state_transitions, state_offsets,
                 chol_process_covariances):
        self.mu0 = np.asarray(initial_mean, dtype=np.float64)
        self.cholP0 = np.asarray(chol_initial_covariance, dtype=np.float64)
        self.A = np.asarray(state_transitions, dtype=np.float64)
        self.b = np.asarray(state_offsets, dtype=np.float64)
        self.cholQ = np.asarray(chol_process_covariances, dtype=np.float64)
        if self.A.shape[-3] == 0:
            # tests/unit/test_state_space_model.py:58-60 (zero transitions is an error)
            raise ValueError("StateSpaceModel needs at least one transition")

    # -- shapes -------------------------------------------------------------------------------
    @property
    def state_dim(self):
        return self.A.shape[-1]

    @property
    def num_transitions(self):
        return self.A.shape[-3]

    @property
    def batch_shape(self):
        return self.A.shape[:-3]

    # -- concatenations (state_space_model.py:180-222) -----------------------------------------
    @property
    def concatenated_state_offsets(self):
        return np.concatenate([self.mu0[..., None, :], self.b], axis=-2)

    @property
    def concatenated_cholesky_process_covariance(self):
        return np.concatenate([self.cholP0[..., None, :, :], self.cholQ], axis=-3)

    # -- precision (state_space_model.py:431-483) ----------------------------------------------
    def precision(self):
        """K^{-1} blocks: diag_k = Q_k^{-1} + A_{k+1}^T Q_{k+1}^{-1} A_{k+1}, sub_k = -Q_{k+1}^{-1} A_{k+1}."""
        inv_q_a = chol_solve(self.cholQ, self.A)
        aqa = _T(self.A) @ inv_q_a
        chols = self.concatenated_cholesky_process_covariance
        eye = np.broadcast_to(np.eye(self.state_dim), chols.shape)
        inv_q = chol_solve(chols, eye)
        diag = inv_q.copy()
        diag[..., :-1, :, :] += aqa
        return diag, -inv_q_a

    # -- marginals (state_space_model.py:232-262, 326-341) -------------------------------------
    @property
    def marginal_means(self):
        """mu = (A^{-1})^{-1} m  (forward recursion mu_{k+1} = A_k mu_k + b_k)."""
        T = self.num_transitions + 1
        out = np.empty(self.batch_shape + (T, self.state_dim))
        out[..., 0, :] = self.mu0
        for k in range(T - 1):
            out[..., k + 1, :] = (self.A[..., k, :, :] @ out[..., k, :, None])[..., 0] + self.b[..., k, :]
        return out

    @property
    def marginal_covariances(self):
        """precision.cholesky.block_diagonal_of_inverse() (state_space_model.py:262)."""
        diag, sub = self.precision()
        Ld, Ls = np_btd.cholesky(diag, sub)
        return np_btd.block_diagonal_of_inverse(Ld, Ls)

    @property
    def marginals(self):
        return self.marginal_means, self.marginal_covariances

    def subsequent_covariances(self, marginal_covariances):
        """Cov(x_{k+1}, x_k) = A_k P_k (state_space_model.py:326-341)."""
        return self.A @ marginal_covariances[..., :-1, :, :]

    def covariance_blocks(self):
        mc = self.marginal_covariances
        return mc, self.subsequent_covariances(mc)

    def log_det_precision(self):
        """state_space_model.py:343-373."""
        d0 = np.diagonal(self.cholP0, axis1=-2, axis2=-1)
        dq = np.diagonal(self.cholQ, axis1=-2, axis2=-1)
        return -(np.sum(np.log(np.square(d0)), axis=-1) + np.sum(np.log(np.square(dq)), axis=(-1, -2)))

    # -- densities -------------------------------------------------------------------------
    def log_pdf(self, states):
        """state_space_model.py:485-526: log p(x0) + sum_k log p(x_{k+1}|x_k)."""
        states = np.asarray(states, dtype=np.float64)
        d = self.state_dim

        def mvn_logpdf(x, mean, chol):
            z = np_btd.solve_lower(chol, (x - mean)[..., None])[..., 0]
            logdet = np.sum(np.log(np.abs(np.diagonal(chol, axis1=-2, axis2=-1))), axis=-1)
            return -0.5 * np.sum(z * z, axis=-1) - logdet - 0.5 * d * np.log(2 * np.pi)

        first = mvn_logpdf(states[..., 0, :], self.mu0, self.cholP0)
        cond = (self.A @ states[..., :-1, :, None])[..., 0] + self.b
        rest = mvn_logpdf(states[..., 1:, :], cond, self.cholQ)
        return first + np.sum(rest, axis=-1)

    def kl_divergence(self, other):
        """
        KL(self || other) (state_space_model.py:528-593), using only diagonal / sub-diagonal blocks of
        the covariance of self and the precision of other.
        """
        cov1 = self.marginal_covariances
        sub1 = self.subsequent_covariances(cov1)
        d2, s2 = other.precision()
        trace = np.sum(d2 * cov1, axis=(-3, -2, -1)) + 2.0 * np.sum(s2 * sub1, axis=(-3, -2, -1))
        diff = other.marginal_means - self.marginal_means
        L2d, L2s = np_btd.cholesky(d2, s2)
        ldiff = np_btd.dense_mult(L2d, L2s, diff, symmetric=False, transpose_left=True)
        maha = np.sum(ldiff * ldiff, axis=(-2, -1))
        dim = (self.num_transitions + 1) * self.state_dim
        return 0.5 * (trace + maha - dim - other.log_det_precision() + self.log_det_precision())

    def sample(self, sample_shape, rng):
        """state_space_model.py:298-324 with an explicit NumPy generator."""
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        T = self.num_transitions + 1
        eps = rng.standard_normal(tuple(sample_shape) + self.batch_shape + (T, self.state_dim))
        chols = self.concatenated_cholesky_process_covariance
        z = (chols @ eps[..., None])[..., 0] + self.concatenated_state_offsets
        out = np.empty_like(z)
        out[..., 0, :] = z[..., 0, :]
        for k in range(T - 1):
            out[..., k + 1, :] = (self.A[..., k, :, :] @ out[..., k, :, None])[..., 0] + z[..., k + 1, :]
        return out


def cholesky_or_zero(cov):
    """state_space_model.py:634-656: Cholesky, or zeros where the matrix is exactly zero."""
    cov = np.asarray(cov, dtype=np.float64)
    mask = np.all(cov == 0.0, axis=(-2, -1))
    fix = np.where(mask[..., None, None], np.eye(cov.shape[-1]), 0.0)
    chol = np.linalg.cholesky(cov + fix)
    return np.where(mask[..., None, None], 0.0, chol)


def state_space_model_from_covariances(initial_mean, initial_covariance, state_transitions,
                                       state_offsets, process_covariances):
    """state_space_model.py:613-664."""
    return StateSpaceModel(initial_mean, cholesky_or_zero(initial_covariance), state_transitions,
                           state_offsets, cholesky_or_zero(process_covariances))
