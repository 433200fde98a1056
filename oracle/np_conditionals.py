"""
Oracle (test infrastructure, see oracle/__init__.py): markovflow/conditionals.py and posterior.py `ConditionalProcess`
in NumPy, plus the sparse CVI model of markovflow/models/sparse_variational_cvi.py (single batch element).
"""
import numpy as np

from . import np_transforms
from .np_ssm import chol_solve

APPROX_INF = 1e10
_T = lambda x: np.swapaxes(x, -1, -2)


def cond_stats_from_transitions(A_mt, Q_mt, A_tp, Q_tp):
    """conditionals.py:111-204."""
    AQ = A_tp @ Q_mt
    Q_mp = Q_tp + A_tp @ _T(AQ)
    E = _T(chol_solve(np.linalg.cholesky(Q_mp), AQ))
    D = A_mt - E @ A_tp @ A_mt
    T = Q_mt - _T(AQ) @ chol_solve(np.linalg.cholesky(Q_mp), AQ)
    return D, E, T


def conditional_statistics(new_t, train_t, kernel):
    """conditionals.py:207-256."""
    idx = np.searchsorted(train_t, new_t, side="left")
    aug = np.concatenate([[-APPROX_INF], train_t, [APPROX_INF]])
    A_mt, Q_mt = kernel.transition_statistics(new_t - aug[idx])
    A_tp, Q_tp = kernel.transition_statistics(aug[idx + 1] - new_t)
    D, E, T = cond_stats_from_transitions(A_mt, Q_mt, A_tp, Q_tp)
    return np.concatenate([D, E], axis=-1), T, idx


def pairwise_marginals(ssm, init_mean, init_cov):
    """conditionals.py:424-470."""
    m, c = ssm.marginals
    s = ssm.subsequent_covariances(c)
    em = np.concatenate([init_mean[None], m, init_mean[None]], axis=0)
    jm = np.concatenate([em[:-1], em[1:]], axis=-1)
    ec = np.concatenate([init_cov[None], c, init_cov[None]], axis=0)
    es = np.concatenate([np.zeros_like(init_cov)[None], s, np.zeros_like(init_cov)[None]], axis=0)
    top = np.concatenate([ec[:-1], _T(es)], axis=-1)
    bot = np.concatenate([es, ec[1:]], axis=-1)
    return jm, np.concatenate([top, bot], axis=-2)


def predict_state(ssm, kernel, cond_t, new_t):
    """ConditionalProcess.predict_state (posterior.py:207-229)."""
    Pinf = kernel.steady_state_covariance() + kernel.jitter * np.eye(kernel.state_dim)
    jm, jc = pairwise_marginals(ssm, kernel.state_mean, Pinf)
    P, T, idx = conditional_statistics(new_t, cond_t, kernel)
    return (P @ jm[idx][..., None])[..., 0], T + P @ jc[idx] @ _T(P)


def predict_f(ssm, kernel, cond_t, new_t):
    m, S = predict_state(ssm, kernel, cond_t, new_t)
    h = kernel.emission_vector()[0]
    return (m @ h)[:, None], np.einsum("i,nij,j->n", h, S, h)[:, None]


class SparseCVIGaussianProcess:
    """sparse_variational_cvi.py:38-292."""

    def __init__(self, kernel, inducing_points, likelihood, learning_rate=0.1):
        self.kernel, self.z, self.lik, self.lr = kernel, np.asarray(inducing_points, dtype=np.float64), likelihood, learning_rate
        M, sd = self.z.shape[0], kernel.state_dim
        self.nat1 = np.zeros((M + 1, 2 * sd))
        self.nat2 = np.zeros((M + 1, 2 * sd, 2 * sd))

    @property
    def dist_p(self):
        return self.kernel.state_space_model(self.z)

    @property
    def dist_q(self):
        pd, ps = self.dist_p.precision()
        sd = self.kernel.state_dim
        lin = self.nat1[1:, :sd] + self.nat1[:-1, sd:]
        diag = self.nat2[1:, :sd, :sd] + self.nat2[:-1, sd:, sd:]
        sub = self.nat2[1:-1, sd:, :sd]
        return np_transforms.ssm_from_params(np_transforms.naturals_to_ssm_params(lin, -0.5 * pd + diag, -ps + 2.0 * sub))

    def update_sites(self, t, y):
        mu, var = predict_f(self.dist_q, self.kernel, self.z, t)
        g1, g2 = self.lik.grads_expectation(mu, var, y)
        P, _, idx = conditional_statistics(t, self.z, self.kernel)
        HP = self.kernel.emission_vector()[None] @ P                     # [N, 1, 2d]
        bp1 = np.sum(HP * g1[..., None], axis=-2)
        bp2 = np.sum(g2[..., None, None] * HP[..., None] * HP[..., None, :], axis=-3)
        s1, s2 = np.zeros_like(self.nat1), np.zeros_like(self.nat2)
        np.add.at(s1, idx, bp1)
        np.add.at(s2, idx, bp2)
        self.nat1 = (1 - self.lr) * self.nat1 + self.lr * s1
        self.nat2 = (1 - self.lr) * self.nat2 + self.lr * s2

    def classic_elbo(self, t, y):
        q = self.dist_q
        mu, var = predict_f(q, self.kernel, self.z, t)
        return np.sum(self.lik.variational_expectations(mu, var, y)) - np.sum(q.kl_divergence(self.dist_p))
