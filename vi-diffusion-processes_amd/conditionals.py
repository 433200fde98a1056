"""
Host-side mirror of markovflow/conditionals.py: prediction between conditioning points (`conditional_statistics`,
`_conditional_statistics_from_transitions`, `conditional_predict`, `base_conditional_predict`, `pairwise_marginals`;
conditionals.py:29-470).  Every query point is independent: these are batched d x d torch operations on the device,
fed by the marginal / cross-covariance blocks of the HIP sweeps (factorisations and solves through vidp_amd.linalg).
"""
import torch

from . import linalg

APPROX_INF = 1e10   # markovflow/base.py:46


def _T(x):
    return x.transpose(-1, -2)


def _conditional_statistics_from_transitions(A_mt, Q_mt, A_tp, Q_tp, return_precision=False):
    """p(x_t | x_-, x_+) = N(D x_- + E x_+, T) from the two transitions around t (conditionals.py:111-204)."""
    A_tp_Q_mt = A_tp @ Q_mt
    Q_mp = Q_tp + A_tp @ _T(A_tp_Q_mt)
    chol = linalg.cholesky(Q_mp)
    Linv = linalg.solve_lower(chol, A_tp_Q_mt)
    E = _T(linalg.solve_lower_t(chol, Linv))
    D = A_mt - E @ A_tp @ A_mt
    if return_precision:
        Q_mt_inv = linalg.spd_inverse(Q_mt)
        LA = linalg.solve_lower(linalg.cholesky(Q_tp), A_tp)
        return D, E, Q_mt_inv + _T(LA) @ LA
    return D, E, Q_mt - _T(Linv) @ Linv


def _conditional_statistics(new_time_points, training_time_points, kernel):
    """(P, T, insertion indices) (conditionals.py:207-256)."""
    idx = torch.searchsorted(training_time_points.contiguous(), new_time_points.contiguous())
    inf = APPROX_INF * torch.ones_like(training_time_points[..., -1:])
    aug = torch.cat([-inf, training_time_points, inf], dim=-1)
    plus = torch.gather(aug, -1, idx + 1)
    minus = torch.gather(aug, -1, idx)
    A_mt, Q_mt = kernel.transition_statistics_local(new_time_points - minus)
    A_tp, Q_tp = kernel.transition_statistics_local(plus - new_time_points)
    F, G, T = _conditional_statistics_from_transitions(A_mt, Q_mt, A_tp, Q_tp)
    return torch.cat([F, G], dim=-1), T, idx


def conditional_statistics(new_time_points, training_time_points, kernel):
    """conditionals.py:79-108."""
    P, T, _ = _conditional_statistics(new_time_points, training_time_points, kernel)
    return P, T


def base_conditional_predict(conditional_projections, conditional_covariances, adjacent_states, pairwise_state_covariances=None):
    """p(x_t) = N(P m, T + P S P^T) (conditionals.py:380-421)."""
    means = (conditional_projections @ adjacent_states[..., None])[..., 0]
    covs = conditional_covariances
    if pairwise_state_covariances is not None:
        covs = covs + conditional_projections @ pairwise_state_covariances @ _T(conditional_projections)
    return means, covs


def conditional_predict(new_time_points, training_time_points, kernel, training_pairwise_means, training_pairwise_covariances=None):
    """conditionals.py:29-76."""
    P, T, idx = _conditional_statistics(new_time_points, training_time_points, kernel)
    d2 = training_pairwise_means.shape[-1]
    pm = torch.gather(training_pairwise_means, -2, idx[..., None].expand(idx.shape + (d2,)))
    pc = None
    if training_pairwise_covariances is not None:
        pc = torch.gather(training_pairwise_covariances, -3, idx[..., None, None].expand(idx.shape + (d2, d2)))
    return base_conditional_predict(P, T, pm, pc)


def pairwise_marginals(dist, initial_mean, initial_covariance):
    """Joint moments of every pair of subsequent states, padded with the prior at both ends (conditionals.py:424-470)."""
    means, covs = dist.marginals
    sub = dist.subsequent_covariances()
    im = initial_mean.to(means.device).expand(means.shape[:-2] + (1, means.shape[-1]))
    ext_m = torch.cat([im, means, im], dim=-2)
    joint_mean = torch.cat([ext_m[..., :-1, :], ext_m[..., 1:, :]], dim=-1)
    ic = initial_covariance.to(covs.device).expand(covs.shape[:-3] + (1,) + tuple(covs.shape[-2:]))
    ext_c = torch.cat([ic, covs, ic], dim=-3)
    zero = torch.zeros_like(ic)
    ext_s = torch.cat([zero, sub, zero], dim=-3)
    top = torch.cat([ext_c[..., :-1, :, :], _T(ext_s)], dim=-1)
    bot = torch.cat([ext_s, ext_c[..., 1:, :, :]], dim=-1)
    return joint_mean, torch.cat([top, bot], dim=-2)
