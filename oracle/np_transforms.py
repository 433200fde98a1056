"""
Oracle (test infrastructure, see oracle/__init__.py): SSM <-> expectation parameters <-> natural
parameters, restating markovflow/ssm_gaussian_transformations.py in NumPy.
"""
import numpy as np

from . import np_btd
from .np_ssm import StateSpaceModel, chol_solve

_T = np_btd._T


def ssm_to_expectations(ssm: StateSpaceModel):
    """ssm_gaussian_transformations.py:32-89."""
    mu = ssm.marginal_means[..., None]
    cov = ssm.marginal_covariances
    eta_diag = cov + mu @ _T(mu)
    eta_sub = ssm.A @ cov[..., :-1, :, :] + mu[..., 1:, :, :] @ _T(mu[..., :-1, :, :])
    return mu[..., 0], eta_diag, eta_sub


def expectations_to_ssm_params(eta_linear, eta_diag, eta_subdiag):
    """ssm_gaussian_transformations.py:93-178.  Returns (As, offsets, chol_P0, chol_Qs, mu0)."""
    m = np.asarray(eta_linear, dtype=np.float64)[..., None]
    cov = eta_diag - m @ _T(m)
    cov_sub = _T(eta_subdiag) - m[..., :-1, :, :] @ _T(m[..., 1:, :, :])
    chols = np.linalg.cholesky(cov)
    As = _T(chol_solve(chols[..., :-1, :, :], cov_sub))
    offsets = (m[..., 1:, :, :] - As @ m[..., :-1, :, :])[..., 0]
    cond = cov[..., 1:, :, :] - As @ (cov[..., :-1, :, :] @ _T(As))
    return As, offsets, chols[..., 0, :, :], np.linalg.cholesky(cond), m[..., 0, :, 0]


def ssm_to_naturals(ssm: StateSpaceModel):
    """ssm_gaussian_transformations.py:182-253."""
    As = ssm.A
    offsets = ssm.concatenated_state_offsets[..., None]
    chols = ssm.concatenated_cholesky_process_covariance
    Linv_As = np_btd.solve_lower(chols[..., 1:, :, :], As)
    theta_sub = np_btd.solve_upper_from_lower(chols[..., 1:, :, :], Linv_As)
    tmp = chol_solve(chols, offsets)
    theta_lin = tmp.copy()
    theta_lin[..., :-1, :, :] -= _T(As) @ tmp[..., 1:, :, :]
    aqa = _T(Linv_As) @ Linv_As
    eye = np.broadcast_to(np.eye(ssm.state_dim), chols.shape)
    prec = chol_solve(chols, eye)
    prec[..., :-1, :, :] += aqa
    return theta_lin[..., 0], -0.5 * prec, theta_sub


def ssm_to_naturals_no_smoothing(ssm: StateSpaceModel):
    """ssm_gaussian_transformations.py:257-329."""
    chols = ssm.concatenated_cholesky_process_covariance
    theta_sub = chol_solve(chols[..., 1:, :, :], ssm.A)
    theta_lin = chol_solve(chols, ssm.concatenated_state_offsets[..., None])[..., 0]
    eye = np.broadcast_to(np.eye(ssm.state_dim), chols.shape)
    return theta_lin, -0.5 * chol_solve(chols, eye), theta_sub


def naturals_to_ssm_params(theta_linear, theta_diag, theta_subdiag):
    """
    ssm_gaussian_transformations.py:333-511.  Returns (As, offsets, chol_P0, chol_Qs, mu0).
    The reference's `solve_triang_band(A^{-T}, P)` keeps only the block diagonal, which is
    Q_k^{-1} = P_kk + A_{k+1}^T P_{k+1,k} (row k of the unit upper-bidiagonal back-substitution).
    """
    theta_linear = np.asarray(theta_linear, dtype=np.float64)
    pd = -2.0 * np.asarray(theta_diag, dtype=np.float64)
    ps = -np.asarray(theta_subdiag, dtype=np.float64)
    Ld, Ls = np_btd.cholesky(pd, ps)
    cov, cov_sub = np_btd.inverse_blocks(Ld, Ls)  # S_kk, S_{k+1,k}
    # As = (S_kk^{-1} S_{k,k+1})^T  (tf.linalg.solve, :462)
    As = _T(np.linalg.solve(cov[..., :-1, :, :], _T(cov_sub)))
    low = np.tril(pd)
    pdsym = low + _T(np.tril(pd, -1))
    cond_prec = pdsym.copy()
    cond_prec[..., :-1, :, :] += _T(As) @ ps
    cond_prec = 0.5 * (cond_prec + _T(cond_prec))
    chol_cp = np.linalg.cholesky(cond_prec)
    eye = np.broadcast_to(np.eye(pd.shape[-1]), pd.shape)
    covs = chol_solve(chol_cp, eye)
    chols = np.linalg.cholesky(covs)
    # (A^{-T})^{-1} theta: z_T = theta_T, z_k = theta_k + A_{k+1}^T z_{k+1}
    T = pd.shape[-3]
    z = np.empty_like(theta_linear)
    z[..., T - 1, :] = theta_linear[..., T - 1, :]
    for k in range(T - 2, -1, -1):
        z[..., k, :] = theta_linear[..., k, :] + (_T(As[..., k, :, :]) @ z[..., k + 1, :, None])[..., 0]
    off = (covs @ z[..., None])[..., 0]
    return As, off[..., 1:, :], chols[..., 0, :, :], chols[..., 1:, :, :], off[..., 0, :]


def naturals_to_ssm_params_no_smoothing(theta_linear, theta_diag, theta_subdiag):
    """ssm_gaussian_transformations.py:515-593."""
    c = np.linalg.cholesky(-2.0 * np.asarray(theta_diag, dtype=np.float64))
    As = chol_solve(c[..., 1:, :, :], theta_subdiag)
    off = chol_solve(c, np.asarray(theta_linear, dtype=np.float64)[..., None])[..., 0]
    eye = np.broadcast_to(np.eye(c.shape[-1]), c.shape)
    chols = np.linalg.cholesky(chol_solve(c, eye))
    return As, off[..., 1:, :], chols[..., 0, :, :], chols[..., 1:, :, :], off[..., 0, :]


def ssm_from_params(params):
    As, offsets, cholP0, cholQ, mu0 = params
    return StateSpaceModel(mu0, cholP0, As, offsets, cholQ)
