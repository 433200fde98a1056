"""
Host-side mirror of markovflow/kalman_filter.py: the sparse-precision ("SpInGP") Kalman filter family
(`KalmanFilter`, `GaussianSitesNat`, `KalmanFilterWithSites`, `KalmanFilterWithSparseSites`;
kalman_filter.py:32-639).  Posterior precision = prior block-tri-diagonal precision + block-diagonal likelihood
precision; one block Cholesky (HIP forward sweep) gives the log-determinant and |L^{-1} G^T Sigma^{-1} y|^2.
The per-observation algebra (output_dim 1-2) is tiny batched torch work; everything sequential in T runs in the sweeps.
"""
import math
import os

import torch

from . import linalg
from ._lib import FULL, SYM, TRI, VEC
from .state_space_model import StateSpaceModel


class GaussianSitesNat:
    """Gaussian sites in natural form (kalman_filter.py:382-436): nat1 [N, D], nat2 [N, D, D]."""

    def __init__(self, nat1, nat2, log_norm=None):
        if nat1.dim() != 2 or nat2.dim() != 3 or nat2.shape != (nat1.shape[0], nat1.shape[1], nat1.shape[1]):
            raise ValueError("GaussianSitesNat: nat1 must be [N, D] and nat2 [N, D, D]")
        self.nat1, self.nat2, self.log_norm = nat1, nat2, log_norm
        self.num_data, self.output_dim = nat1.shape

    @property
    def means(self):
        return 0.5 * linalg.cholesky_solve(self.nat1, linalg.cholesky(-self.nat2))     # -1/2 nat2^{-1} nat1

    @property
    def precisions(self):
        return -2.0 * self.nat2

    @property
    def log_det_precisions(self):
        return torch.log(-2.0 * self.nat2)


class BaseKalmanFilter:
    """kalman_filter.py:32-271."""

    def __init__(self, state_space_model: StateSpaceModel, emission_model):
        self.prior_ssm = state_space_model
        self.emission = emission_model

    # subclasses: _r_inv, observations, _log_det_observation_precision
    def _H(self):
        """emission matrix broadcast to [B, T, o, d]."""
        ssm = self.prior_ssm
        H = self.emission.emission_matrix
        return H.expand(ssm.batch_shape + tuple(H.shape[-3:])).reshape((ssm.B,) + tuple(H.shape[-3:]))

    def _rinv_full(self):
        """observation precision broadcast to [B, T, o, o]."""
        ssm = self.prior_ssm
        R = self._r_inv
        o = R.shape[-1]
        return R.expand((ssm.B, ssm.T, o, o)) if R.dim() <= 3 else R.reshape(ssm.B, ssm.T, o, o)

    def _post_precision(self):
        """K^{-1} + G^T Sigma^{-1} G (kalman_filter.py:86-101), packed."""
        ssm, pl = self.prior_ssm, self.prior_ssm.plan
        pr = ssm._precision_packed()
        H, Rinv = self._H(), self._rinv_full()
        hrh = torch.einsum("...ji,...jk,...kl->...il", H, Rinv, H)
        D = pl.pack(SYM, hrh.contiguous())
        pl.lincomb(D, 1.0, D, 1.0, pr["diag"])
        return D, pr["sub"], pr

    def _back_project(self, y):
        """(G^T Sigma^{-1}) y (kalman_filter.py:257-271): y [B, T, o] -> [B, T, d]."""
        return torch.einsum("...ij,...ki,...k->...j", self._H(), self._rinv_full(), y)

    def _obs(self):
        y = self.observations
        return y.expand(self.prior_ssm.batch_shape + tuple(y.shape[-2:])).reshape((self.prior_ssm.B,) + tuple(y.shape[-2:]))

    def log_likelihood(self):
        """log p(obs) summed over the batch (kalman_filter.py:184-255)."""
        ssm, pl = self.prior_ssm, self.prior_ssm.plan
        D, S, pr = self._post_precision()
        mu_p = pl.unpack(VEC, ssm._posterior_packed()["s"]["x"])
        marginal = torch.einsum("...ij,...j->...i", self._H(), mu_p)
        disp = self._obs() - marginal
        o = self._H().shape[-2]
        cst = -0.5 * math.log(2 * math.pi) * o * self._num_data()
        term1 = -0.5 * self._term1(disp)
        obs_proj = self._back_project(self._disp_grid(disp))
        f = pl.factor(D, S, pl.pack(VEC, obs_proj.contiguous()), want_logdet=True, want_quad=True)
        pl.check_info()
        term2 = 0.5 * f["quad"]
        term3 = -pr["sumlogchol"] - f["logdet"] + 0.5 * self._log_det_observation_precision
        return (cst + term1 + term2 + term3).sum()

    # defaults for the dense variants ---------------------------------------------------------------------
    def _num_data(self):
        return self.prior_ssm.T

    def _term1(self, disp):
        return torch.einsum("...op,...p,...o->...o", self._rinv_full(), disp, disp).sum(dim=(-1, -2))

    def _disp_grid(self, disp):
        return disp

    def posterior_state_space_model(self) -> StateSpaceModel:
        """
        The posterior as a StateSpaceModel (kalman_filter.py:109-182).  The reference runs a backward UDU^T recursion
        (upper_diagonal_lower); the posterior's natural parameters determine the same (A_k, b_k, Q_k, mu0, P0) uniquely,
        so they are obtained with the forward / backward sweeps of naturals_to_ssm_params instead.
        """
        from .ssm_gaussian_transformations import naturals_to_ssm_params_packed
        ssm, pl = self.prior_ssm, self.prior_ssm.plan
        D, S, pr = self._post_precision()
        # theta_lin = G^T Sigma^{-1} y + K_prior^{-1} mu_prior ; theta_diag = -1/2 P_diag ; theta_sub = -P_sub
        lin = pl.pack(VEC, self._back_project(self._obs_grid()).contiguous())
        pl.lincomb(lin, 1.0, lin, 1.0, pr["lin"])
        td = pl.lincomb(pl.empty(SYM), -0.5, D)
        ts = pl.lincomb(pl.empty(FULL), -1.0, S)
        post = naturals_to_ssm_params_packed(pl, lin, td, ts)
        post.batch_shape = ssm.batch_shape
        return post

    def _obs_grid(self):
        return self._obs()


class KalmanFilter(BaseKalmanFilter):
    """kalman_filter.py:275-345: one observation-noise Cholesky [o, o] shared by all time steps."""

    def __init__(self, state_space_model, emission_model, observations, chol_obs_covariance):
        super().__init__(state_space_model, emission_model)
        o = emission_model.output_dim
        if tuple(chol_obs_covariance.shape) != (o, o):
            raise ValueError("The shape of the observation covariance matrix and the emission matrix are not compatible")
        want = tuple(state_space_model.batch_shape) + (state_space_model.T, o)
        if tuple(observations.shape) != want:
            raise ValueError("The shape of the observations and the state-space-model parameters are not compatible")
        self._chol_obs_covariance = chol_obs_covariance
        self._observations = observations
        self._rinv = linalg.spd_inverse(chol=chol_obs_covariance)

    @property
    def _r_inv(self):
        return self._rinv

    @property
    def observations(self):
        return self._observations

    @property
    def _log_det_observation_precision(self):
        return -2.0 * self.prior_ssm.T * torch.log(torch.diagonal(self._chol_obs_covariance)).sum()


def fused_sites_call(ssm, emission, sites):
    """(KfSites struct, keep-alive tuple) for the fused Kalman-filter-with-sites entry points (mfgm_kf_sites_*), or None when the
    problem does not qualify: state_dim <= 8, output_dim <= 2, a time-invariant emission matrix, one site per time point."""
    import ctypes
    from . import _lib
    Hc = getattr(emission, "constant_matrix", None)
    if Hc is None or ssm.d > 8 or ssm.plan.d != ssm.d or Hc.shape[0] > 2 or os.environ.get("VIDP_FUSED_KF", "1") == "0":
        return None
    o, T = int(Hc.shape[0]), ssm.T
    n1, n2 = sites.nat1, sites.nat2
    if n1.dim() != 2 or tuple(n1.shape) not in ((T, o), (ssm.B * T, o)) or not n1.is_cuda:
        return None
    sv = _lib.KfSites()
    Hh = Hc.detach().to("cpu", torch.float64).reshape(-1).tolist() if "_kf_H" not in emission.__dict__ else emission._kf_H
    emission.__dict__["_kf_H"] = Hh
    for i, v in enumerate(Hh):
        sv.H[i] = v
    n1c, n2c = n1.contiguous(), n2.contiguous()
    sv.o, sv.site_batch = o, (1 if n1.shape[0] == T else ssm.B)
    sv.nat1, sv.nat2 = n1c.data_ptr(), n2c.data_ptr()
    # H mu_prior: zero for the kernel priors; computed once per prior otherwise
    cache = ssm.__dict__.setdefault("_kf_cache", {})
    if "zero_mean" not in cache:
        cache["zero_mean"] = bool((ssm._mu0 == 0).all() and (ssm._b == 0).all())
    hmu = None
    if not cache["zero_mean"]:
        key = ("Hmu", tuple(Hh))
        if key not in cache:
            mu = ssm.plan.unpack(VEC, ssm._posterior_packed()["s"]["x"])
            cache[key] = torch.einsum("ai,bti->bta", Hc.to(mu.device), mu).contiguous()
        hmu = cache[key]
        sv.Hmu = hmu.data_ptr()
    return sv, (n1c, n2c, hmu), cache


def _kf_scratch(cache, pl, names):
    bufs = cache.setdefault("bufs", {})
    kinds = dict(D=SYM, r=VEC, L=TRI, y=VEC, Sig=SYM, x=VEC)
    for nm in names:
        if nm not in bufs:
            bufs[nm] = pl.empty(kinds[nm])
    return bufs


class KalmanFilterWithSites(BaseKalmanFilter):
    """kalman_filter.py:440-500: time-dependent Gaussian likelihood terms (sites) in natural form."""

    def __init__(self, state_space_model, emission_model, sites: GaussianSitesNat):
        if sites.output_dim != emission_model.output_dim:
            raise ValueError("The shape of the site matrices and the emission matrix are not compatible")
        self.sites = sites
        super().__init__(state_space_model, emission_model)

    def log_likelihood(self):
        """kalman_filter.py:184-255.  With a time-invariant emission matrix: one library call (assembly of the posterior precision
        in the packed layout, the forward sweeps, the observation-term sums), a handful of [B]-sized operations here."""
        import ctypes
        from .packed import _ptr, _stream
        from . import _lib
        ssm, pl = self.prior_ssm, self.prior_ssm.plan
        call = fused_sites_call(ssm, self.emission, self.sites)
        if call is None:
            return super().log_likelihood()
        sv, keep, cache = call
        pr = ssm._precision_packed()
        b = _kf_scratch(cache, pl, ("D", "r", "L", "y"))
        total = torch.empty(1, dtype=torch.float64, device=pl.device)
        pl.epoch += 1
        cst = -0.5 * math.log(2 * math.pi) * sv.o * ssm.T
        # cst + term1 + term2 + term3 (kalman_filter.py:229-255) assembled by the library call itself; NaN when a pivot block was not
        # positive definite (no host synchronisation in the loop: plan.check_info() raises for it when asked, VIDP_EAGER_CHECK=1 asks
        # every time)
        _lib.check(pl.lib.mfgm_kf_sites_elbo(pl.h, ctypes.byref(sv), _ptr(pr["diag"]), _ptr(pr["sub"]), _ptr(b["D"]), _ptr(b["r"]),
                                             _ptr(b["L"]), _ptr(b["y"]), _ptr(pr["sumlogchol"]), cst, None, None, _ptr(total), _ptr(pl.ws),
                                             _ptr(pl.info), _stream()), "mfgm_kf_sites_elbo")
        # the factorisation now in (L, y) and the workspace belongs to these sites: with a zero-mean prior the prediction at the data
        # solves the same system with the same right-hand side (_predict_f_fused reuses it).  The site tensors are kept alive here,
        # so that `is` identifies them.
        cache["factor_of"] = (keep[0], keep[0]._version, keep[1], keep[1]._version, pl.epoch) if cache["zero_mean"] else None
        if os.environ.get("VIDP_EAGER_CHECK", "0") == "1":
            pl.check_info()
        return total[0]

    @property
    def _r_inv(self):
        return self.sites.precisions

    @property
    def observations(self):
        return self.sites.means

    @property
    def _log_det_observation_precision(self):
        return linalg.logdet_spd(self._r_inv).sum(-1)


class KalmanFilterWithSparseSites(BaseKalmanFilter):
    """kalman_filter.py:504-639: sites only at `observations_index` of a finer time grid (output_dim 1, no batch)."""

    def __init__(self, state_space_model, emission_model, sites: GaussianSitesNat, num_grid_points, observations_index,
                 observations):
        self.sites = sites
        self.observations_index = observations_index.reshape(-1).to(torch.int64)
        if observations.dim() == 3:
            if observations.shape[0] != 1:
                raise ValueError("KalmanFilterWithSparseSites doesn't support batches")
            observations = observations[0]
        self.sparse_observations = observations
        self.num_grid_points = int(num_grid_points)
        super().__init__(state_space_model, emission_model)
        if state_space_model.B != 1:
            raise ValueError("KalmanFilterWithSparseSites doesn't support batches")

    def _scatter(self, vals, tail):
        out = torch.zeros((self.num_grid_points,) + tail, dtype=vals.dtype, device=vals.device)
        out.index_add_(0, self.observations_index, vals)      # tf.scatter_nd accumulates duplicates
        return out

    @property
    def _r_inv(self):
        return self._scatter(self.sites.precisions, (1, 1))

    @property
    def observations(self):
        return self._scatter(self.sparse_observations, (1,))

    def _num_data(self):
        return self.observations_index.numel()

    def _term1(self, disp):
        marginal = self._obs()[0] - disp[0]
        dd = self.sparse_observations - marginal[self.observations_index]
        return torch.einsum("...op,...p,...o->...o", self.sites.precisions, dd, dd).sum(dim=(-1, -2))

    @property
    def _log_det_observation_precision(self):
        return linalg.logdet_spd(self.sites.precisions).sum(-1)
