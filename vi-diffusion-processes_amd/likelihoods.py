"""
Host-side mirror of the Gaussian likelihoods on the path (markovflow/likelihoods/multivariate_gaussian.py:80-115;
gpflow.likelihoods.Gaussian for the scalar CVI-GP case): variational expectations and their gradients in
closed form (the reference differentiates them with a GradientTape, variational_cvi_sde.py:204-220).
Observation counts are tiny next to the time grid (n_obs << T), so these run as small batched torch ops on
the gathered observation nodes.
"""
import math
import weakref

import torch

from . import linalg


class MultivariateGaussian:
    """p(y | f) = N(y; f, L L^T) (multivariate_gaussian.py:29-160)."""

    # d VE / d(eta) = (S^{-1} y, -1/2 S^{-1}) does not depend on q, and its second part is the same block for every observation: the
    # CVI-DP model then keeps the data sites out of the per-node arrays (variational_cvi_sde.CVISitesSDE, cq state)
    uniform_site_gradient = True

    def __init__(self, chol_covariance):
        self.chol_covariance = chol_covariance

    @property
    def chol_covariance(self):
        return self._chol

    @chol_covariance.setter
    def chol_covariance(self, chol_covariance):
        """Everything derived from the factor is rebuilt when it is replaced (no stale inverse / constant / gradient cache)."""
        self._chol = chol_covariance
        self.obs_dim = chol_covariance.shape[-1]
        # one d x d inverse, reused by every call (no per-observation triangular solves)
        self.inv_covariance = linalg.spd_inverse(chol=chol_covariance)
        self.log_det_chol = torch.log(torch.diagonal(chol_covariance)).sum()
        # additive constant of the variational expectations, as a host scalar (one synchronisation, here)
        self.ve_constant = -float(self.log_det_chol) - 0.5 * self.obs_dim * math.log(2.0 * math.pi)
        self._g_cache = None

    def variational_expectations(self, f_means, f_covariances, observations):
        """-1/2 tr(S^{-1} S_i) + log N(y_i; mu_i, S) (multivariate_gaussian.py:80-115); shape [..., n]."""
        Sinv = self.inv_covariance
        diff = observations - f_means
        # element-wise d x d contraction (a [n, d] x [d, d] GEMM call costs ~0.2 ms of launch-bound rocBLAS time at n ~ 1e5)
        quad = (diff[..., :, None] * Sinv * diff[..., None, :]).sum(dim=(-1, -2))
        logp = -0.5 * quad - self.log_det_chol - 0.5 * self.obs_dim * math.log(2 * math.pi)
        return -0.5 * (Sinv * f_covariances).sum(dim=(-1, -2)) + logp

    def ve_gradients_expectation(self, f_means, f_covariances, observations):
        """
        Gradient of sum_i VE_i with respect to the expectation parameters (mu, S + mu mu^T):
        d/dmu = S^{-1}(y - mu), d/dS = -1/2 S^{-1}, then gradient_transformation_mean_var_to_expectation
        (variational_cvi.py:448-462): g1 = d/dmu - 2 (d/dS) mu = S^{-1} y,  g2 = -1/2 S^{-1}.
        """
        # both gradients depend on the observations only: computed once per observation TENSOR OBJECT and version (a raw address
        # is no key: the caching allocator hands freed blocks back at the same address with the same shape)
        c = self._g_cache
        if (c is None or c[0]() is not observations or c[1] != observations._version or c[2] != tuple(f_covariances.shape)
                or c[3] != self.inv_covariance._version):
            Sinv = self.inv_covariance
            g1 = (Sinv * observations[..., None, :]).sum(-1)           # S^{-1} y (S symmetric)
            g2 = (-0.5 * Sinv).expand(f_covariances.shape).contiguous()
            c = self._g_cache = (weakref.ref(observations), observations._version, tuple(f_covariances.shape), Sinv._version, g1, g2)
        return c[4], c[5]


class Gaussian:
    """Scalar Gaussian likelihood with variance `variance` (gpflow.likelihoods.Gaussian), obs_dim 1."""

    def __init__(self, variance):
        self.variance = float(variance)

    def variational_expectations(self, f_means, f_vars, observations):
        v = self.variance
        return -0.5 * math.log(2 * math.pi) - 0.5 * math.log(v) - 0.5 * ((observations - f_means) ** 2 + f_vars) / v

    def variational_expectations_terms(self, f_means, f_vars, observations):
        """(c, w, [t1, t2]) with  sum of variational_expectations = c + w (t1 + t2): the two reductions |y - m|^2 and sum var as device
        scalars, for callers that assemble their bound in one launch (mfgm_combine_terms)."""
        v = self.variance
        r = (observations - f_means).reshape(-1)
        return -0.5 * r.numel() * (math.log(2 * math.pi) + math.log(v)), -0.5 / v, [torch.dot(r, r).reshape(1), f_vars.sum().reshape(1)]

    def variational_expectations_sum(self, f_means, f_vars, observations):
        """sum of variational_expectations over all points, from two reductions instead of seven element-wise passes and one."""
        c, w, (t1, t2) = self.variational_expectations_terms(f_means, f_vars, observations)
        return c + w * (t1 + t2)[0]

    def ve_gradients_expectation(self, f_means, f_vars, observations):
        # (y / v, -1/2 / v) depend on the observations only: computed once per observation tensor object and version
        v = self.variance
        c = getattr(self, "_g_cache", None)
        if c is None or c[0]() is not observations or c[1] != observations._version or c[2] != tuple(f_vars.shape) or c[3] != v:
            c = self._g_cache = (weakref.ref(observations), observations._version, tuple(f_vars.shape), v, observations / v,
                                 torch.full_like(f_vars, -0.5 / v))
        return c[4], c[5]
