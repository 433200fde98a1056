"""
Host-side mirror of markovflow/ssm_natgrad.py (`SSMNaturalGradient`, ssm_natgrad.py:33-244): natural-gradient descent on a
`StateSpaceModel` q,  theta <- theta - gamma dL/d eta,  followed by naturals_to_ssm_params (HIP sweeps).

The reference obtains dL/d eta from a persistent GradientTape through the loss and `expectations_to_ssm_params`
(ssm_natgrad.py:142-172), which needs gradients *through* the banded Cholesky / sparse inverse.  Differentiated sweeps are
not built yet; instead the loss object supplies dL/d eta itself (`grad_wrt_expectations`), which is available in closed form
for the losses on the path: the ELBO of a Gauss-Markov q is  sum_t VE_t(mu_t, Sigma_tt) - KL[q || p]  with
d KL / d eta = theta_q - theta_p.  `GaussMarkovELBO` below is that loss (the `VariationalGaussianProcess.elbo` of the reference,
models/variational.py:129-152).  The Adam-like momentum variant (ssm_natgrad.py:177-208) needs Fisher-vector products and is
not implemented.
"""
import torch

from ._lib import FULL, SYM, TRI, VEC
from .ssm_gaussian_transformations import naturals_to_ssm_params_packed
from .state_space_model import StateSpaceModel
from .variational_cvi import back_project_nats


class GaussMarkovELBO:
    """-ELBO of q (a StateSpaceModel) for prior SSM p, emission H [.., T, 1, d], likelihood and observations [.., T, 1]."""

    def __init__(self, prior_ssm: StateSpaceModel, emission_model, likelihood, observations):
        self.p, self.emission, self.likelihood, self.y = prior_ssm, emission_model, likelihood, observations

    def _f_marginals(self, q: StateSpaceModel):
        mu, cov = q.marginals
        return self.emission.project_state_to_f(mu), self.emission.project_state_covariance_to_f(cov, full_output_cov=False)

    def elbo(self, q: StateSpaceModel):
        fm, fv = self._f_marginals(q)
        ve = self.likelihood.variational_expectations(fm, fv, self.y).sum()
        return ve - q.kl_divergence(self.p).sum()

    def __call__(self, q):
        return -self.elbo(q)

    def grad_wrt_expectations(self, q: StateSpaceModel):
        """d(-ELBO)/d(eta_lin, eta_diag, eta_sub), packed on q's plan: (theta_q - theta_p) - back-projected VE gradients."""
        pl = q.plan
        if self.p.plan is not pl:
            raise ValueError("the prior and q must share a partition plan")
        nq = pl.ssm_to_naturals(q.packed.A, q.packed.off, q.packed.chol)
        np_ = pl.ssm_to_naturals(self.p.packed.A, self.p.packed.off, self.p.packed.chol)
        fm, fv = self._f_marginals(q)
        g1, g2 = self.likelihood.ve_gradients_expectation(fm, fv, self.y)
        H = self.emission.emission_matrix
        bp1, bp2 = back_project_nats(g1, g2, H)
        B, T, d = q.B, q.T, q.d
        bp1 = pl.pack(VEC, bp1.expand(q.batch_shape + (T, d)).reshape(B, T, d).contiguous())
        bp2 = pl.pack(SYM, bp2.expand(q.batch_shape + (T, d, d)).reshape(B, T, d, d).contiguous())
        gl = pl.lincomb(pl.empty(VEC), 1.0, nq["lin"], -1.0, np_["lin"], -1.0, bp1)
        gd = pl.lincomb(pl.empty(SYM), 1.0, nq["diag"], -1.0, np_["diag"], -1.0, bp2)
        gs = pl.lincomb(pl.empty(FULL), 1.0, nq["sub"], -1.0, np_["sub"])
        return (gl, gd, gs), nq


class SSMNaturalGradient:
    """ssm_natgrad.py:33-244."""

    def __init__(self, gamma=0.1, momentum=False, beta1=0.9, beta2=0.99, epsilon=1e-8, name=None):
        if momentum:
            raise NotImplementedError("the momentum variant needs Fisher-vector products through the sweeps (not built yet)")
        self.gamma = float(gamma)

    def minimize(self, loss_fn, ssm: StateSpaceModel):
        """One natural-gradient step on `ssm` in place (ssm_natgrad.py:95-119)."""
        self._natgrad_step(loss_fn, ssm)

    def _natgrad_step(self, loss_fn, ssm: StateSpaceModel):
        """theta <- theta - gamma dL/d eta, then back to SSM parameters (ssm_natgrad.py:121-218)."""
        if not hasattr(loss_fn, "grad_wrt_expectations"):
            raise NotImplementedError(
                "SSMNaturalGradient needs a loss object with grad_wrt_expectations(ssm): gradients through the block-tri-diagonal "
                "sweeps (the reference's GradientTape through banded ops) are not implemented")
        pl = ssm.plan
        (gl, gd, gs), nq = loss_fn.grad_wrt_expectations(ssm)
        tl = pl.lincomb(pl.empty(VEC), 1.0, nq["lin"], -self.gamma, gl)
        td = pl.lincomb(pl.empty(SYM), 1.0, nq["diag"], -self.gamma, gd)
        ts = pl.lincomb(pl.empty(FULL), 1.0, nq["sub"], -self.gamma, gs)
        new = naturals_to_ssm_params_packed(pl, tl, td, ts)
        # assign in place (ssm_natgrad.py:213-218)
        keep_batch = ssm.batch_shape
        ssm.__dict__.update(new.__dict__)
        ssm.batch_shape = keep_batch
