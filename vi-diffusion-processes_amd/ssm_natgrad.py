"""
Host-side mirror of markovflow/ssm_natgrad.py (`SSMNaturalGradient`, ssm_natgrad.py:33-244): natural-gradient descent on a
`StateSpaceModel` q,  theta <- theta - gamma dL/d eta,  followed by naturals_to_ssm_params (HIP sweeps).

The reference obtains dL/d eta from a persistent GradientTape through the loss and `expectations_to_ssm_params`
(ssm_natgrad.py:142-172), which needs gradients *through* the banded Cholesky / sparse inverse.  Two routes here:
  * a loss object that supplies dL/d eta itself (`grad_wrt_expectations`), available in closed form for the losses on the path: the
    ELBO of a Gauss-Markov q is  sum_t VE_t(mu_t, Sigma_tt) - KL[q || p]  with d KL / d eta = theta_q - theta_p.  `GaussMarkovELBO`
    below is that loss (the `VariationalGaussianProcess.elbo` of the reference, models/variational.py:129-152);
  * any closure `loss_fn(q)` of a differentiable view q of the model (`vidp_amd.tape.TapeSSM`): torch autograd with the sweeps as a
    custom Function whose backward is the Fisher-vector product (vidp_amd/tape.py) -- the tape route of the reference.

The Adam-like momentum variant (ssm_natgrad.py:36-58, 177-208; the reference's default) keeps moving averages m of the natural
gradient g = dL/d eta and v of its squared norm in the Fisher metric, <g, dL/d theta> = g^T (d eta / d theta) g.  The reference
gets dL/d theta from a second tape pass; here the Fisher-vector product is the directional derivative of the expectation
parameters, (eta(theta + e g) - eta(theta - e g)) / 2e, from two extra factor + selected-inverse sweeps.  The reference has no
test for this variant (parity unpinned); tests check the update against the same formulas on the oracle.
"""
import torch

from ._lib import FULL, SYM, VEC
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

    def __init__(self, gamma=0.1, momentum=True, beta1=0.9, beta2=0.99, epsilon=1e-8, name=None):
        self.gamma = float(gamma)
        self._momentum = bool(momentum)
        self._beta1, self._beta2, self._epsilon = float(beta1), float(beta2), float(epsilon)
        self._ms, self._v, self._step_counter = None, 0.0, 1.0      # ssm_natgrad.py:88-93, 155-168
        self._effective_lr = None

    def minimize(self, loss_fn, ssm: StateSpaceModel):
        """One natural-gradient step on `ssm` in place (ssm_natgrad.py:95-119)."""
        self._natgrad_step(loss_fn, ssm)

    def _natgrad_step(self, loss_fn, ssm: StateSpaceModel):
        """theta <- theta - gamma dL/d eta, then back to SSM parameters (ssm_natgrad.py:121-218)."""
        pl = ssm.plan
        if hasattr(loss_fn, "grad_wrt_expectations"):
            (gl, gd, gs), nq = loss_fn.grad_wrt_expectations(ssm)         # closed form for the losses on the path
        else:
            # an arbitrary closure loss_fn(q) of a differentiable view q of the model (vidp_amd.tape.TapeSSM): the reference's
            # persistent-tape route (ssm_natgrad.py:142-172), gradients through the sweeps included
            from . import tape
            _, (g1, g2, g3) = tape.natgrad_wrt_expectations(loss_fn, ssm)
            gl, gd = pl.pack(VEC, g1.contiguous()), pl.pack(SYM, g2.contiguous())
            gs = pl.pack(FULL, g3.contiguous()) if ssm.T > 1 else pl.zeros(FULL)
            nq = pl.ssm_to_naturals(ssm.packed.A, ssm.packed.off, ssm.packed.chol)
        step, dl, dd, ds = self.gamma, gl, gd, gs
        if self._momentum:
            # ssm_natgrad.py:177-208
            b1, b2 = self._beta1, self._beta2
            lr = self.gamma * (1.0 - b2 ** self._step_counter) ** 0.5 / (1.0 - b1 ** self._step_counter)
            if self._ms is None:
                self._ms = [torch.zeros_like(g) for g in (gl, gd, gs)]
            for m, g in zip(self._ms, (gl, gd, gs)):
                pl.lincomb(m, b1, m, 1.0 - b1, g)
            self._v = self._v * b2 + (1.0 - b2) * self.natgrad_norm(pl, nq, (gl, gd, gs))
            self._step_counter += 1.0
            step = lr / (self._v ** 0.5 + self._epsilon)
            self._effective_lr = step
            dl, dd, ds = self._ms
        tl = pl.lincomb(pl.empty(VEC), 1.0, nq["lin"], -step, dl)
        td = pl.lincomb(pl.empty(SYM), 1.0, nq["diag"], -step, dd)
        ts = pl.lincomb(pl.empty(FULL), 1.0, nq["sub"], -step, ds)
        new = naturals_to_ssm_params_packed(pl, tl, td, ts)
        # assign in place (ssm_natgrad.py:213-218)
        keep_batch = ssm.batch_shape
        ssm.__dict__.update(new.__dict__)
        ssm.batch_shape = keep_batch

    @staticmethod
    def natgrad_norm(pl, nq, g, rel_step=1e-5):
        """
        <g, F g> with F = d eta / d theta the Fisher matrix of q in natural coordinates (the reference's
        sum(dL/d eta * dL/d theta), sub-diagonal part counted twice, ssm_natgrad.py:189-192), summed over all chains.
        Central difference of eta along g: two extra factor + selected-inverse passes.
        """
        gl, gd, gs = g
        T = pl.T
        # norms over the nodes of the chains only: packed arrays carry uninitialised padding lanes
        vl, vd, vs = pl.unpack(VEC, gl), pl.unpack(SYM, gd), pl.unpack(FULL, gs, T - 1)
        tnorm = max(float(pl.unpack(SYM, nq["diag"]).abs().max()), 1e-300)
        gnorm = max(float(vd.abs().max()), float(vs.abs().max()) if vs.numel() else 0.0, float(vl.abs().max()), 1e-300)
        e = rel_step * tnorm / gnorm
        etas = []
        for sgn in (1.0, -1.0):
            tl = pl.lincomb(pl.empty(VEC), 1.0, nq["lin"], sgn * e, gl)
            td = pl.lincomb(pl.empty(SYM), 1.0, nq["diag"], sgn * e, gd)
            ts = pl.lincomb(pl.empty(FULL), 1.0, nq["sub"], sgn * e, gs)
            f = pl.factor(td, ts, tl, aD=-2.0, aS=-1.0, aR=1.0, want_logdet=False)
            s_ = pl.selinv(f["L"], f["G"], f["y"], want_sub=True)
            pl.check_info()
            mu, cov, sub = pl.unpack(VEC, s_["x"]), pl.unpack(SYM, s_["Sig"]), pl.unpack(FULL, s_["Sub"], T - 1)
            etas.append((mu, cov + mu[..., :, None] * mu[..., None, :], sub + mu[:, 1:, :, None] * mu[:, :-1, None, :]))
        (m1, d1, s1), (m0, d0, s0) = etas
        tot = (vl * (m1 - m0)).sum() + (vd * (d1 - d0)).sum() + 2.0 * (vs * (s1 - s0)).sum()
        return float(tot) / (2.0 * e)
