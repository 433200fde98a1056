"""
Host-side mirror of markovflow/models/variational_cvi.py (`GaussianProcessWithSitesBase`, `CVIGaussianProcess`,
`back_project_nats`, `gradient_transformation_mean_var_to_expectation`; variational_cvi.py:32-462) and of
`GaussianProcessRegression` (models/gaussian_process_regression.py:118-152, the known-answer oracle of the reference's
CVI tests).  Sites live on f = H s (one scalar site per data point); the posterior is prior naturals + back-projected
sites, refreshed by the HIP sweeps.
"""
import torch

from ._lib import FULL, SYM, VEC
from .kalman_filter import GaussianSitesNat, KalmanFilter, KalmanFilterWithSites
from .ssm_gaussian_transformations import naturals_to_ssm_params_packed


def back_project_nats(nat1, nat2, C):
    """[theta_g1, theta_g2] = [theta_f1 C, theta_f2 C^T C] (variational_cvi.py:423-445): nat1, nat2 [N, 1], C [N, 1, D]."""
    if nat1.shape[-1] != 1 or nat2.shape[-1] != 1 or C.shape[-2] != 1:
        raise ValueError("back_project_nats expects nat1 [N, 1], nat2 [N, 1] and C [N, 1, D]")
    bp1 = (C * nat1[..., None]).sum(dim=-2)
    bp2 = (nat2[..., None, None] * C[..., None] * C[..., None, :]).sum(dim=-3)
    return bp1, bp2


def gradient_transformation_mean_var_to_expectation(inputs, grads):
    """Gradient wrt [mu, sigma^2] -> gradient wrt [mu, sigma^2 + mu^2] (variational_cvi.py:448-462)."""
    if grads[1].dim() == 2:
        return grads[0] - 2.0 * grads[1] * inputs[0], grads[1]
    return grads[0] - 2.0 * (grads[1] @ inputs[0][..., None])[:, :, 0], grads[1]


class GaussianProcessRegression:
    """GPR by Kalman filtering (gaussian_process_regression.py:118-152): log_likelihood() is the exact log marginal likelihood."""

    def __init__(self, input_data, kernel, chol_obs_covariance=None, mean_function=None):
        self._time_points, self._observations = input_data
        self._kernel = kernel
        if chol_obs_covariance is None:
            chol_obs_covariance = torch.zeros((1, 1), dtype=torch.float64, device=self._observations.device)
        self._chol_obs_covariance = chol_obs_covariance

    @property
    def _kalman(self):
        ssm = self._kernel.state_space_model(self._time_points)
        return KalmanFilter(ssm, self._kernel.generate_emission_model(self._time_points), self._observations,
                            self._chol_obs_covariance)

    def log_likelihood(self):
        return self._kalman.log_likelihood()

    @property
    def posterior_state_space_model(self):
        return self._kalman.posterior_state_space_model()


class GaussianProcessWithSitesBase:
    """variational_cvi.py:32-222."""

    def __init__(self, input_data, kernel, likelihood, mean_function=None):
        self._time_points, self._observations = input_data
        if self._observations.shape[-1] != 1:
            raise ValueError("sites are univariate: observation_dim must be 1")
        self._kernel = kernel
        self._likelihood = likelihood
        y = self._observations
        # nat1 = 0, nat2 = -1e-10 (variational_cvi.py:99-103)
        self.sites = GaussianSitesNat(torch.zeros_like(y), torch.full(tuple(y.shape) + (1,), -1e-10, dtype=y.dtype, device=y.device))
        self._dist_p = None
        self._cache = None

    @property
    def time_points(self):
        return self._time_points

    @property
    def observations(self):
        return self._observations

    @property
    def kernel(self):
        return self._kernel

    @property
    def likelihood(self):
        return self._likelihood

    @property
    def dist_p(self):
        """Prior Gauss-Markov distribution at the data points (variational_cvi.py:212-216); cached: the kernel is fixed here."""
        if self._dist_p is None:
            self._dist_p = self._kernel.state_space_model(self._time_points)
        return self._dist_p

    def _emission(self):
        if getattr(self, "_em", None) is None:       # the time points are fixed: one emission model per model object
            self._em = self._kernel.generate_emission_model(self._time_points)
        return self._em

    def _posterior_naturals(self):
        """theta = prior naturals + back-projected sites (variational_cvi.py:106-135), packed."""
        ssm = self.dist_p
        pl = ssm.plan
        pk = ssm.packed
        nat = pl.ssm_to_naturals(pk.A, pk.off, pk.chol, precision=False)
        H = self._emission().emission_matrix
        bp1, bp2 = back_project_nats(self.sites.nat1, self.sites.nat2[..., 0], H)
        B, T, d = ssm.B, ssm.T, ssm.d
        lin = pl.pack(VEC, bp1.expand(ssm.batch_shape + (T, d)).reshape(B, T, d).contiguous())
        # the reference uses bp_nat1 alone as theta_linear (variational_cvi.py:124-127): the prior mean is zero here
        diag = pl.pack(SYM, bp2.expand(ssm.batch_shape + (T, d, d)).reshape(B, T, d, d).contiguous())
        pl.lincomb(diag, 1.0, diag, 1.0, nat["diag"])
        return pl, lin, diag, nat["sub"]

    @property
    def dist_q(self):
        pl, lin, diag, sub = self._posterior_naturals()
        q = naturals_to_ssm_params_packed(pl, lin, diag, sub)
        q.batch_shape = self.dist_p.batch_shape
        return q

    @property
    def posterior_kalman(self):
        return KalmanFilterWithSites(self.dist_p, self._emission(), self.sites)

    def log_likelihood(self):
        return self.posterior_kalman.log_likelihood()

    def loss(self):
        return -self.log_likelihood()

    def predict_f_at_data(self):
        """posterior.predict_f(self.time_points): at the conditioning points this is (H mu, H Sigma H^T) of dist_q."""
        fused = self._predict_f_fused()
        if fused is not None:
            return fused
        pl, lin, diag, sub = self._posterior_naturals()
        f = pl.factor(diag, sub, lin, aD=-2.0, aS=-1.0, aR=1.0, want_logdet=False)
        s = pl.selinv(f["L"], f["G"], f["y"], want_sub=False)
        pl.check_info()
        ssm = self.dist_p
        mu = pl.unpack(VEC, s["x"]).reshape(ssm.batch_shape + (ssm.T, ssm.d))
        cov = pl.unpack(SYM, s["Sig"]).reshape(ssm.batch_shape + (ssm.T, ssm.d, ssm.d))
        em = self._emission()
        return em.project_state_to_f(mu), em.project_state_covariance_to_f(cov, full_output_cov=False)


def _predict_f_fused(self):
    """predict_f at the data points through mfgm_kf_sites_predict (time-invariant emission, state_dim <= 8): assembly, sweeps and the
    projection onto f in one library call; None when the model does not qualify."""
    import ctypes
    from . import _lib
    from .kalman_filter import _kf_scratch, fused_sites_call
    from .packed import _ptr, _stream
    ssm = self.dist_p
    call = fused_sites_call(ssm, self._emission(), self.sites)
    if call is None:
        return None
    sv, keep, cache = call
    pl = ssm.plan
    pr = ssm._precision_packed()
    b = _kf_scratch(cache, pl, ("D", "r", "L", "y", "Sig", "x"))
    o = sv.o
    Fmu = torch.empty((ssm.B, ssm.T, o), dtype=torch.float64, device=pl.device)
    Fvar = torch.empty_like(Fmu)
    fo = cache.get("factor_of")
    if (fo is not None and fo[0] is keep[0] and fo[1] == keep[0]._version and fo[2] is keep[1] and fo[3] == keep[1]._version
            and fo[4] == pl.epoch):
        # elbo() has just factorised exactly this system (same sites, zero-mean prior, nothing else ran on the plan since): the
        # selected inverse and the projection are all that is left (variational_cvi.py:351-379 calls the two back to back)
        _lib.check(pl.lib.mfgm_kf_sites_predict_factored(pl.h, ctypes.byref(sv), _ptr(pr["sub"]), _ptr(b["L"]), _ptr(b["y"]), _ptr(b["Sig"]),
                                                         _ptr(b["x"]), _ptr(Fmu), _ptr(Fvar), _ptr(pl.ws), _stream()),
                   "mfgm_kf_sites_predict_factored")
    else:
        pl.epoch += 1
        _lib.check(pl.lib.mfgm_kf_sites_predict(pl.h, ctypes.byref(sv), _ptr(pr["diag"]), _ptr(pr["sub"]),
                                                None if cache["zero_mean"] else _ptr(pr["lin"]), _ptr(b["D"]), _ptr(b["r"]), _ptr(b["L"]),
                                                _ptr(b["y"]), _ptr(b["Sig"]), _ptr(b["x"]), _ptr(Fmu), _ptr(Fvar), _ptr(pl.ws),
                                                _ptr(pl.info), _stream()), "mfgm_kf_sites_predict")
        cache["factor_of"] = (keep[0], keep[0]._version, keep[1], keep[1]._version, pl.epoch) if cache["zero_mean"] else None
    shp = ssm.batch_shape + (ssm.T, o)
    return Fmu.reshape(shp), Fvar.reshape(shp)


GaussianProcessWithSitesBase._predict_f_fused = _predict_f_fused


class CVIGaussianProcess(GaussianProcessWithSitesBase):
    """variational_cvi.py:225-421."""

    def __init__(self, input_data, kernel, likelihood, mean_function=None, learning_rate=0.1):
        super().__init__(input_data, kernel, likelihood, mean_function)
        self.learning_rate = learning_rate

    def local_objective(self, Fmu, Fvar, Y):
        return self._likelihood.variational_expectations(Fmu, Fvar, Y)

    def local_objective_and_gradients(self, Fmu, Fvar):
        """Local objective and its gradient wrt [mu, sigma^2 + mu^2] (variational_cvi.py:332-349), closed form per likelihood."""
        obj = self.local_objective(Fmu, Fvar, self._observations).sum()
        return obj, self._likelihood.ve_gradients_expectation(Fmu, Fvar, self._observations)

    def update_sites(self):
        """theta <- (1 - rho) theta + rho g (variational_cvi.py:351-368)."""
        fx_mus, fx_covs = self.predict_f_at_data()
        # the gradients alone: the reference's local_objective_and_gradients also returns the objective value, which update_sites drops
        # (:360-362) -- half a dozen element-wise launches and a 100k-element reduction per step here
        grads = self._likelihood.ve_gradients_expectation(fx_mus, fx_covs, self._observations)
        lr = self.learning_rate
        # (1 - lr) theta + lr g assigned to the site variables in place (tf.Variable.assign in the reference, :366-368), one launch
        # (torch._foreach_lerp_ on these two small tensors runs a 25 us multi-tensor kernel: 15 % of the config-2 step)
        n1, n2, g1, g2 = self.sites.nat1, self.sites.nat2, grads[0], grads[1]
        if n1.is_cuda and n1.is_contiguous() and n2.is_contiguous() and g1.shape == n1.shape and g2.numel() == n2.numel():
            from . import _lib
            from .packed import _ptr, _stream
            g1, g2 = g1.contiguous(), g2.contiguous()
            _lib.check(_lib.load().mfgm_site_lerp(_ptr(n1), _ptr(g1), n1.numel(), _ptr(n2), _ptr(g2), n2.numel(), float(lr), _stream()),
                       "mfgm_site_lerp")
            torch.autograd.graph.increment_version(n1)      # written behind torch's back: the factor caches key on ._version
            torch.autograd.graph.increment_version(n2)
        else:
            torch._foreach_lerp_([n1, n2], [g1, g2[..., None]], lr)

    def elbo(self):
        """The marginal likelihood of the model whose likelihood terms are the Gaussian sites (variational_cvi.py:370-379)."""
        return self.log_likelihood()

    def classic_elbo_tape(self, nat1=None, nat2=None):
        """classic_elbo as a differentiable function of the site parameters (leaves of a torch graph; default: detached copies of the
        model's sites with requires_grad) -- the gradient the reference takes with a GradientTape over `trainable_variables` in
        tests/integration/models/test_variational_cvi.py:104-110 (kernel and likelihood frozen there).  Returns (elbo, (nat1, nat2)).
        One chain per batch entry; the posterior naturals  prior + back-projected sites  go through vidp_amd.tape."""
        from . import tape
        ssm = self.dist_p
        pl, B, T, d = ssm.plan, ssm.B, ssm.T, ssm.d
        if pl.d != d or d > 8:
            raise NotImplementedError("the tape route runs on the lane-per-segment plans (state dimension <= 8)")
        n1 = (self.sites.nat1.detach().clone() if nat1 is None else nat1).requires_grad_(True)
        n2 = (self.sites.nat2.detach().clone() if nat2 is None else nat2).requires_grad_(True)
        pk = ssm.packed
        nat = pl.ssm_to_naturals(pk.A, pk.off, pk.chol, precision=False)
        plin, pdiag = pl.unpack(VEC, nat["lin"]), pl.unpack(SYM, nat["diag"])
        psub = pl.unpack(FULL, nat["sub"], T - 1)
        H = self._emission().emission_matrix
        bp1, bp2 = back_project_nats(n1, n2[..., 0], H)
        q = tape.TapeNaturals(plin + bp1.expand(ssm.batch_shape + (T, d)).reshape(B, T, d),
                              pdiag + bp2.expand(ssm.batch_shape + (T, d, d)).reshape(B, T, d, d), psub, pl)
        mu, cov = q.marginals
        Hb = H.expand(ssm.batch_shape + tuple(H.shape[-3:])).reshape(B, T, H.shape[-2], d)
        fmu = (Hb @ mu[..., None])[..., 0]
        fvar = torch.diagonal(Hb @ cov @ Hb.transpose(-1, -2), dim1=-2, dim2=-1)
        obs = self._observations.expand(ssm.batch_shape + tuple(self._observations.shape[-2:])).reshape(B, T, -1)
        ve = self._likelihood.variational_expectations(fmu, fvar, obs).sum()
        return ve - q.kl_divergence(ssm).sum(), (n1, n2)

    def classic_elbo_tape_hyper(self, leaves=None):
        """classic_elbo as a differentiable function of the KERNEL's hyper-parameters (the sites held fixed): (elbo, leaves) with leaves
        the kernel's hyperparameter_leaves -- what the reference gets from a GradientTape over the kernel's tf.Variables
        (tests/integration/models/test_variational_cvi.py:93-110 with the kernel not frozen).  One chain; d <= 8."""
        from . import tape
        ssm = self.dist_p
        pl, T, d = ssm.plan, ssm.T, ssm.d
        if ssm.B != 1 or d > 8:
            raise NotImplementedError("the hyper-parameter tape runs on one chain with state dimension <= 8")
        p, leaves = self._kernel.differentiable_ssm(self._time_points, leaves, plan=pl)
        plin, pdiag, psub = p.naturals()
        H = self._emission().emission_matrix
        bp1, bp2 = back_project_nats(self.sites.nat1.detach(), self.sites.nat2.detach()[..., 0], H)
        q = tape.TapeNaturals(plin + bp1.reshape(1, T, d), pdiag + bp2.reshape(1, T, d, d), psub, pl)
        mu, cov = q.marginals
        Hb = H.reshape(1, T, H.shape[-2], d)
        fmu = (Hb @ mu[..., None])[..., 0]
        fvar = torch.diagonal(Hb @ cov @ Hb.transpose(-1, -2), dim1=-2, dim2=-1)
        ve = self._likelihood.variational_expectations(fmu, fvar, self._observations.reshape(1, T, -1)).sum()
        return ve - tape.kl_divergence_tape(q, p).sum(), leaves

    def step_graph(self):
        """`update_sites(); elbo()` -- the inner loop of CVI (variational_cvi.py:351-379) -- captured ONCE in a HIP graph: returns a
        callable that replays it and hands back the ELBO (a device scalar that every replay overwrites).  On one chain the step is a
        sequence of ~20 short dependent launches (level sweeps of a few microseconds each): replayed from a graph they cost no host time
        at all.  The captured order is SELF-CONTAINED: update_sites factorises the system of the current sites itself (it does not
        take the eager shortcut of reusing the factorisation elbo() left in the plan's scratch -- that decision depends on host-side
        stamps a replay cannot re-evaluate), and every replay advances those stamps (site versions, plan epoch) and drops the
        "factorisation belongs to these sites" record, so eager calls and replays can be interleaved in any order."""
        from .kalman_filter import fused_sites_call
        fx_mus, fx_covs = self.predict_f_at_data()
        self._likelihood.ve_gradients_expectation(fx_mus, fx_covs, self._observations)
        self.elbo()
        call = fused_sites_call(self.dist_p, self._emission(), self.sites)
        cache = call[2] if call is not None else {}
        cache["factor_of"] = None                   # capture factor + selected inverse + projection, not the shortcut
        pl = self.dist_p.plan
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.update_sites()
            e = self.elbo()
        cache["factor_of"] = None                   # nothing ran during the capture: the scratch holds no factorisation of these sites

        def step():
            graph.replay()
            # the replay moved the sites and overwrote the plan's scratch behind the host's back
            torch.autograd.graph.increment_version(self.sites.nat1)
            torch.autograd.graph.increment_version(self.sites.nat2)
            pl.epoch += 1
            cache["factor_of"] = None
            return e
        step.graph = graph
        return step

    def classic_elbo(self):
        """sum_i E_q log p(y_i | f_i) - KL[q(s) || p(s)] (variational_cvi.py:381-404)."""
        fx_mus, fx_covs = self.predict_f_at_data()
        ve = self._likelihood.variational_expectations(fx_mus, fx_covs, self._observations).sum()
        kl = self.dist_q.kl_divergence(self.dist_p).sum()
        return ve - kl
