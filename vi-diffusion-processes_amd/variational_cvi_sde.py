"""
Host-side mirror of markovflow/models/variational_cvi_sde.py: `CVISitesSSM` (site-based posterior over the
state trajectory of a linear SSM prior) and `CVISitesSDE` (CVI-DP: non-linear SDE prior, linearised).

Same method names as the reference (`full_sites`, `dist_q`, `update_data_sites`, `update_girsanov_sites`,
`variational_expectation`, `KL_q_p`, `classic_elbo`, `set_linearized_prior`).  Differences, all additive:
  * a leading batch of B independent trajectories is supported (the reference asserts batch_shape == []):
    observations [B, n_obs, d]; every quantity is per trajectory and `classic_elbo()` returns the sum over B
    (the reduction that becomes the RCCL all-reduce when trajectories are sharded over GPUs);
  * all per-time-step state (prior naturals, Girsanov sites, posterior naturals, marginals) lives in the
    packed device layout and is refreshed once per site update, not once per property access.
"""
import math
import os

import numpy as np
import torch

from ._lib import FULL, SYM, VEC
from .packed import CqState, Plan, aligned_segment_length, observation_period
from .state_space_model import StateSpaceModel


def grid_indices(time_grid, obs_times):
    """Indices of the observation times on the grid (variational_cvi_sde.py:108-110: exact equality)."""
    tg = torch.as_tensor(time_grid, dtype=torch.float64).cpu()
    ot = torch.as_tensor(obs_times, dtype=torch.float64).cpu()
    idx = torch.searchsorted(tg, ot)
    idx = idx.clamp(max=tg.numel() - 1)
    if not torch.equal(tg[idx], ot):
        raise ValueError("every observation time must be a point of the time grid")
    return idx


class PackedBTDNat:
    """Natural parameters of a block-tri-diagonal Gaussian in packed form (BTDGaussian, gauss_markov.py:220-242)."""

    def __init__(self, lin, diag, sub):
        self.lin, self.diag, self.sub = lin, diag, sub


class CVISitesSSM:
    """variational_cvi_sde.py:49-366."""

    def __init__(self, prior_ssm, time_grid, input_data, likelihood, prior_initial_state=None,
                 initial_posterior_path=None, plan=None):
        obs_times, observations = input_data
        if observations.dim() == 2:
            observations = observations[None]
        self.likelihood = likelihood
        self.time_grid = torch.as_tensor(time_grid, dtype=torch.float64)
        self.dt = float(self.time_grid[1] - self.time_grid[0])
        self._observations = observations.contiguous()
        self.B, self.n_obs, self.state_dim = observations.shape
        self.output_dim = self.state_dim
        self.T = int(self.time_grid.numel())
        self.device = observations.device
        self.obs_sites_indices = grid_indices(self.time_grid, obs_times).to(self.device)
        if plan is None:
            # segments aligned with an equally spaced observation grid (packed.aligned_segment_length)
            r0 = aligned_segment_length(self.B, self.T, self.state_dim, observation_period(self.obs_sites_indices))
            plan = prior_ssm.plan if prior_ssm is not None else Plan(self.B, self.T, self.state_dim, R0=r0, device=self.device)
        self.plan = plan
        self.obs_node_ids = plan.node_ids(self.obs_sites_indices)
        d, pl = self.state_dim, plan
        # Girsanov sites start at nat1 = 0, nat2 = -1e-10 * ones (variational_cvi_sde.py:141-152).  They are not stored:
        # theta_q = theta_prior + girsanov + scatter(data) is the resident state and the sites are recovered from it on
        # demand (property `girsanov_sites`), which halves the traffic of every Girsanov update.
        self._g_init = True
        # data sites: nat1 = 0, nat2 = +1e-10 * I  (variational_cvi_sde.py:96-103)
        n = self.B * self.n_obs
        self.data_nat1 = torch.zeros((n, d), dtype=torch.float64, device=self.device)
        self.data_nat2 = (1e-10 * torch.eye(d, dtype=torch.float64, device=self.device)).expand(n, d, d).contiguous()
        self.prior_initial_state = prior_initial_state
        # initial posterior path: zero mean, identity covariance (variational_cvi_sde.py:131-139)
        if initial_posterior_path is None:
            self.fx_mus_obs = torch.zeros((n, d), dtype=torch.float64, device=self.device)
            self.fx_covs_obs = torch.eye(d, dtype=torch.float64, device=self.device).expand(n, d, d).contiguous()
            self._path = None
        else:
            mu, cov = initial_posterior_path
            self._path = (pl.pack(VEC, mu.reshape(self.B, self.T, d)), pl.pack(SYM, cov.reshape(self.B, self.T, d, d)))
            self.fx_mus_obs = pl.gather_nodes(VEC, self._path[0], self.obs_node_ids)
            self.fx_covs_obs = pl.gather_nodes(SYM, self._path[1], self.obs_node_ids)
        self._theta_q = None        # dense posterior naturals, allocated when first built
        self._started = False
        self._bufs = dict(f={}, s={})
        self._q = None          # cached posterior refresh (factor + selected inverse) for the current sites
        self.dist_p = None
        if prior_ssm is not None:
            self._set_prior(prior_ssm)

    # -- prior ---------------------------------------------------------------------------------------
    def _set_prior(self, ssm: StateSpaceModel, move_theta_q=True):
        """Cache the prior's natural parameters, marginal means and log-determinant (packed).  With move_theta_q the posterior
        naturals follow the prior (the implicit Girsanov sites stay what they are); without, theta_q stays fixed (the sites absorb
        the change of prior: tranform_girsanov_sites)."""
        if ssm.plan is not self.plan:
            ssm = StateSpaceModel(ssm.initial_mean, ssm.cholesky_initial_covariance, ssm.state_transitions,
                                  ssm.state_offsets, ssm.cholesky_process_covariances, plan=self.plan)
        self.dist_p = ssm
        pk = ssm.packed
        nat = self.plan.ssm_to_naturals(pk.A, pk.off, pk.chol, precision=False, want_logdet=True)
        old_p, had_q = getattr(self, "_theta_p", None), getattr(self, "_theta_q_valid", False)
        self._theta_p = PackedBTDNat(nat["lin"], nat["diag"], nat["sub"])
        if had_q and old_p is not None and move_theta_q:
            # the (implicit) Girsanov sites stay what they are: theta_q moves with the prior
            tq = self._theta_q
            for qq, new_, old_ in ((tq.lin, nat["lin"], old_p.lin), (tq.diag, nat["diag"], old_p.diag), (tq.sub, nat["sub"], old_p.sub)):
                self.plan.lincomb(qq, 1.0, qq, 1.0, new_, -1.0, old_)
        self._p_sumlogchol = nat["sumlogchol"]
        # prior marginal means K theta_lin: one factor + solve of the prior itself
        f = self.plan.factor(nat["diag"], nat["sub"], nat["lin"], aD=-2.0, aS=-1.0, aR=1.0, want_logdet=False)
        s = self.plan.selinv(f["L"], f["G"], f["y"], want_sub=False)
        self._p_mu = s["x"]
        self._q = None
        self._theta_q_valid = bool(had_q and (old_p is not None or not move_theta_q))

    # -- sites -> posterior --------------------------------------------------------------------------------
    def _rebuild_theta_q(self, g=None):
        """theta_q = theta_prior + girsanov sites + scattered data sites (variational_cvi_sde.py:161-174), from scratch."""
        pl, tp = self.plan, self._theta_p
        if self._theta_q is None:
            self._theta_q = PackedBTDNat(pl.empty(VEC), pl.empty(SYM), pl.empty(FULL))
        tq = self._theta_q
        if g is None:   # initial Girsanov sites
            g = PackedBTDNat(pl.zeros(VEC), pl.zeros(SYM).fill_(-1e-10), pl.zeros(FULL).fill_(-1e-10))
        pl.lincomb(tq.lin, 1.0, tp.lin, 1.0, g.lin)
        pl.lincomb(tq.diag, 1.0, tp.diag, 1.0, g.diag)
        pl.lincomb(tq.sub, 1.0, tp.sub, 1.0, g.sub)
        pl.scatter_nodes(VEC, tq.lin, self.obs_node_ids, self.data_nat1, accumulate=True)
        pl.scatter_nodes(SYM, tq.diag, self.obs_node_ids, self.data_nat2, accumulate=True)
        self._theta_q_valid = True

    @property
    def girsanov_sites(self):
        """The Girsanov sites (BTDGaussian in the reference): theta_q - theta_prior - scatter(data sites), packed."""
        pl, tq, tp = self.plan, self.full_sites(), self._theta_p
        g = PackedBTDNat(pl.lincomb(pl.empty(VEC), 1.0, tq.lin, -1.0, tp.lin), pl.lincomb(pl.empty(SYM), 1.0, tq.diag, -1.0, tp.diag),
                         pl.lincomb(pl.empty(FULL), 1.0, tq.sub, -1.0, tp.sub))
        pl.scatter_nodes(VEC, g.lin, self.obs_node_ids, self.data_nat1, accumulate=True, scale=-1.0)
        pl.scatter_nodes(SYM, g.diag, self.obs_node_ids, self.data_nat2, accumulate=True, scale=-1.0)
        return g

    def full_sites(self):
        """
        The posterior natural parameters theta_q (variational_cvi_sde.py:161-174).  They are kept resident and
        updated incrementally by the site updates instead of being re-summed on every access.
        """
        if not getattr(self, "_theta_q_valid", False):
            self._rebuild_theta_q()
        return self._theta_q

    _need_sub = True     # the linear-prior KL (kl_terms) reads the full cross-covariance blocks
    # site update / observation-node variational expectations in one launch each (VIDP_FUSED_OBS=0: torch arithmetic around the
    # sparse gather / scatter kernels)
    fused_obs_kernels = os.environ.get("VIDP_FUSED_OBS", "1") != "0"

    def _sweep_fusion(self):
        """True when the level-0 backward sweep can do the model's local work itself (CVISitesSDE, mfgm_girsanov.h)."""
        return False

    def _refresh(self, want_sub=None, want_mom=None, want_marginals=False):
        """theta_q -> (L, log|L|, mu, Sigma_tt, [Sigma_{t+1,t}], [moments | KL sum]) in one forward and one backward sweep."""
        want_sub = self._need_sub if want_sub is None else want_sub
        fuse = self._sweep_fusion() and not want_sub
        want_mom = (not fuse) if want_mom is None else want_mom
        if self._q is not None and ((want_sub and self._q["Sub"] is None) or (want_mom and self._q["mom"] is None)):
            self._q = None      # cached refresh lacks the cross-covariances / moments now requested
        if self._q is None:
            pl = self.plan
            tq = self.full_sites()
            # without the full cross-covariances the forward sweep need not store L_{t+1,t}: the backward sweep rebuilds what it
            # uses from theta_sub (d^2 doubles per node fewer written)
            lean = (not want_sub) and pl.d <= 8
            f = pl.factor(tq.diag, tq.sub, tq.lin, aD=-2.0, aS=-1.0, aR=1.0, want_logdet=True, out=self._bufs["f"], store_G=not lean)
            self._bufs["f"].update(L=f["L"], y=f["y"])
            if f["G"] is not None:
                self._bufs["f"]["G"] = f["G"]
            if fuse and not want_mom:
                # the KL sum comes out of the backward sweep; the moment array is not written
                s = pl.selinv_kl(f["L"], tq.sub, -1.0, f["y"], self._sde_prm, out=self._bufs["s"])
                self._bufs["s"].update(Sig=s["Sig"], x=s["x"])
                self._q = dict(logdetL=f["logdet"], mu=s["x"], Sig=s["Sig"], Sub=None, mom=None, klpart=s["klpart"])
                return self._q
            s = pl.selinv_mom(f["L"], f["G"], f["y"], want_sub=want_sub, out=self._bufs["s"], S=tq.sub if lean else None, aS=-1.0)
            self._bufs["s"].update(Sig=s["Sig"], x=s["x"], mom=s["mom"])
            if s["Sub"] is not None:
                self._bufs["s"]["Sub"] = s["Sub"]
            self._q = dict(logdetL=f["logdet"], mu=s["x"], Sig=s["Sig"], Sub=s["Sub"], mom=s["mom"], klpart=None)
        return self._q

    @property
    def dist_q_marginals_packed(self):
        q = self._refresh(want_sub=True)
        return q["mu"], q["Sig"], q["Sub"]

    @property
    def dist_q(self) -> StateSpaceModel:
        """The posterior as a StateSpaceModel (variational_cvi_sde.py:177-192), via naturals_to_ssm_params."""
        from .ssm_gaussian_transformations import naturals_to_ssm_params_packed
        tq = self.full_sites()
        return naturals_to_ssm_params_packed(self.plan, tq.lin, tq.diag, tq.sub)

    def _gather_obs(self):
        q = self._refresh(want_marginals=True)
        # persistent buffers, updated in place: the state that crosses iterations keeps its addresses (a captured HIP graph
        # of one iteration can then be replayed)
        self.plan.gather_nodes_pair(q["mu"], q["Sig"], self.obs_node_ids, self.fx_mus_obs, self.fx_covs_obs)
        self._obs_fresh = True

    @property
    def fx_mus(self):
        return self.plan.unpack(VEC, self._refresh(want_marginals=True)["mu"])

    @property
    def fx_covs(self):
        return self.plan.unpack(SYM, self._refresh(want_marginals=True)["Sig"])

    # -- updates -------------------------------------------------------------------------------------------
    def _obs_flat(self):
        # one view object for the model's lifetime: the likelihood keys its gradient cache on the tensor object and its version
        if getattr(self, "_obs_flat_view", None) is None:
            self._obs_flat_view = self._observations.view(self.B * self.n_obs, self.state_dim)
        return self._obs_flat_view

    def _obs_marginals(self):
        """(mu, Sigma) at the observation times for the current sites (the reference's self.fx_mus / fx_covs gathered)."""
        if self._started and not getattr(self, "_obs_fresh", False):
            self._gather_obs()
        return self.fx_mus_obs, self.fx_covs_obs

    def update_data_sites(self, lr: float):
        """theta_data <- (1-lr) theta_data + lr dVE/d(eta) at the current marginals (variational_cvi_sde.py:301-317)."""
        mu_o, cov_o = self._obs_marginals()
        self._started = True      # before the first update the marginals are those of the initial posterior path
        g1, g2 = self.likelihood.ve_gradients_expectation(mu_o, cov_o, self._obs_flat())
        tq = self.full_sites()
        if (self.fused_obs_kernels and self.plan.d <= 8 and g1.shape == self.data_nat1.shape and g2.shape == self.data_nat2.shape
                and self.data_nat1.is_contiguous() and self.data_nat2.is_contiguous()):
            # blend, difference, copy back and scatter in one pass over the site arrays
            self.plan.site_update_pair(tq.lin, tq.diag, self.obs_node_ids, self.data_nat1, self.data_nat2, g1.contiguous(),
                                       g2.contiguous(), lr)
        else:
            new1 = (1 - lr) * self.data_nat1 + lr * g1
            new2 = (1 - lr) * self.data_nat2 + lr * g2
            self.plan.scatter_nodes_pair(tq.lin, tq.diag, self.obs_node_ids, new1 - self.data_nat1, new2 - self.data_nat2)
            self.data_nat1.copy_(new1)
            self.data_nat2.copy_(new2)
        self._q = None
        self._obs_fresh = False   # marginals at the observation times are gathered lazily, when next needed

    def grad_kl_wrt_exp_param(self):
        """
        d KL[q || p] / d(eta) for a linear prior is theta_q - theta_p (natural-gradient identity; the reference's
        SSM_KL_with_grads_wrt_exp_params, sde_utils.py:376-461, differentiates a quadrature of the same KL).
        """
        tq, tp = self.full_sites(), self._theta_p
        pl = self.plan
        return (pl.lincomb(pl.empty(VEC), 1.0, tq.lin, -1.0, tp.lin), pl.lincomb(pl.empty(SYM), 1.0, tq.diag, -1.0, tp.diag),
                pl.lincomb(pl.empty(FULL), 1.0, tq.sub, -1.0, tp.sub))

    def update_girsanov_sites(self, lr: float):
        """
        g <- g + lr (scatter(data sites) - dKL/d eta) (variational_cvi_sde.py:279-299), with dKL/d eta = theta_q - theta_p.
        Sites and posterior naturals are updated together: theta_q <- theta_q + (g_new - g).
        """
        pl, tq, tp = self.plan, self.full_sites(), self._theta_p
        for qq, pp in ((tq.lin, tp.lin), (tq.diag, tp.diag), (tq.sub, tp.sub)):
            pl.lincomb(qq, 1.0 - lr, qq, lr, pp)               # theta_q += lr (theta_p - theta_q)
        pl.scatter_nodes_pair(tq.lin, tq.diag, self.obs_node_ids, self.data_nat1, self.data_nat2, scale=lr)
        self._q = None
        self._obs_fresh = False
        self._started = True

    # -- objective -----------------------------------------------------------------------------------------
    def variational_expectation(self):
        """sum_i E_q log p(y_i | x_i), per trajectory [B] (variational_cvi_sde.py:319-337)."""
        lik = self.likelihood
        if self.fused_obs_kernels and self.plan.d <= 8 and hasattr(lik, "inv_covariance") and hasattr(lik, "log_det_chol"):
            # multivariate Gaussian likelihood: gather, element-wise arithmetic and per-trajectory sum in one launch (it also
            # refreshes the gathered marginals)
            q = self._refresh()
            ve = self.plan.mvn_obs_ve(q["mu"], q["Sig"], self.obs_node_ids, self.n_obs, self._obs_flat(), lik.inv_covariance,
                                      lik.ve_constant, out_mu=self.fx_mus_obs, out_cov=self.fx_covs_obs)
            self._obs_fresh = True
            return ve
        if self._q is None or not getattr(self, "_obs_fresh", False):
            self._gather_obs()
        ve = lik.variational_expectations(self.fx_mus_obs, self.fx_covs_obs, self._obs_flat())
        return ve.reshape(self.B, self.n_obs).sum(-1)

    def KL_q_p(self):
        """KL[q || p] per trajectory [B]; exact Gauss-Markov KL against the linear prior (state_space_model.py:528-593)."""
        q = self._refresh()
        tp = self._theta_p
        tr, mh = self.plan.kl_terms(q["Sig"], q["Sub"], q["mu"], tp.diag, tp.sub, self._p_mu, aD=-2.0, aS=-1.0)
        dim = float(self.T * self.state_dim)
        # log det P_p = -2 sumlogchol_p ; log det P_q = 2 log|L_q|
        return 0.5 * (tr + mh - dim + 2.0 * self._p_sumlogchol + 2.0 * q["logdetL"])

    def classic_elbo_per_trajectory(self):
        return self.variational_expectation() - self.KL_q_p()

    def classic_elbo(self):
        """E_q[log p(Y|X)] - KL[q || p], summed over trajectories (variational_cvi_sde.py:339-352)."""
        return self.classic_elbo_per_trajectory().sum()


class CVISitesSDE(CVISitesSSM):
    """
    CVI-DP (variational_cvi_sde.py:368-518): non-linear SDE prior, linearised along the current posterior path;
    KL[q || p_SDE] and its gradient with respect to the expectation parameters in closed form (HIP kernel k_sde_kl).
    `prior_initial_state` is a (mean [d], covariance [d, d]) pair; default N(0, q) as in the reference (:437-444).
    """

    def __init__(self, prior_sde, time_grid, input_data, likelihood, prior_initial_state=None, initial_posterior_path=None,
                 stabilize_ssm=True, clip_state_transitions=(-1.0, 1.0), plan=None):
        self.prior_sde = prior_sde
        d = input_data[1].shape[-1]
        if prior_initial_state is None:
            q = prior_sde.q.cpu().numpy()
            prior_initial_state = (torch.zeros(d, dtype=torch.float64).numpy(), q * (torch.ones((d, d), dtype=torch.float64).numpy()))
        self.stabilize_ssm = stabilize_ssm
        self.clip_state_transitions = clip_state_transitions
        self._cq, self._cq_off, self._cq_dense, self._cq_p0_moved = None, False, None, False
        super().__init__(None, time_grid, input_data, likelihood, prior_initial_state=prior_initial_state,
                         initial_posterior_path=initial_posterior_path, plan=plan)
        self._sde_prm = prior_sde.params(self.dt, prior_initial_state[0], prior_initial_state[1])
        self.dist_p_linearized = None
        self.set_linearized_prior()

    def _path_packed(self):
        """Current posterior path (mu, Sigma) in packed form: the initial path before the first refresh."""
        if self._q is not None:
            q = self._refresh(want_marginals=True)      # the cq refresh leaves the marginal arrays out until someone asks
            return q["mu"], q["Sig"]
        if self._path is not None:
            return self._path
        pl, d = self.plan, self.state_dim
        mu = pl.zeros(VEC)
        eye = torch.eye(d, dtype=torch.float64, device=self.device).expand(self.B, self.T, d, d).contiguous()
        self._path = (mu, pl.pack(SYM, eye))
        return self._path

    def set_linearized_prior(self, move_theta_q=True):
        """Linearise the SDE on the current posterior (variational_cvi_sde.py:408-432) and install it as dist_p."""
        pl = self.plan
        mu, Sig = self._path_packed()
        prm = self.prior_sde.params(self.dt, self.prior_initial_state[0], self.prior_initial_state[1], clip=None)
        A, off, chol = pl.linearize_cubic(prm, mu, Sig)
        self.dist_p_linearized = _ssm_from_packed(pl, A, off, chol)
        if self.stabilize_ssm:
            prm = self.prior_sde.params(self.dt, self.prior_initial_state[0], self.prior_initial_state[1],
                                        clip=self.clip_state_transitions)
            A, off, chol = pl.linearize_cubic(prm, mu, Sig)
            self._set_prior(_ssm_from_packed(pl, A, off, chol), move_theta_q=move_theta_q)
        else:
            self._set_prior(self.dist_p_linearized, move_theta_q=move_theta_q)

    def relinearize(self):
        """
        Re-linearise on the current posterior and move the Girsanov sites to the new prior so that the posterior is
        unchanged: the trainer's sequence `dist_p_last = dist_p; set_linearized_prior(); tranform_girsanov_sites(...)`
        (docs/diffusion_processes/cvi_dp_trainer.py:127-134, sde_utils.py:550-568).
        """
        # tranform_girsanov_sites adds theta(old prior) - theta(new prior) to the sites, i.e. it keeps
        # theta_q = theta_p + g + data fixed.  The sites are implicit here (theta_q - theta_p - data), so leaving
        # theta_q alone under the new prior IS the transformation and the cached posterior stays valid.
        q_valid = self._q
        if self._cq is None:
            self.full_sites()         # theta_q exists (and is marked valid) before the prior is replaced
        self.set_linearized_prior(move_theta_q=False)
        self._q = q_valid

    # ---- structured ("cq") posterior naturals: csrc/mfgm_cq.h -------------------------------------------------------------------------
    # With a per-dimension drift, diagonal diffusion and a likelihood whose site gradient is one block shared by all observations
    # (MultivariateGaussian), theta_prior + Girsanov sites is, per node, 3 d numbers plus uniform off-diagonals; the data sites stay
    # in their own small arrays and the sweeps add them on load.  VIDP_CQ=0 keeps the dense arrays (the two routes are held to each
    # other in tests/test_gpu_api.py).
    cq_enabled = os.environ.get("VIDP_CQ", "1") != "0"

    def _cq_eligible(self):
        return (self.cq_enabled and not self._cq_off and self._sweep_fusion()
                and getattr(self.likelihood, "uniform_site_gradient", False))

    @property
    def data_nat2(self):
        """Data-site nat2, [B n_obs, d, d] (in cq mode one block shared by every observation, expanded on demand)."""
        if self._cq is not None:
            return self._sym_full(self._cq.site_sym).expand(self.B * self.n_obs, -1, -1).contiguous()
        return self._data_nat2

    @data_nat2.setter
    def data_nat2(self, value):
        if getattr(self, "_cq", None) is not None:
            self._cq_leave()
        self._data_nat2 = value

    def _sym_full(self, packed):
        d = self.state_dim
        r, c = torch.tril_indices(d, d, device=packed.device)
        m = torch.zeros((d, d), dtype=torch.float64, device=packed.device)
        m[r, c] = packed
        m[c, r] = packed
        return m

    def _sym_packed(self, full):
        d = self.state_dim
        r, c = torch.tril_indices(d, d, device=full.device)
        return full[r, c].contiguous()

    def _cq_try_enter(self, lin, diag, sub):
        """Move the dense, DATA-FREE naturals (theta_prior + Girsanov sites) into the cq state.  Returns False (and switches the cq
        route off for this model) when they do not have the structure: off-diagonals not uniform, observations sharing a node, data
        sites that differ between observations."""
        pl, d = self.plan, self.state_dim
        n2 = self._data_nat2
        if n2.numel() and float((n2 - n2[:1]).abs().max()) > 0.0:
            self._cq_off = True
            return False
        slot = pl.cq_slots(self.obs_node_ids) if n2.numel() else None
        if n2.numel() and slot is None:
            self._cq_off = True
            return False
        dyn, (dlo, dhi), (slo, shi) = pl.cq_pack(lin, diag, sub)
        tol = lambda lo, hi: hi - lo <= 1e-12 * max(abs(lo), abs(hi), 1e-300)
        # node 0 of every chain: off-diagonal part beyond the uniform value = that of theta_prior (-1/2 P0^{-1}), the same for all chains
        blk0 = pl.gather_nodes(SYM, diag, pl.node_ids([0]))
        p0 = blk0 - torch.diag_embed(torch.diagonal(blk0, dim1=-2, dim2=-1))
        off = (1.0 - torch.eye(d, dtype=torch.float64, device=self.device))
        p0 = p0 - dlo * off
        if not (tol(dlo, dhi) and tol(slo, shi)) or float((p0 - p0[:1]).abs().max()) > 1e-9 * max(1.0, float(p0.abs().max())):
            self._cq_off = True
            return False
        cq = CqState(dyn, dlo, slo, p0_off=self._sym_packed(p0[0]))
        if n2.numel():
            cq.slot, cq.site_lin, cq.site_sym = slot, self.data_nat1, self._sym_packed(n2[0])
        self._cq, self._cq_dense = cq, None
        self._theta_q_valid = True
        self._theta_q = None          # the dense arrays are materialised on demand only
        return True

    def _cq_dense_free(self):
        """Dense packed (lin, diag, sub) of theta_prior + Girsanov sites (no data sites) from the cq state."""
        return self.plan.cq_unpack(self._cq)

    def _cq_leave(self):
        """Back to the dense arrays (a caller assigned sites the structured state cannot hold)."""
        cq, pl = self._cq, self.plan
        lin, diag, sub = self._cq_dense_free()
        if cq.site_sym is not None:
            self._data_nat2 = self._sym_full(cq.site_sym).expand(self.B * self.n_obs, -1, -1).contiguous()
        self._cq, self._cq_dense = None, None
        if self._data_nat2.numel():
            pl.scatter_nodes(VEC, lin, self.obs_node_ids, self.data_nat1, accumulate=True)
            pl.scatter_nodes(SYM, diag, self.obs_node_ids, self._data_nat2, accumulate=True)
        self._theta_q = PackedBTDNat(lin, diag, sub)
        self._theta_q_valid = True

    def _rebuild_theta_q(self, g=None):
        """theta_q from the prior, the Girsanov sites `g` (default: the initial ones) and the data sites; into the cq state when the
        model qualifies, else into the dense arrays (variational_cvi_sde.py:161-174)."""
        if self._cq is not None:
            self._cq, self._cq_dense = None, None
        if not self._cq_eligible():
            return super()._rebuild_theta_q(g)
        pl, tp = self.plan, self._theta_p
        if g is None:
            g = PackedBTDNat(pl.zeros(VEC), pl.zeros(SYM).fill_(-1e-10), pl.zeros(FULL).fill_(-1e-10))
        free = PackedBTDNat(pl.lincomb(pl.empty(VEC), 1.0, tp.lin, 1.0, g.lin), pl.lincomb(pl.empty(SYM), 1.0, tp.diag, 1.0, g.diag),
                            pl.lincomb(pl.empty(FULL), 1.0, tp.sub, 1.0, g.sub))
        if self._cq_try_enter(free.lin, free.diag, free.sub):
            return
        pl.scatter_nodes(VEC, free.lin, self.obs_node_ids, self.data_nat1, accumulate=True)
        pl.scatter_nodes(SYM, free.diag, self.obs_node_ids, self._data_nat2, accumulate=True)
        self._theta_q, self._theta_q_valid = free, True

    def full_sites(self):
        """The posterior natural parameters theta_q, dense and packed (variational_cvi_sde.py:161-174); in cq mode they are
        materialised from the structured state (and cached until the next update)."""
        if not getattr(self, "_theta_q_valid", False):
            self._rebuild_theta_q()
        if self._cq is None:
            return self._theta_q
        if self._cq_dense is None:
            pl = self.plan
            lin, diag, sub = self._cq_dense_free()
            if self._cq.slot is not None:
                pl.scatter_nodes(VEC, lin, self.obs_node_ids, self.data_nat1, accumulate=True)
                pl.scatter_nodes(SYM, diag, self.obs_node_ids, self.data_nat2, accumulate=True)
            self._cq_dense = PackedBTDNat(lin, diag, sub)
        return self._cq_dense

    def _cq_state(self):
        """The cq state, built on first use; None when the model runs on the dense arrays."""
        if not getattr(self, "_theta_q_valid", False):
            self._rebuild_theta_q()
        return self._cq

    def _set_prior(self, ssm, move_theta_q=True):
        cq = self._cq
        if cq is None:
            return super()._set_prior(ssm, move_theta_q=move_theta_q)
        if not move_theta_q:
            # theta_q stays what it is: only the prior-side caches are replaced
            self._theta_q_valid = False
            super()._set_prior(ssm, move_theta_q=False)
            self._theta_q_valid = True
            return
        # theta_q follows the prior: through the dense data-free arrays (a rare call: prior-parameter gradients)
        lin, diag, sub = self._cq_dense_free()
        if cq.site_sym is not None:
            self._data_nat2 = self._sym_full(cq.site_sym).expand(self.B * self.n_obs, -1, -1).contiguous()
        self._theta_q, self._theta_q_valid, self._cq, self._cq_dense = PackedBTDNat(lin, diag, sub), True, None, None
        super()._set_prior(ssm, move_theta_q=True)
        tq = self._theta_q
        keep_sites = (cq.slot, cq.site_lin, cq.site_sym)
        if not self._cq_try_enter(tq.lin, tq.diag, tq.sub):
            if self._data_nat2.numel():
                self.plan.scatter_nodes(VEC, tq.lin, self.obs_node_ids, self.data_nat1, accumulate=True)
                self.plan.scatter_nodes(SYM, tq.diag, self.obs_node_ids, self._data_nat2, accumulate=True)
            self._theta_q, self._theta_q_valid = tq, True
        else:
            self._cq.slot, self._cq.site_lin, self._cq.site_sym = keep_sites

    # the level-0 backward sweep makes the Girsanov-site update / the KL sum itself (VIDP_FUSED_GIRSANOV=0: separate kernels on
    # the moment array)
    fused_girsanov = os.environ.get("VIDP_FUSED_GIRSANOV", "1") != "0"

    def _sweep_fusion(self):
        pl, prm = self.plan, getattr(self, "_sde_prm", None)
        return self.fused_girsanov and prm is not None and pl.d <= 8 and pl.nlevels >= 2 and prm.kind == 0

    _need_sub = False    # the closed-form SDE KL needs only (mu, diag Sigma, diag Sigma_sub): the moment array

    # Pipelining across steps (VIDP_PIPELINE=0 switches it off).  The loop of cvi_dp_trainer.py:72-75 is update_data_sites ->
    # update_girsanov_sites -> classic_elbo, two factorisations of theta_q per step.  Under a Gaussian likelihood the data sites a step
    # ends up with do not depend on q, so the state the FIRST factorisation of the next step will see (dyn as it is after this step's
    # Girsanov update, the sites one more blend ahead) is known while this step's SECOND factorisation runs: its level-0 reduce -- an
    # arithmetic-bound kernel on 18 doubles per node -- is launched on a second stream next to that factorisation's bandwidth-bound
    # forward sweep (Plan.cq_factor(next_sites=...)), and the next step starts from the separator system it left.  The prediction
    # assumes the learning rate of the last update_data_sites; it is keyed on the identity / version of the cq state and of the site
    # tensors, and anything that does not match (another learning rate, a re-linearisation, sites assigned by hand) simply finds no
    # record and runs the reduce itself.  Results are bit-identical either way (the record holds the numbers the reduce would write).
    pipelined = os.environ.get("VIDP_PIPELINE", "1") != "0"
    # VIDP_PIPE_STREAMS=1: the reduce made ahead runs as a kernel of its own on a second stream (round 4's first form; A/B) instead of
    # as the second wavefront of the forward sweep's workgroups (k_forward_reduce_cq: the records are read once)
    pipe_two_streams = os.environ.get("VIDP_PIPE_STREAMS", "0") == "1"
    # below this many nodes (B T) a level-0 reduce is a few microseconds: the second stream's event round trips would cost more than
    # they hide (config 1, T = 1001: 0.133 -> 0.174 ms with it)
    pipeline_min_nodes = 200000
    _pre = None           # the record of a separator system made ahead (dict), None when there is none

    def _pipe_sites_gradient(self):
        g1, g2 = self.likelihood.ve_gradients_expectation(self.fx_mus_obs, self.fx_covs_obs, self._obs_flat())
        if getattr(self, "_cq_g2", (None,))[0] is not g2:
            self._cq_g2 = (g2, self._sym_packed(g2[0]))
        return g1, self._cq_g2[1]

    _pipe_side = None     # the second stream of the two-stream form

    def _pipe_drop(self):
        """Forget a separator system made ahead (the stream it was made on is joined first: its buffers are about to be reused)."""
        if getattr(self, "_pre", None) is not None:
            if self._pipe_side is not None:
                torch.cuda.current_stream().wait_stream(self._pipe_side)
            self._pre = None

    def _pipe_predict(self, cq):
        """(site_lin, site_sym) as the next update_data_sites(self._pipe_lr) will leave them, into the model's two spare buffers (one
        launch; update_data_sites then takes them over by exchanging the buffers)."""
        g1, g2p = self._pipe_sites_gradient()
        if getattr(self, "_pipe_bufs", None) is None:
            self._pipe_bufs = [torch.empty_like(self.data_nat1), torch.empty_like(cq.site_sym)]
            if self.pipe_two_streams:
                self._pipe_side = torch.cuda.Stream(device=self.device)
        lin, sym = self._pipe_bufs
        self.plan.site_lerp_to(lin, self.data_nat1, g1.contiguous(), sym, cq.site_sym, g2p, self._pipe_lr)
        return lin, sym

    def update_data_sites(self, lr: float):
        cq = self._cq_state()
        if cq is None or cq.slot is None:
            return super().update_data_sites(lr)
        # the site gradient of such a likelihood does not depend on the marginals: only the small site arrays move, and the sweeps
        # read them where they need them (no scatter into per-node arrays)
        self._started = True
        pre = self._pre
        if (pre is not None and pre["lr"] == float(lr) and pre["cq"] is cq and pre["ver"] == cq.version
                and pre["v1"] == (id(self.data_nat1), self.data_nat1._version) and pre["v2"] == (id(cq.site_sym), cq.site_sym._version)):
            # exactly the blend the last ELBO refresh predicted (same inputs, same learning rate): its result is taken over -- the
            # site arrays exchange roles with the spare buffers it was written to -- and the separator system of the state it gives
            # is already being made
            self._pipe_bufs = [self.data_nat1, cq.site_sym]
            self.data_nat1, cq.site_sym = pre["lin"], pre["sym"]
            cq.site_lin = self.data_nat1
            pre["armed"] = (id(self.data_nat1), self.data_nat1._version, id(cq.site_sym), cq.site_sym._version)
        else:
            self._pipe_drop()
            g1, g2p = self._pipe_sites_gradient()
            # (1 - lr) site + lr gradient (variational_cvi_sde.py:301-317), both site arrays in one launch
            self.plan.site_lerp_to(self.data_nat1, self.data_nat1, g1.contiguous(), cq.site_sym, cq.site_sym, g2p, lr)
            torch.autograd.graph.increment_version(self.data_nat1)      # written behind torch's back
            torch.autograd.graph.increment_version(cq.site_sym)
        self._pipe_lr = float(lr)
        self._q, self._cq_dense, self._obs_fresh = None, None, False

    def snapshot(self):
        """Everything update_data_sites / update_girsanov_sites move, copied: the checkpoint a trainer that synchronises the host once
        per batch of iterations returns to when a learning-rate rule fires inside a batch (trainers.CVISitesTrainer)."""
        cq = self._cq_state()
        if cq is not None:
            return dict(kind="cq", cq=cq, dyn=cq.dyn.clone(), d_off=cq.d_off, s_off=cq.s_off,
                        p0=None if cq.p0_off is None else cq.p0_off.clone(), lin=self.data_nat1.clone(),
                        sym=None if cq.site_sym is None else cq.site_sym.clone(), started=self._started)
        tq = self.full_sites()
        return dict(kind="dense", tq=(tq.lin.clone(), tq.diag.clone(), tq.sub.clone()), lin=self.data_nat1.clone(), n2=self.data_nat2.clone(),
                    started=self._started)

    def restore(self, snap):
        self._pipe_drop()
        if snap["kind"] == "cq":
            cq = snap["cq"]
            if self._cq is not cq:
                raise RuntimeError("the model left the structured state since the snapshot was taken")
            cq.dyn.copy_(snap["dyn"])
            cq.d_off, cq.s_off = snap["d_off"], snap["s_off"]
            if snap["p0"] is not None:
                cq.p0_off.copy_(snap["p0"])
            self.data_nat1.copy_(snap["lin"])
            if snap["sym"] is not None:
                cq.site_sym.copy_(snap["sym"])
            cq.version += 1
        else:
            tq = self.full_sites()
            for dst, src in zip((tq.lin, tq.diag, tq.sub), snap["tq"]):
                dst.copy_(src)
            self.data_nat1.copy_(snap["lin"])
            self.data_nat2.copy_(snap["n2"])
        self._started = snap["started"]
        self._q, self._cq_dense, self._obs_fresh = None, None, False

    def _pipe_take(self, cq):
        """Whether a separator system was made ahead for the state as it is NOW: consumed by the next factorisation."""
        pre = getattr(self, "_pre", None)
        if pre is None:
            return False
        ok = (pre.get("armed") == (id(self.data_nat1), self.data_nat1._version, id(cq.site_sym), cq.site_sym._version)
              and pre["cq"] is cq and pre["ver"] == cq.version)
        if not ok or pre["epoch"] != self.plan.epoch:      # (another factorisation on the plan may have used the workspace since)
            self._pipe_drop()
            return False
        self._pre = None          # the consuming call joins the side stream itself
        return True

    def _refresh(self, want_sub=None, want_mom=None, want_marginals=False):
        """want_marginals: the full marginal arrays (mu, Sig) are wanted.  The ELBO / site-update loop needs only the KL sum and the
        marginals at the observation nodes, which the backward sweep hands over directly; the [B, T] marginal arrays (216 B per node
        of writes at d = 6) are produced on demand by one more backward pass over the factor that is still in place."""
        cq = self._cq_state()
        want_sub_ = self._need_sub if want_sub is None else want_sub
        if cq is None or want_sub_ or want_mom:
            return super()._refresh(want_sub=want_sub, want_mom=want_mom)      # dense route (on the materialised naturals in cq mode)
        pl = self.plan
        obs = cq.slot is not None
        lazy = obs and not want_marginals and os.environ.get("VIDP_LAZY_MARGINALS", "1") != "0"
        if self._q is None:
            nxt = None
            if (self.pipelined and obs and getattr(self, "_pipe_lr", None) is not None and self._sde_prm.kind == 0
                    and self.B * self.T >= self.pipeline_min_nodes and self.state_dim <= 6):      # (d = 7, 8: the kernel pair does not fit)
                # the level-0 reduce of the next step's first factorisation rides next to this factorisation's forward sweep
                self._pipe_drop()
                lin, sym = self._pipe_predict(cq)
                nxt = dict(lr=self._pipe_lr, cq=cq, ver=cq.version, v1=(id(self.data_nat1), self.data_nat1._version),
                           v2=(id(cq.site_sym), cq.site_sym._version), lin=lin, sym=sym)
            f = pl.cq_factor(cq, want_logdet=True, out=self._bufs["f"], use_ahead=self._pipe_take(cq) if nxt is None else False,
                             next_sites=(nxt["lin"], nxt["sym"]) if nxt else None, side=self._pipe_side)
            if nxt is not None:
                nxt["epoch"] = pl.epoch
            self._pre = nxt
            self._bufs["f"].update(L=f["L"], y=f["y"])
            s = pl.cq_selinv_kl(cq, f["L"], f["y"], self._sde_prm, out=self._bufs["s"], obs_mu=self.fx_mus_obs if obs else None,
                                obs_cov=self.fx_covs_obs if obs else None, want_marginals=not lazy)
            if not lazy:
                self._bufs["s"].update(Sig=s["Sig"], x=s["x"])
            self._q = dict(logdetL=f["logdet"], mu=s["x"], Sig=s["Sig"], Sub=None, mom=None, klpart=s["klpart"], epoch=pl.epoch)
            self._obs_fresh = obs
        elif self._q["mu"] is None and not lazy:
            # the marginal arrays after all: the factor of this posterior is in the buffers (and its coarse levels in the workspace,
            # unless another factorisation ran on the plan since)
            f = self._bufs["f"]
            if self._q.get("epoch") != pl.epoch:
                f = pl.cq_factor(cq, want_logdet=False, out=self._bufs["f"])
                self._q["epoch"] = pl.epoch
            s = pl.cq_selinv_kl(cq, f["L"], f["y"], self._sde_prm, out=self._bufs["s"], obs_mu=self.fx_mus_obs if obs else None,
                                obs_cov=self.fx_covs_obs if obs else None, want_marginals=True)
            self._bufs["s"].update(Sig=s["Sig"], x=s["x"])
            self._q.update(mu=s["x"], Sig=s["Sig"])
        return self._q

    def variational_expectation(self, partials=False):
        cq, lik = self._cq_state(), self.likelihood
        if cq is None or cq.slot is None:
            return super().variational_expectation()
        self._refresh()
        if not getattr(self, "_obs_fresh", False):
            self._gather_obs()
        return self.plan.mvn_ve_compact(self.fx_mus_obs, self.fx_covs_obs, self.n_obs, self._obs_flat(), lik.inv_covariance, lik.ve_constant,
                                        partials=partials)

    def _cq_elbo(self):
        """(per-trajectory ELBO [B], their sum) assembled by ONE launch from what the sweeps left (mfgm_cq_elbo): the variational
        expectations' block sums, the KL sum of the backward sweep and log|L_q|; None when the model is not on that route."""
        cq = self._cq_state()
        if cq is None or cq.slot is None:
            return None
        q = self._refresh()
        if q["klpart"] is None:
            return None
        ve = self.variational_expectation(partials=True)
        return self.plan.cq_elbo(ve, q["klpart"], q["logdetL"], -0.5 * self.T * self.state_dim)

    def classic_elbo_per_trajectory(self):
        e = self._cq_elbo()
        return e[0] if e is not None else super().classic_elbo_per_trajectory()

    def classic_elbo(self):
        e = self._cq_elbo()
        return e[1] if e is not None else super().classic_elbo()

    def KL_q_p(self):
        """
        KL between the posterior chain and the SDE prior, per trajectory [B] (variational_cvi_sde.py:446-486), as
        -H[q] - E_q[log p]: log|L_q| - T d / 2 + the moment-array sum of k_sde_lean.
        """
        q = self._refresh()
        if q["klpart"] is None and q["mom"] is None:
            q = self._refresh(want_mom=True)
        part = q["klpart"] if q["klpart"] is not None else self.plan.sde_lean(self._sde_prm, q["mom"], q["Sig"], mode=0)
        return part + q["logdetL"] - 0.5 * self.T * self.state_dim

    def KL_q_p_full(self):
        """The same KL through the full-block kernel (k_sde_kl mode 0): transition-wise Gaussian conditionals."""
        q = self._refresh(want_sub=True)
        return self.plan.sde_kl(self._sde_prm, q["mu"], q["Sig"], q["Sub"], mode=0)

    def grad_kl_wrt_exp_param(self):
        """(KL [B], (d/d eta_lin, d/d eta_diag, d/d eta_sub) packed) (variational_cvi_sde.py:488-493)."""
        q = self._refresh(want_sub=True)
        pl = self.plan
        grads = (pl.empty(VEC), pl.empty(SYM), pl.empty(FULL))
        kl = pl.sde_kl(self._sde_prm, q["mu"], q["Sig"], q["Sub"], mode=1, grads=grads)
        return kl, grads

    # -- prior-parameter learning (variational_cvi_sde.py:495-518) ------------------------------------------------------------
    def _refresh_sde_params(self):
        self._sde_prm = self.prior_sde.params(self.dt, self.prior_initial_state[0], self.prior_initial_state[1])
        if self._q is not None:
            self._q["klpart"] = None      # a KL sum taken inside the sweep was that of the previous prior parameters

    def set_prior_initial_state(self, mean, cov):
        """Replace p(x0) (the trainer re-sets it to the stationary OU covariance after every decay update)."""
        self.prior_initial_state = (mean, cov)
        self._refresh_sde_params()
        self._cq_p0_moved = True

    def grad_KL_wrt_cubic(self):
        """
        (d KL / d alpha, d KL / d beta) of the Euler map u(x) = alpha x - beta x^3, summed over trajectories, time and the
        state dimensions: KL = sum_t 1/2 w E_q[(x' - u(x))^2] + terms free of the drift, with Gaussian moments of
        (x, x') up to order six from the moment array (mu, diag Sigma, diag Sigma_{t+1,t}).
        """
        q = self._refresh(want_mom=True)
        d, T = self.state_dim, self.T
        mom = self.plan.unpack_moments(q["mom"])
        m, s, c = mom[:, :-1, :d], mom[:, :-1, d:2 * d], mom[:, :-1, 2 * d:]
        mn = mom[:, 1:, :d]
        al, be = self.prior_sde.cubic(self.dt)
        w = torch.tensor([1.0 / (self.dt * v) for v in self.prior_sde.q_diag], dtype=torch.float64, device=mom.device)
        m2, s2 = m * m, s * s
        Ex2 = m2 + s
        Ex3 = m * (m2 + 3.0 * s)
        Ex4 = m2 * m2 + 6.0 * m2 * s + 3.0 * s2
        Ex6 = m2 * m2 * m2 + 15.0 * m2 * m2 * s + 45.0 * m2 * s2 + 15.0 * s2 * s
        Exxn = m * mn + c                                # E[x x']
        Ex3xn = mn * Ex3 + 3.0 * c * Ex2                 # E[x^3 x'] (Stein)
        dal = -(w * (Exxn - al * Ex2 + be * Ex4)).sum()
        dbe = (w * (Ex3xn - al * Ex4 + be * Ex6)).sum()
        return float(dal), float(dbe)

    def grad_KL_wrt_prior_params(self):
        """d KL[q || p_SDE] / d (trainable drift parameters), in `prior_sde.trainable_variables` order (variational_cvi_sde.py:495-506)."""
        dal, dbe = self.grad_KL_wrt_cubic()
        jac = self.prior_sde.cubic_jacobian(self.dt)
        return [dal * jac[n][0] + dbe * jac[n][1] for n in self.prior_sde.trainable_variables]

    def _prior_naturals_on(self, path, frozen=None):
        """Naturals (lin [B,T,d], diag [B,T,d,d], sub [B,T-1,d,d], natural layout) of the prior linearised on the packed path (mu, Sigma)
        with the CURRENT drift parameters -- the (stabilised) prior `set_linearized_prior` installs; the model's state is not touched.
        frozen = (keep_A, A_clipped, keep_off, off_clipped) (packed): the clipping DECISION of another parameter value -- entries that
        were inside the clipping range there follow the unclipped linearisation here, the others stay at their bound -- which is how a
        tape differentiates tf.clip_by_value (zero through a clipped entry, one through a free one)."""
        pl = self.plan
        clip = self.clip_state_transitions if (self.stabilize_ssm and frozen is None) else None
        prm = self.prior_sde.params(self.dt, self.prior_initial_state[0], self.prior_initial_state[1], clip=clip)
        A, off, chol = pl.linearize_cubic(prm, path[0], path[1])
        if frozen is not None:
            A = torch.where(frozen[0], A, frozen[1])
            off = torch.where(frozen[2], off, frozen[3])
        nat = pl.ssm_to_naturals(A, off, chol)
        return pl.unpack(VEC, nat["lin"]), pl.unpack(SYM, nat["diag"]), pl.unpack(FULL, nat["sub"], self.T - 1)

    def _clip_decision(self, path):
        """(keep_A, A_clipped, keep_off, off_clipped) of the linearisation at the current drift parameters (None without clipping)."""
        if not self.stabilize_ssm:
            return None
        pl = self.plan
        lo, hi = self.clip_state_transitions
        prm = self.prior_sde.params(self.dt, self.prior_initial_state[0], self.prior_initial_state[1], clip=None)
        A, off, _ = pl.linearize_cubic(prm, path[0], path[1])
        return (A > lo) & (A < hi), A.clamp(lo, hi), (off > lo) & (off < hi), off.clamp(lo, hi)

    def grad_VE_wrt_prior_params(self, rel_step=1e-6, finite_difference=False):
        """
        d(-E_q log p(Y | X)) / d (trainable drift parameters) with q = linearised prior + sites (variational_cvi_sde.py:508-518: the
        reference re-linearises the prior on the current posterior inside a GradientTape and differentiates -VE of the resulting q).
        Exact chain rule, no re-factorisation per parameter (round 3):
            theta_q(kappa) = theta_p(kappa; path) + sites        (sites fixed, path = the current posterior marginals, fixed),
            d(-VE) / d kappa = -< d VE / d eta , F_q d theta_p / d kappa > = -< F_q g , d theta_p / d kappa >,
        with g = d VE / d eta the likelihood's site gradient at the observation nodes, F_q the Fisher matrix of the re-linearised q
        (symmetric: ONE Fisher-vector product, tape.fisher_vector_product, serves every parameter) and d theta_p / d kappa the derivative
        of the LOCAL linearisation map on the fixed path -- a polynomial of degree two in the Euler-map coefficients, for which the
        central difference used here is exact up to rounding; with `stabilize_ssm` the clipping decision of the current parameters is
        held fixed across the difference (zero slope through a clipped entry, as a tape through tf.clip_by_value).  `finite_difference=True` is the round-2 evaluation (two re-linearised
        posterior refreshes per parameter), kept as a cross-check.  Like the reference's, the call leaves the prior re-linearised at
        the current posterior.
        """
        from . import tape
        sde, pl = self.prior_sde, self.plan
        q = self._refresh(want_marginals=True)
        path = (q["mu"].clone(), q["Sig"].clone())

        def relinearize_on_path():
            self._path, self._q = path, None
            self._refresh_sde_params()
            self.set_linearized_prior()
            self._path, self._q = path, None

        if finite_difference:
            def neg_ve():
                relinearize_on_path()
                return -float(self.variational_expectation().sum())
            grads = []
            for n in sde.trainable_variables:
                v0 = sde.get(n)
                h = rel_step * max(abs(v0), 1.0)
                sde.assign(n, v0 + h)
                up = neg_ve()
                sde.assign(n, v0 - h)
                dn = neg_ve()
                sde.assign(n, v0)
                grads.append((up - dn) / (2.0 * h))
            neg_ve()
            return grads
        # the q the reference differentiates: prior re-linearised on the path at the current parameters, plus the sites
        relinearize_on_path()
        T, d, B = self.T, self.state_dim, self.B
        qn = self._refresh(want_sub=True, want_marginals=True)
        mu, cov, csub = pl.unpack(VEC, qn["mu"]), pl.unpack(SYM, qn["Sig"]), pl.unpack(FULL, qn["Sub"], T - 1)
        tq = self.full_sites()
        diag, sub = pl.unpack(SYM, tq.diag), pl.unpack(FULL, tq.sub, T - 1)
        mu_o, cov_o = self._obs_marginals()
        g1, g2 = self.likelihood.ve_gradients_expectation(mu_o, cov_o, self._obs_flat())
        g_lin = torch.zeros((B * T, d), dtype=torch.float64, device=self.device).index_add_(0, self.obs_node_ids, g1.reshape(-1, d))
        g_diag = torch.zeros((B * T, d, d), dtype=torch.float64, device=self.device).index_add_(0, self.obs_node_ids, g2.reshape(-1, d, d))
        u = tape.fisher_vector_product(pl, diag, sub, mu, cov, csub, g_lin.view(B, T, d), g_diag.view(B, T, d, d), torch.zeros_like(sub))
        grads = []
        # the clipping decision is taken ONCE, at the current parameters, and held while they are stepped: the unclipped map is a
        # polynomial of degree two in the Euler-map coefficients (the central difference is exact for it), whereas a difference taken
        # THROUGH the clipping blends the two slopes of every transition within h |dA / d kappa| of a bound
        frozen = self._clip_decision(path)
        for n in sde.trainable_variables:
            v0 = sde.get(n)
            h = 1e-3 * max(abs(v0), 1.0)
            sde.assign(n, v0 + h)
            up = self._prior_naturals_on(path, frozen)
            sde.assign(n, v0 - h)
            dn = self._prior_naturals_on(path, frozen)
            sde.assign(n, v0)
            grads.append(-float(sum(((a - b) * w).sum() for a, b, w in zip(up, dn, u))) / (2.0 * h))
        self._path, self._q = path, None
        return grads

    def update_girsanov_sites(self, lr: float):
        """Fused: g <- g + lr (scatter(data) - dKL/d eta), theta_q moves by the same increment (variational_cvi_sde.py:279-299)."""
        pl = self.plan
        self._sde_prm.lr = float(lr)
        cq = self._cq_state()
        if cq is not None:
            # reduce -> forward -> backward sweep that writes (1 - lr) dyn + lr theta~ into the spare buffer; the uniform off-diagonals
            # scale by (1 - lr); the data sites do not enter (they are not part of the resident state)
            f = pl.cq_factor(cq, want_logdet=False, out=self._bufs["f"], use_ahead=self._pipe_take(cq), side=self._pipe_side)
            self._bufs["f"].update(L=f["L"], y=f["y"])
            if cq.spare is None:
                cq.spare = torch.empty_like(cq.dyn)
            pl.cq_selinv_girsanov(cq, f["L"], f["y"], self._sde_prm, cq.spare)
            cq.dyn, cq.spare = cq.spare, cq.dyn
            cq.version += 1
            cq.d_off *= 1.0 - lr
            cq.s_off *= 1.0 - lr
            if self._cq_p0_moved and cq.p0_off is not None:
                # p(x0) was replaced after the state was built: the node-0 block follows -1/2 P0^{-1} like every other entry
                P0inv = self._sym_full(torch.tensor(list(self._sde_prm.P0inv)[:self.state_dim * (self.state_dim + 1) // 2],
                                                    dtype=torch.float64, device=self.device))
                tgt = -0.5 * (P0inv - torch.diag(torch.diagonal(P0inv)))
                cq.p0_off.lerp_(self._sym_packed(tgt), lr)
            self._q, self._cq_dense, self._obs_fresh, self._started = None, None, False, True
            return
        tq = self.full_sites()
        if (self._q is None or self._q["mom"] is None) and self._sweep_fusion():
            # no refresh of these sites is cached: the backward sweep of the refresh makes the update itself, without ever
            # writing the marginals (mfgm_girsanov.h); theta_q moves to the spare buffer
            f = pl.factor(tq.diag, tq.sub, tq.lin, aD=-2.0, aS=-1.0, aR=1.0, want_logdet=False, out=self._bufs["f"], store_G=False)
            self._bufs["f"].update(L=f["L"], y=f["y"])
            if getattr(self, "_theta_spare", None) is None:
                self._theta_spare = PackedBTDNat(pl.zeros(VEC), pl.zeros(SYM), pl.zeros(FULL))
            sp = self._theta_spare
            pl.selinv_girsanov(f["L"], tq.sub, -1.0, f["y"], self._sde_prm, (tq.lin, tq.diag, tq.sub), (sp.lin, sp.diag, sp.sub))
            self._theta_q, self._theta_spare = sp, tq
            tq = sp
        else:
            q = self._refresh(want_mom=True)
            pl.sde_lean(self._sde_prm, q["mom"], mode=3, theta_q=(tq.lin, tq.diag, tq.sub))
        pl.scatter_nodes_pair(tq.lin, tq.diag, self.obs_node_ids, self.data_nat1, self.data_nat2, scale=lr)
        self._q = None
        self._obs_fresh = False
        self._started = True


class CVISitesSDEQuadrature(CVISitesSSM):
    """
    CVI-DP (variational_cvi_sde.py:368-518) for drifts the closed-form kernels do not cover: drifts that couple the state dimensions
    (`sde.VanderPolOscillatorSDE`) or have no polynomial form (`sde.MLPDrift`).  Everything that is LOCAL in time follows the reference's
    own formulation -- the linearisation by 10-point-per-dimension Gauss-Hermite rules (sde_utils.py:119-179), KL[q || p_SDE] along the
    Gaussian path by the 20-point rule (sde_utils.py:262-359) and its gradient with respect to the expectation parameters through
    expectations_to_ssm_params (sde_utils.py:473-547; torch autograd stands in for the GradientTape) -- as batched torch operations on
    natural-layout tensors; everything SEQUENTIAL in time (the posterior refresh: factorisation, selected inverse, solves) runs in the
    HIP sweeps on the dense posterior naturals, as for every other model.  A small-model route: the tensor grids have 10^d / 20^d
    points per time step (the reference's own experiment is T = 501, d = 2).
    """

    def __init__(self, prior_sde, time_grid, input_data, likelihood, prior_initial_state=None, initial_posterior_path=None,
                 stabilize_ssm=True, clip_state_transitions=(-1.0, 1.0), plan=None):
        self.prior_sde = prior_sde
        d = input_data[1].shape[-1]
        if prior_initial_state is None:
            q = prior_sde.q.cpu().numpy()
            prior_initial_state = (torch.zeros(d, dtype=torch.float64).numpy(), q * (torch.ones((d, d), dtype=torch.float64).numpy()))
        self.stabilize_ssm, self.clip_state_transitions = stabilize_ssm, clip_state_transitions
        super().__init__(None, time_grid, input_data, likelihood, prior_initial_state=prior_initial_state,
                         initial_posterior_path=initial_posterior_path, plan=plan)
        # the local quantities on the HIP quadrature kernels (csrc/mfgm_quad.h; drifts with a `quad_kind`, d <= 3); VIDP_QUAD_TORCH=1
        # keeps the batched torch evaluation + autograd of round 3, which the tests hold the kernels to
        self.native = (os.environ.get("VIDP_QUAD_TORCH", "0") != "1" and getattr(prior_sde, "quad_kind", None) is not None
                       and self.state_dim <= 3)
        self.dist_p_linearized = None
        self.set_linearized_prior()

    def _quad_prm(self, clip=None):
        return self.prior_sde.quad_params(self.dt, self.prior_initial_state[0], self.prior_initial_state[1], clip=clip)

    def _refresh_sde_params(self):
        """The drift parameters moved (trainers call this after every optimiser step): nothing is cached on this route."""

    def set_prior_initial_state(self, mean, cov):
        self.prior_initial_state = (mean, cov)

    # -- linearisation ------------------------------------------------------------------------------------------------------------
    def _path_natural(self):
        """(mu [B, T, d], Sigma [B, T, d, d]) of the current posterior; before the first refresh: the initial path."""
        pl, d = self.plan, self.state_dim
        if self._q is not None:
            q = self._refresh(want_marginals=True)
            return pl.unpack(VEC, q["mu"]), pl.unpack(SYM, q["Sig"])
        if self._path is not None:
            return pl.unpack(VEC, self._path[0]), pl.unpack(SYM, self._path[1])
        mu = torch.zeros((self.B, self.T, d), dtype=torch.float64, device=self.device)
        return mu, torch.eye(d, dtype=torch.float64, device=self.device).expand(self.B, self.T, d, d).contiguous()

    def set_linearized_prior(self, move_theta_q=True):
        """A_k = I + dt E_q[df/dx], b_k = dt (E_q f - E_q[df/dx] m), Q_k = dt q on the current path (sde_utils.py:119-179;
        LinearDrift.to_ssm, drift.py:66-117), installed as dist_p; stabilised: transitions and offsets clipped
        (variational_cvi_sde.py:420-432)."""
        from . import linalg
        sde, d, dt = self.prior_sde, self.state_dim, self.dt
        mu, cov = self._path_natural()
        # transition k is linearised on the marginal at k + 1, as the reference hands `fx_mus[1:]` to linearize_sde (:412)
        m = mu[:, 1:]
        if self.native:
            from . import quad
            A, b = quad.linearize(self._quad_prm(), m, cov[:, 1:])
        else:
            chol = linalg.cholesky(cov[:, 1:].contiguous())
            from .sde import mvnquad
            with torch.no_grad():
                if hasattr(sde, "expected_drift"):
                    Ef = sde.expected_drift(m, chol)
                    J = sde.expected_gradient_drift(m, chol)
                else:       # per-dimension drifts (here for their full diffusion matrix): the same 10-point rule, diagonal Jacobian
                    Ef = mvnquad(lambda x: sde.drift(x), m, chol, 10, (d,))
                    J = torch.diag_embed(mvnquad(lambda x: sde.gradient_drift(x), m, chol, 10, (d,)))
            eye = torch.eye(d, dtype=torch.float64, device=self.device)
            A = eye + dt * J
            b = dt * (Ef - (J @ m[..., None])[..., 0])
        cholQ = linalg.cholesky((dt * sde.q).to(self.device)).expand(self.B, self.T - 1, d, d).contiguous()
        mu0 = torch.as_tensor(self.prior_initial_state[0], dtype=torch.float64, device=self.device).expand(self.B, d).contiguous()
        cholP0 = linalg.cholesky(torch.as_tensor(self.prior_initial_state[1], dtype=torch.float64, device=self.device))
        cholP0 = cholP0.expand(self.B, d, d).contiguous()
        self.dist_p_linearized = StateSpaceModel(mu0, cholP0, A.contiguous(), b.contiguous(), cholQ, plan=self.plan)
        if self.stabilize_ssm:
            lo, hi = self.clip_state_transitions
            self._set_prior(StateSpaceModel(mu0, cholP0, A.clamp(lo, hi).contiguous(), b.clamp(lo, hi).contiguous(), cholQ, plan=self.plan),
                            move_theta_q=move_theta_q)
        else:
            self._set_prior(self.dist_p_linearized, move_theta_q=move_theta_q)

    def relinearize(self):
        """cvi_dp_trainer.py:127-134 (see CVISitesSDE.relinearize): theta_q stays, the implicit Girsanov sites absorb the change."""
        q_valid = self._q
        self.full_sites()
        self.set_linearized_prior(move_theta_q=False)
        self._q = q_valid

    # -- KL[q || p_SDE] and its gradient --------------------------------------------------------------------------------------------
    def _kl_from_expectations(self, e1, ed, es):
        """The scalar SDE_SSM_KL_with_grads_wrt_exp_params differentiates (sde_utils.py:496-543), per chain [B], as a torch graph of the
        expectation parameters (e1 [B,T,d], ed [B,T,d,d], es [B,T-1,d,d])."""
        from . import linalg, tape
        from .sde import mvnquad
        sde, d, dt = self.prior_sde, self.state_dim, self.dt
        A, b, cP0, cQ, mu0 = tape.expectations_to_ssm_params(e1, ed, es)
        cov = ed - e1[..., :, None] * e1[..., None, :]
        Qq = cQ @ cQ.transpose(-1, -2)
        Qp = (dt * sde.q).to(self.device)
        Qp_inv = linalg.spd_inverse(Qp)
        ld = lambda c: 2.0 * torch.log(torch.diagonal(c, dim1=-2, dim2=-1)).sum(-1)
        const = -(ld(cQ) - linalg.logdet_spd(Qp)) - d + (Qp_inv * Qq).sum(dim=(-1, -2))                # [B, T-1]
        m, chol = e1[:, :-1], tape.cholesky(cov[:, :-1])

        def sq(x):                                                                                      # x [H^d, B, T-1, d]
            diff = x + dt * sde.drift(x) - ((A @ x[..., None])[..., 0] + b)
            return ((diff @ Qp_inv) * diff).sum(-1)
        path = 0.5 * (mvnquad(sq, m, chol, 20) + const).sum(-1)
        # KL[q(x0) || p(x0)]
        p_mu = torch.as_tensor(self.prior_initial_state[0], dtype=torch.float64, device=self.device)
        P0 = torch.as_tensor(self.prior_initial_state[1], dtype=torch.float64, device=self.device)
        P0inv = linalg.spd_inverse(P0)
        S0 = cP0 @ cP0.transpose(-1, -2)
        dm = p_mu - mu0
        kl0 = 0.5 * ((P0inv * S0).sum(dim=(-1, -2)) + ((dm @ P0inv) * dm).sum(-1) - d + linalg.logdet_spd(P0) - ld(cP0))
        return path + kl0

    def _expectations_natural(self):
        pl, T = self.plan, self.T
        q = self._refresh(want_sub=True, want_marginals=True)
        mu, cov, sub = pl.unpack(VEC, q["mu"]), pl.unpack(SYM, q["Sig"]), pl.unpack(FULL, q["Sub"], T - 1)
        return mu, cov + mu[..., :, None] * mu[..., None, :], sub + mu[:, 1:, :, None] * mu[:, :-1, None, :]

    def _marginals_natural(self):
        pl, T = self.plan, self.T
        q = self._refresh(want_sub=True, want_marginals=True)
        return pl.unpack(VEC, q["mu"]), pl.unpack(SYM, q["Sig"]), pl.unpack(FULL, q["Sub"], T - 1)

    def KL_q_p(self):
        if self.native:
            from . import quad
            return quad.kl(self._quad_prm(), *self._marginals_natural())
        with torch.no_grad():
            return self._kl_from_expectations(*self._expectations_natural())

    def grad_KL_wrt_prior_params(self):
        """d KL[q || p_SDE] / d (trainable drift parameters), summed over the trajectories, in `trainable_variables` order
        (variational_cvi_sde.py:495-506); a parameter that is a vector (the network's weights) gets a vector."""
        from . import quad
        sde = self.prior_sde
        _, gth = quad.kl(self._quad_prm(), *self._marginals_natural(), param_grad=True)
        g = gth.sum(0).cpu().numpy()
        out = []
        for n, jac in sde.quad_param_jacobian().items():
            out.append(g.copy() if jac is None else float(sum(gk * jk for gk, jk in zip(g, jac))))
        return out

    def grad_VE_wrt_prior_params(self, rel_step=1e-4):
        """d(-E_q log p(Y | X)) / d (trainable drift parameters) with q = prior re-linearised on the current path + sites
        (variational_cvi_sde.py:508-518), as CVISitesSDE.grad_VE_wrt_prior_params: ONE exact Fisher-vector product F_q g serves every
        parameter, contracted with d theta_p / d kappa of the local linearisation map on the fixed path (a fourth-order central
        difference of the quadrature kernel's linearisation; the map is smooth in the drift parameters)."""
        from . import quad, tape
        sde, pl = self.prior_sde, self.plan
        T, d, B = self.T, self.state_dim, self.B
        mu0, cov0 = self._path_natural()
        path = (pl.pack(VEC, mu0.reshape(B, T, d).contiguous()), pl.pack(SYM, cov0.reshape(B, T, d, d).contiguous()))
        self._path, self._q = path, None
        self.set_linearized_prior()
        self._path = path
        mu, cov, csub = self._marginals_natural()
        tq = self.full_sites()
        diag, sub = pl.unpack(SYM, tq.diag), pl.unpack(FULL, tq.sub, T - 1)
        mu_o, cov_o = self._obs_marginals()
        g1, g2 = self.likelihood.ve_gradients_expectation(mu_o, cov_o, self._obs_flat())
        g_lin = torch.zeros((B * T, d), dtype=torch.float64, device=self.device).index_add_(0, self.obs_node_ids, g1.reshape(-1, d))
        g_diag = torch.zeros((B * T, d, d), dtype=torch.float64, device=self.device).index_add_(0, self.obs_node_ids, g2.reshape(-1, d, d))
        u = tape.fisher_vector_product(pl, diag, sub, mu, cov, csub, g_lin.view(B, T, d), g_diag.view(B, T, d, d), torch.zeros_like(sub))

        def prior_nat():
            clip = self.clip_state_transitions if self.stabilize_ssm else None
            A, b = quad.linearize(self._quad_prm(clip=clip), mu0[:, 1:], cov0[:, 1:])
            ssm = StateSpaceModel(self.dist_p.initial_mean, self.dist_p.cholesky_initial_covariance, A, b,
                                  self.dist_p.cholesky_process_covariances, plan=pl)
            nat = pl.ssm_to_naturals(ssm.packed.A, ssm.packed.off, ssm.packed.chol)
            return pl.unpack(VEC, nat["lin"]), pl.unpack(SYM, nat["diag"]), pl.unpack(FULL, nat["sub"], T - 1)

        def directional(assign, h):
            vals = {}
            for k in (-2, -1, 1, 2):
                assign(k * h)
                vals[k] = prior_nat()
            assign(0.0)
            return [(8.0 * (a1 - b1) - (a2 - b2)) / (12.0 * h) for a1, b1, a2, b2 in zip(vals[1], vals[-1], vals[2], vals[-2])]

        grads = []
        for n in sde.trainable_variables:
            v0 = sde.get(n)
            if isinstance(v0, float):
                dth = directional(lambda e: sde.assign(n, v0 + e), rel_step * max(abs(v0), 1.0))
                grads.append(-float(sum((a * w).sum() for a, w in zip(dth, u))))
            else:
                comp = []
                for k in range(len(v0)):
                    def assign(e, k=k):
                        v = v0.copy()
                        v[k] += e
                        sde.assign(n, v)
                    dth = directional(assign, rel_step * max(abs(float(v0[k])), 1.0))
                    comp.append(-float(sum((a * w).sum() for a, w in zip(dth, u))))
                grads.append(np.array(comp))
        self._q = None
        return grads

    def grad_kl_wrt_exp_param(self):
        """(KL [B], (d/d eta_lin, d/d eta_diag, d/d eta_sub) packed) (variational_cvi_sde.py:488-493)."""
        pl = self.plan
        if self.native:
            from . import quad
            kl, (g1, gd, gs) = quad.kl(self._quad_prm(), *self._marginals_natural(), grad=True)
            return kl, (pl.pack(VEC, g1), pl.pack(SYM, gd), pl.pack(FULL, gs))
        eta = [e.detach().requires_grad_(True) for e in self._expectations_natural()]
        kl = self._kl_from_expectations(*eta)
        g1, gd, gs = torch.autograd.grad(kl.sum(), eta)
        gd = 0.5 * (gd + gd.transpose(-1, -2))
        return kl.detach(), (pl.pack(VEC, g1), pl.pack(SYM, gd), pl.pack(FULL, gs.contiguous()))

    def update_girsanov_sites(self, lr: float):
        """g <- g + lr (scatter(data sites) - dKL/d eta) (variational_cvi_sde.py:279-299); theta_q = theta_prior + g + data moves by
        the same increment."""
        pl = self.plan
        _, (g1, gd, gs) = self.grad_kl_wrt_exp_param()
        tq = self.full_sites()
        for qq, gg in ((tq.lin, g1), (tq.diag, gd), (tq.sub, gs)):
            pl.lincomb(qq, 1.0, qq, -float(lr), gg)
        pl.scatter_nodes_pair(tq.lin, tq.diag, self.obs_node_ids, self.data_nat1, self.data_nat2, scale=lr)
        self._q = None
        self._obs_fresh = False
        self._started = True


def tranform_girsanov_sites(plan, girsanov_sites, current_prior_nat, new_prior_nat):
    """
    g <- g + theta(current prior) - theta(new prior), in place (sde_utils.py:550-568; the reference's spelling).
    All arguments are packed natural parameters (PackedBTDNat).
    """
    for g, a, b in ((girsanov_sites.lin, current_prior_nat.lin, new_prior_nat.lin),
                    (girsanov_sites.diag, current_prior_nat.diag, new_prior_nat.diag),
                    (girsanov_sites.sub, current_prior_nat.sub, new_prior_nat.sub)):
        plan.lincomb(g, 1.0, g, 1.0, a, -1.0, b)
    return girsanov_sites


def _ssm_from_packed(plan, A, off, chol):
    """StateSpaceModel whose packed parameter arrays are given (natural tensors are produced lazily on demand)."""
    T, d = plan.T, plan.d
    ssm = StateSpaceModel.__new__(StateSpaceModel)
    ssm.batch_shape = (plan.B,)
    ssm.B, ssm.T, ssm.d = plan.B, T, d
    ssm.plan = plan
    from .state_space_model import PackedSSM
    ssm._packed = PackedSSM(plan, A, off, chol)
    ssm._prec = None
    ssm._post = None
    ssm._lazy_natural = True
    return ssm
