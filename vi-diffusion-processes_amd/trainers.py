"""
Host-side mirror of the experiment trainers that drive the hot path: `CVISitesTrainer`
(docs/diffusion_processes/cvi_dp_trainer.py:19-200) and `VIMarkovGPTrainer`
(docs/diffusion_processes/vi_markov_gp_trainer.py:17-135): the inference loops with their learning-rate decay and
convergence rules, NLPD / RMSE on held-out grid points (exp_dp_utils.py:189-224), and the prior-parameter learning loops
(cvi_dp_trainer.py:138-250, vi_markov_gp_trainer.py:163-215: Adam on the drift parameters).  The wandb / hydra
plumbing is out of scope.  The site-optimisation loop of CVISitesTrainer synchronises the host once per `sync_every` iterations
(`_optimize_sites_batched`: the iterations of a batch are issued back to back with their ELBO / metric sums kept on the device; the
learning-rate decay and convergence rules are then applied to the batch's values exactly as the reference applies them one by one,
and when a rule fires inside a batch the model is put back to the batch's checkpoint and the iterations up to that point are replayed --
the kernels are deterministic, so the sequence of states and numbers is the reference's); sync_every=1 is the plain loop.

Trajectories spread over processes (SURVEY 8e first row; each process builds its model on its own shard of the batch): every scalar
that steers a loop -- the ELBO, the NLPD / RMSE sums -- and every hyper-parameter gradient is summed over the ranks
(`distributed.sum_over_ranks`, one all-reduce of a few doubles per use) BEFORE it is used, so that all ranks take the same Adam step
and the same learning-rate / convergence decisions as a single process holding all trajectories.
"""
import logging
import math

import torch

from . import linalg
from ._lib import SYM, VEC
from .distributed import sum_over_ranks
from .variational_cvi_sde import grid_indices

logger = logging.getLogger(__name__)


def _sum_grads_over_ranks(grads):
    """sum_over_ranks for a list whose entries are floats or NumPy vectors (one all-reduce of the flattened list)."""
    import numpy as np
    sizes = [int(np.size(g)) for g in grads]
    flat = sum_over_ranks(np.concatenate([np.atleast_1d(np.asarray(g, dtype=np.float64)).reshape(-1) for g in grads]).tolist())
    out, at = [], 0
    for g, n in zip(grads, sizes):
        out.append(float(flat[at]) if np.ndim(g) == 0 else np.array(flat[at:at + n]))
        at += n
    return out


class _Metrics:
    """NLPD / RMSE of the (batched) posterior at held-out grid points."""

    def __init__(self, model, test_data, time_grid):
        self.model = model
        if test_data is None:
            self.idx = None
            return
        t_test, y_test = test_data
        if y_test.dim() == 2:
            y_test = y_test[None]
        self.y = y_test
        self.idx = grid_indices(time_grid, t_test).to(y_test.device)
        self.node_ids = model.plan.node_ids(self.idx)

    def sums(self, mu_packed, Sig_packed):
        """(sum of log predictive densities, sum of squared errors, number of points, number of entries) over the local trajectories."""
        if self.idx is None:
            return 0.0, 0.0, 0.0, 0.0
        pl, lik = self.model.plan, self.model.likelihood
        B, n, d = self.y.shape
        m = pl.gather_nodes(VEC, mu_packed, self.node_ids)
        S = pl.gather_nodes(SYM, Sig_packed, self.node_ids)
        # likelihood.predict_mean_and_var: y* ~ N(m, S + R)
        R = lik.chol_covariance @ lik.chol_covariance.transpose(-1, -2)
        chol = linalg.cholesky(S + R)
        z = linalg.solve_lower(chol, self.y.reshape(B * n, d) - m)
        logp = -0.5 * (z * z).sum(-1) - torch.log(torch.diagonal(chol, dim1=-2, dim2=-1)).sum(-1) - 0.5 * d * math.log(2 * math.pi)
        return float(logp.sum()), float(((m - self.y.reshape(B * n, d)) ** 2).sum()), float(B * n), float(B * n * d)

    def sums_device(self, mu_packed, Sig_packed):
        """The same four sums as a device tensor [4] (no host synchronisation)."""
        pl, lik = self.model.plan, self.model.likelihood
        B, n, d = self.y.shape
        m = pl.gather_nodes(VEC, mu_packed, self.node_ids)
        S = pl.gather_nodes(SYM, Sig_packed, self.node_ids)
        R = lik.chol_covariance @ lik.chol_covariance.transpose(-1, -2)
        chol = linalg.cholesky(S + R, check=False)          # (no synchronisation: a failed pivot shows as NaN in the sums)
        z = linalg.solve_lower(chol, self.y.reshape(B * n, d) - m)
        logp = -0.5 * (z * z).sum(-1) - torch.log(torch.diagonal(chol, dim1=-2, dim2=-1)).sum(-1) - 0.5 * d * math.log(2 * math.pi)
        cnt = torch.tensor([float(B * n), float(B * n * d)], dtype=torch.float64, device=m.device)
        return torch.cat([logp.sum().reshape(1), ((m - self.y.reshape(B * n, d)) ** 2).sum().reshape(1), cnt])

    @staticmethod
    def finish(logp_sum, se_sum, n_points, n_entries):
        if n_points == 0:
            return float("nan"), float("nan")
        return -logp_sum / n_points, math.sqrt(se_sum / n_entries)

    def __call__(self, mu_packed, Sig_packed):
        return self.finish(*sum_over_ranks(self.sums(mu_packed, Sig_packed)))


class _Adam:
    """tf.optimizers.Adam defaults (beta_1 0.9, beta_2 0.999, epsilon 1e-7) on a short list of scalars."""

    def __init__(self, lr, n):
        self.lr, self.m, self.v, self.t = float(lr), [0.0] * n, [0.0] * n, 0

    def step(self, values, grads):
        """values / grads: floats, or NumPy vectors for a vector-valued parameter (the network drift's weights)."""
        import numpy as np
        self.t += 1
        out = []
        for i, (x, g) in enumerate(zip(values, grads)):
            self.m[i] = 0.9 * self.m[i] + 0.1 * g
            self.v[i] = 0.999 * self.v[i] + 0.001 * g * g
            a = self.lr * math.sqrt(1.0 - 0.999 ** self.t) / (1.0 - 0.9 ** self.t)
            out.append(x - a * self.m[i] / (np.sqrt(self.v[i]) + 1e-7))
        return out


class CVISitesTrainer:
    """cvi_dp_trainer.py:19-250."""

    def __init__(self, model, test_data=None, prior_sde=None, max_itr=100, optim_tol=1e-2, max_itr_sites_optim=20,
                 girsanov_sites_lr=0.1, data_sites_lr=0.1, learn_prior_sde=False, prior_sde_lr=1e-2, learning_max_itr=100,
                 learning_tol=1e-2, sync_every=1):
        self.model, self.prior_sde = model, prior_sde
        # iterations of the site loop between host synchronisations (1: the reference's loop as written; > 1 needs a model with
        # snapshot() / restore(), i.e. CVISitesSDE)
        self.sync_every = int(sync_every) if hasattr(model, "snapshot") else 1
        self.max_itr, self.optim_tol, self.max_itr_sites_optim = max_itr, optim_tol, max_itr_sites_optim
        self.girsanov_sites_lr, self.data_sites_lr = girsanov_sites_lr, data_sites_lr
        self._metrics = _Metrics(model, test_data, model.time_grid)
        self.learn_prior_sde, self.prior_sde_lr = bool(learn_prior_sde), float(prior_sde_lr)
        self.learning_max_itr, self.learning_tol = int(learning_max_itr), float(learning_tol)
        self.prior_params = {}
        if self.learn_prior_sde:
            names = model.prior_sde.trainable_variables
            if not names:
                raise ValueError("learn_prior_sde needs a prior SDE with trainable parameters")
            self.prior_sde_optim = _Adam(self.prior_sde_lr, len(names))
            self.store_prior_param_vals()

    def store_prior_param_vals(self):
        """cvi_dp_trainer.py:52-61: history of the trainable drift parameters, keyed by their index."""
        sde = self.model.prior_sde
        for i, n in enumerate(sde.trainable_variables):
            self.prior_params.setdefault(i, []).append(sde.get(n))

    def optimize_prior_sde(self):
        """cvi_dp_trainer.py:207-250: Adam steps on the drift parameters with d(KL - VE)/d params until the ELBO settles."""
        from .sde import OrnsteinUhlenbeckSDE
        model, sde = self.model, self.model.prior_sde
        elbo_vals, nlpd_vals, rmse_vals = [self._elbo()], [], []
        for _ in range(self.learning_max_itr):
            grads_kl = model.grad_KL_wrt_prior_params()
            grads_ve = model.grad_VE_wrt_prior_params()
            names = sde.trainable_variables
            # the gradient of the whole batch: summed over the ranks before the optimiser sees it
            grads = _sum_grads_over_ranks([a + b for a, b in zip(grads_kl, grads_ve)])
            new = self.prior_sde_optim.step([sde.get(n) for n in names], grads)
            for n, v in zip(names, new):
                sde.assign(n, v)
            # the reference's sequence (cvi_dp_trainer.py:221-235): the iteration's ELBO / NLPD / RMSE are taken with the new drift
            # parameters but the OLD p(x0); only then is p(x0) reset to the stationary OU covariance
            model._refresh_sde_params()
            el, nl, rm = self._elbo_nlpd_rmse()
            elbo_vals.append(el)
            nlpd_vals.append(nl)
            rmse_vals.append(rm)
            self.store_prior_param_vals()
            if isinstance(sde, OrnsteinUhlenbeckSDE):
                # stationary initial state q / (2 decay) (cvi_dp_trainer.py:231-235)
                d = model.state_dim
                cov = torch.diag(torch.tensor(sde.q_diag, dtype=torch.float64)) / (2.0 * sde.decay)
                model.set_prior_initial_state(torch.zeros(d, dtype=torch.float64).numpy(), cov.numpy())
            if elbo_vals[-1] < elbo_vals[-2]:
                logger.info("Decaying the LR!!!")
                self.prior_sde_optim.lr /= 2
            if abs(elbo_vals[-2] - elbo_vals[-1]) < self.learning_tol:
                logger.info("Prior parameter optimized successfully!!!")
                break
        return elbo_vals[1:], nlpd_vals, rmse_vals

    def _elbo(self):
        """The ELBO of all trajectories of all ranks."""
        return sum_over_ranks([float(self.model.classic_elbo())])[0]

    def _nlpd_rmse(self):
        q = self.model._refresh(want_marginals=True)
        return self._metrics(q["mu"], q["Sig"])

    def _elbo_nlpd_rmse(self):
        """ELBO, NLPD and RMSE of the whole batch from ONE all-reduce."""
        e = float(self.model.classic_elbo())
        if self._metrics.idx is None:
            return sum_over_ranks([e])[0], float("nan"), float("nan")
        q = self.model._refresh(want_marginals=True)
        tot = sum_over_ranks([e, *self._metrics.sums(q["mu"], q["Sig"])])
        return (tot[0], *self._metrics.finish(*tot[1:]))

    def _device_sums(self):
        """[ELBO, sum log p, sum of squared errors, points, entries] of the local trajectories as ONE device tensor (no host
        synchronisation): what _elbo_nlpd_rmse reduces over the ranks and finishes on the host."""
        e = self.model.classic_elbo().reshape(1)
        if self._metrics.idx is None:
            return e
        q = self.model._refresh(want_marginals=True)
        return torch.cat([e, self._metrics.sums_device(q["mu"], q["Sig"])])

    def _optimize_sites_batched(self):
        """cvi_dp_trainer.py:63-95 with one host synchronisation per `sync_every` iterations (module docstring)."""
        from .distributed import sum_tensor_over_ranks
        model, k = self.model, self.sync_every
        elbos = [self._elbo()]
        nlpds, rmses = [], []
        done = False
        while (len(elbos) - 1) < self.max_itr_sites_optim and not done:
            n_it = min(k, self.max_itr_sites_optim - (len(elbos) - 1))
            snap = model.snapshot() if n_it > 1 else None

            def run(n):
                out = []
                for _ in range(n):
                    model.update_data_sites(self.data_sites_lr)
                    model.update_girsanov_sites(self.girsanov_sites_lr)
                    out.append(self._device_sums())
                return sum_tensor_over_ranks(torch.stack(out)).cpu()      # the batch's one synchronisation (and one all-reduce)

            vals = run(n_it)
            # the reference's rules, iteration by iteration, on the batch's values
            fired, decay = n_it, False
            for i in range(n_it):
                prev = elbos[-1] if i == 0 else float(vals[i - 1, 0])
                cur = float(vals[i, 0])
                n_seen = len(elbos) + i + 1
                dec = n_seen > 2 and prev > cur
                conv = n_seen > 2 and abs(prev - cur) < self.optim_tol
                if dec or conv:
                    fired, decay, done = i + 1, dec, conv
                    break
            if fired < n_it:
                # a rule fired inside the batch: the iterations after it ran with learning rates (or at all) the reference would not
                # have used -- back to the checkpoint, the iterations up to the rule again (deterministic kernels: the same numbers)
                model.restore(snap)
                vals = run(fired)
            for i in range(fired):
                elbos.append(float(vals[i, 0]))
                nl, rm = (self._metrics.finish(*[float(v) for v in vals[i, 1:]]) if self._metrics.idx is not None
                          else (float("nan"), float("nan")))
                nlpds.append(nl)
                rmses.append(rm)
            if not math.isfinite(elbos[-1]):
                model.plan.check_info()      # a NaN ELBO is how a failed pivot surfaces without a per-iteration synchronisation
            if decay:
                logger.info("Decaying LR! ELBO decreasing!!!")
                self.girsanov_sites_lr /= 10
                self.data_sites_lr /= 10
            if done:
                logger.info("Breaking the site updates loop. ELBO converged!")
        return elbos[1:], nlpds, rmses

    def _optimize_sites_under_stable_prior(self):
        """cvi_dp_trainer.py:63-95."""
        if self.sync_every > 1:
            return self._optimize_sites_batched()
        elbos = [self._elbo()]
        nlpds, rmses = [], []
        while (len(elbos) - 1) < self.max_itr_sites_optim:
            self.model.update_data_sites(self.data_sites_lr)
            self.model.update_girsanov_sites(self.girsanov_sites_lr)
            el, nl, rm = self._elbo_nlpd_rmse()
            if not math.isfinite(el):
                # the sweeps do not synchronise the host: a pivot block that is not positive definite surfaces as a NaN bound, and a NaN
                # compares as "not worse" in the rules below -- raise with the failing location instead of iterating on it
                self.model.plan.check_info()
            elbos.append(el)
            nlpds.append(nl)
            rmses.append(rm)
            if len(elbos) > 2 and elbos[-2] > elbos[-1]:
                logger.info("Decaying LR! ELBO decreasing!!!")
                self.girsanov_sites_lr /= 10
                self.data_sites_lr /= 10
            if len(elbos) > 2 and abs(elbos[-2] - elbos[-1]) < self.optim_tol:
                logger.info("Breaking the site updates loop. ELBO converged!")
                break
        return elbos[1:], nlpds, rmses

    def perform_inference(self):
        """cvi_dp_trainer.py:97-136: site optimisation under the stabilised prior, then re-linearisation with site transformation."""
        elbo_vals, nlpd_vals, rmse_vals = [self._elbo()], [], []
        relin = hasattr(self.model, "relinearize")
        for i in range(self.max_itr):
            before = self._elbo()
            e, n, r = self._optimize_sites_under_stable_prior()
            # (the reference swaps to the unclipped linearised prior here; the posterior, hence the ELBO under the SDE
            #  prior, is unchanged by the site transformation)
            after = self._elbo()
            elbo_vals += e
            nlpd_vals += n
            rmse_vals += r
            if abs(before - after) < self.optim_tol:
                logger.info("ELBO converged! Optimization successfully completed!")
                break
            if relin and i != self.max_itr - 1:
                self.model.relinearize()
        return elbo_vals[1:], nlpd_vals, rmse_vals

    def optimize(self):
        """cvi_dp_trainer.py:138-187: inference, then (optionally) alternate with prior-parameter learning."""
        e0, n0, r0 = self._elbo_nlpd_rmse()
        elbo_vals, nlpd_vals, rmse_vals = [e0], [n0], [r0]
        previous = []
        for _ in range(self.max_itr):
            e, n, r = self.perform_inference()
            elbo_vals, nlpd_vals, rmse_vals = elbo_vals + e, nlpd_vals + n, rmse_vals + r
            if not self.learn_prior_sde:
                break
            pe, pn, pr = self.optimize_prior_sde()
            done = abs(elbo_vals[-1] - pe[-1]) < self.optim_tol
            elbo_vals, nlpd_vals, rmse_vals = elbo_vals + pe, nlpd_vals + pn, rmse_vals + pr
            if done:
                logger.info("Model successfully optimized!!!")
                break
            if len(previous) > 4 and (abs(previous[-1] - previous[-3]) < 1e-4 or abs(previous[-2] - previous[-4]) < 1e-4):
                logger.info("The objective is most probably jumping between two values!!!")
                break
            previous.append(elbo_vals[-1])
        return elbo_vals, nlpd_vals, rmse_vals, self.prior_params


class VIMarkovGPTrainer:
    """vi_markov_gp_trainer.py:17-215."""

    def __init__(self, model, test_data=None, q_lr=0.1, x0_lr=0.1, max_itr=1000, lr_tol=1e-2, optim_tol=1e-4, warmup_x0_itr=10,
                 warmup_itr=20, learn_prior_sde=False, prior_sde_lr=1e-2, learning_max_itr=100, learning_tol=1e-2,
                 optimize_prior_initial_state=False, prior_initial_state_lr=0.1):
        self.model = model
        self.q_lr, self.x0_lr, self.max_itr = q_lr, x0_lr, max_itr
        self.lr_tol, self.optim_tol, self.warmup_x0_itr, self.warmup_itr = lr_tol, optim_tol, warmup_x0_itr, warmup_itr
        self._metrics = _Metrics(model, test_data, model.grid)
        self.learn_prior_sde, self.prior_sde_lr = bool(learn_prior_sde), float(prior_sde_lr)
        self.learning_max_itr, self.learning_tol = int(learning_max_itr), float(learning_tol)
        self.optimize_prior_initial_state = bool(optimize_prior_initial_state)
        self.prior_initial_state_lr = float(prior_initial_state_lr)
        self.prior_params = {}
        if self.learn_prior_sde:
            names = model.prior_sde.trainable_variables
            if not names:
                raise ValueError("learn_prior_sde needs a prior SDE with trainable parameters")
            self.prior_sde_optim = _Adam(self.prior_sde_lr, len(names))
            self.store_prior_param_vals()

    def store_prior_param_vals(self):
        sde = self.model.prior_sde
        for i, n in enumerate(sde.trainable_variables):
            self.prior_params.setdefault(i, []).append(sde.get(n))

    def perform_inference(self):
        """vi_markov_gp_trainer.py:50-92."""
        mdl = self.model
        mS = mdl._forward_packed()
        elbos, nlpds, rmses = [self._elbo_nlpd_rmse(mS)[0]], [], []
        q_lr, x0_lr = self.q_lr, self.x0_lr
        for i in range(self.max_itr):
            # the marginals that close an iteration (for its ELBO) are those the next one starts from: the parameters do not change
            # in between, so the reference's second forward_pass per iteration (vi_markov_gp_trainer.py:60) is not repeated
            mdl.update_lagrange_and_param(mS, lr=q_lr)
            if i > self.warmup_x0_itr:
                mdl.update_initial_statistics(lr=x0_lr)
            mS = mdl._forward_packed()
            el, nl, rm = self._elbo_nlpd_rmse(mS)
            elbos.append(el)
            nlpds.append(nl)
            rmses.append(rm)
            if elbos[-2] > elbos[-1] or abs(elbos[-2] - elbos[-1]) < self.lr_tol:
                q_lr /= 10
                x0_lr /= 10
            if abs(elbos[-2] - elbos[-1]) < self.optim_tol:
                break
        return elbos[1:], nlpds, rmses

    def _elbo_nlpd_rmse(self, mS=None):
        """ELBO, NLPD and RMSE of all trajectories of all ranks from ONE all-reduce."""
        mdl = self.model
        mS = mS if mS is not None else mdl._forward_packed()
        e = float(mdl.elbo(mS))
        if self._metrics.idx is None:
            return sum_over_ranks([e])[0], float("nan"), float("nan")
        tot = sum_over_ranks([e, *self._metrics.sums(*mS)])
        return (tot[0], *self._metrics.finish(*tot[1:]))

    def optimize_prior_x0(self):
        """
        vi_markov_gp_trainer.py:203-215, as written there: the new "Cholesky factor" M is q / (2 decay) for an OU prior and
        scale - lr * d KL / d scale otherwise, and the covariance handed to the model is M @ M (no transpose, and for the OU
        prior the square of the stationary variance) -- mirrored, not corrected.
        """
        from . import linalg
        from .sde import OrnsteinUhlenbeckSDE
        mdl, sde = self.model, self.model.prior_sde
        if isinstance(sde, OrnsteinUhlenbeckSDE):
            mean = mdl.p0_mu
            M = (torch.diag(torch.tensor(sde.q_diag, dtype=torch.float64)) / (2.0 * sde.decay)).numpy()
        else:
            g_loc, g_scale = mdl.grad_initial_state()
            # sums over the trajectories: of every rank
            g = sum_over_ranks(torch.cat([g_loc.reshape(-1), g_scale.reshape(-1)]).tolist())
            g_loc = torch.tensor(g[:g_loc.numel()], dtype=torch.float64, device=g_loc.device).view_as(g_loc)
            g_scale = torch.tensor(g[g_loc.numel():], dtype=torch.float64, device=g_scale.device).view_as(g_scale)
            P0 = torch.from_numpy(mdl.p0_cov).to(mdl.device)
            mean = mdl.p0_mu - self.prior_initial_state_lr * g_loc.cpu().numpy()
            M = (linalg.cholesky(P0) - self.prior_initial_state_lr * g_scale).cpu().numpy()
        mdl.set_prior_initial_state(mean, M @ M)

    def optimize_prior_sde(self):
        """vi_markov_gp_trainer.py:163-201: Adam on the drift parameters with dE_sde/d params at the current (m, S)."""
        mdl, sde = self.model, self.model.prior_sde
        elbo_vals, nlpd_vals, rmse_vals = [self._elbo_nlpd_rmse()[0]], [], []
        for _ in range(self.learning_max_itr):
            grads = _sum_grads_over_ranks(mdl.grad_prior_sde_params())      # of the whole batch, before the optimiser sees it
            names = sde.trainable_variables
            for n, v in zip(names, self.prior_sde_optim.step([sde.get(n) for n in names], grads)):
                sde.assign(n, v)
            mdl._refresh_drift_params()
            if self.optimize_prior_initial_state:
                self.optimize_prior_x0()
            mS = mdl._forward_packed()
            el, nl, rm = self._elbo_nlpd_rmse(mS)
            elbo_vals.append(el)
            nlpd_vals.append(nl)
            rmse_vals.append(rm)
            self.store_prior_param_vals()
            if abs(elbo_vals[-2] - elbo_vals[-1]) < self.learning_tol:
                logger.info("Prior parameter optimized successfully!!!")
                break
        return elbo_vals[1:], nlpd_vals, rmse_vals

    def optimize(self):
        """vi_markov_gp_trainer.py:94-135: warm-up iterations at lr = 1e-6, then inference alternating with prior learning."""
        mdl = self.model
        for _ in range(self.warmup_itr):
            mS = mdl._forward_packed()
            mdl.update_lagrange(mS)
            mdl.update_param(mS, lr=1e-6)
        mS = mdl._forward_packed()
        e0, n0, r0 = self._elbo_nlpd_rmse(mS)
        elbo_vals, nlpd_vals, rmse_vals = [e0], [n0], [r0]
        for _ in range(self.max_itr):
            e, n, r = self.perform_inference()
            elbo_vals, nlpd_vals, rmse_vals = elbo_vals + e, nlpd_vals + n, rmse_vals + r
            if not self.learn_prior_sde:
                break
            pe, pn, pr = self.optimize_prior_sde()
            elbo_vals, nlpd_vals, rmse_vals = elbo_vals + pe, nlpd_vals + pn, rmse_vals + pr
            if len(pe) > 2 and abs(pe[-2] - pe[-1]) < self.optim_tol:
                logger.info("Model successfully optimized!!!")
                break
        return elbo_vals, nlpd_vals, rmse_vals, self.prior_params
