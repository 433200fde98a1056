"""
Host-side mirror of the experiment trainers that drive the hot path: `CVISitesTrainer`
(docs/diffusion_processes/cvi_dp_trainer.py:19-200) and `VIMarkovGPTrainer`
(docs/diffusion_processes/vi_markov_gp_trainer.py:17-135): the inference loops with their learning-rate decay and
convergence rules, NLPD / RMSE on held-out grid points (exp_dp_utils.py:189-224), and the prior-parameter learning loops
(cvi_dp_trainer.py:138-250, vi_markov_gp_trainer.py:163-215: Adam on the drift parameters).  The wandb / hydra
plumbing is out of scope.  One host synchronisation per iteration remains (the ELBO scalar decides
the learning-rate decay), everything else stays on the device.
"""
import logging
import math

import torch

from . import linalg
from ._lib import SYM, VEC
from .variational_cvi_sde import grid_indices

logger = logging.getLogger(__name__)


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

    def __call__(self, mu_packed, Sig_packed):
        if self.idx is None:
            return float("nan"), float("nan")
        pl, lik = self.model.plan, self.model.likelihood
        B, n, d = self.y.shape
        m = pl.gather_nodes(VEC, mu_packed, self.node_ids)
        S = pl.gather_nodes(SYM, Sig_packed, self.node_ids)
        # likelihood.predict_mean_and_var: y* ~ N(m, S + R)
        R = lik.chol_covariance @ lik.chol_covariance.transpose(-1, -2)
        chol = linalg.cholesky(S + R)
        z = linalg.solve_lower(chol, self.y.reshape(B * n, d) - m)
        logp = -0.5 * (z * z).sum(-1) - torch.log(torch.diagonal(chol, dim1=-2, dim2=-1)).sum(-1) - 0.5 * d * math.log(2 * math.pi)
        nlpd = float(-logp.mean())
        rmse = float(torch.sqrt(((m - self.y.reshape(B * n, d)) ** 2).mean()))
        return nlpd, rmse


class _Adam:
    """tf.optimizers.Adam defaults (beta_1 0.9, beta_2 0.999, epsilon 1e-7) on a short list of scalars."""

    def __init__(self, lr, n):
        self.lr, self.m, self.v, self.t = float(lr), [0.0] * n, [0.0] * n, 0

    def step(self, values, grads):
        self.t += 1
        out = []
        for i, (x, g) in enumerate(zip(values, grads)):
            self.m[i] = 0.9 * self.m[i] + 0.1 * g
            self.v[i] = 0.999 * self.v[i] + 0.001 * g * g
            a = self.lr * math.sqrt(1.0 - 0.999 ** self.t) / (1.0 - 0.9 ** self.t)
            out.append(x - a * self.m[i] / (math.sqrt(self.v[i]) + 1e-7))
        return out


class CVISitesTrainer:
    """cvi_dp_trainer.py:19-250."""

    def __init__(self, model, test_data=None, prior_sde=None, max_itr=100, optim_tol=1e-2, max_itr_sites_optim=20,
                 girsanov_sites_lr=0.1, data_sites_lr=0.1, learn_prior_sde=False, prior_sde_lr=1e-2, learning_max_itr=100,
                 learning_tol=1e-2):
        self.model, self.prior_sde = model, prior_sde
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
        elbo_vals, nlpd_vals, rmse_vals = [float(model.classic_elbo())], [], []
        for _ in range(self.learning_max_itr):
            grads_kl = model.grad_KL_wrt_prior_params()
            grads_ve = model.grad_VE_wrt_prior_params()
            names = sde.trainable_variables
            new = self.prior_sde_optim.step([sde.get(n) for n in names], [a + b for a, b in zip(grads_kl, grads_ve)])
            for n, v in zip(names, new):
                sde.assign(n, v)
            # the reference's sequence (cvi_dp_trainer.py:221-235): the iteration's ELBO / NLPD / RMSE are taken with the new drift
            # parameters but the OLD p(x0); only then is p(x0) reset to the stationary OU covariance
            model._refresh_sde_params()
            elbo_vals.append(float(model.classic_elbo()))
            nl, rm = self._nlpd_rmse()
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

    def _nlpd_rmse(self):
        q = self.model._refresh(want_marginals=True)
        return self._metrics(q["mu"], q["Sig"])

    def _optimize_sites_under_stable_prior(self):
        """cvi_dp_trainer.py:63-95."""
        elbos = [float(self.model.classic_elbo())]
        nlpds, rmses = [], []
        while (len(elbos) - 1) < self.max_itr_sites_optim:
            self.model.update_data_sites(self.data_sites_lr)
            self.model.update_girsanov_sites(self.girsanov_sites_lr)
            elbos.append(float(self.model.classic_elbo()))
            nl, rm = self._nlpd_rmse()
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
        elbo_vals, nlpd_vals, rmse_vals = [float(self.model.classic_elbo())], [], []
        relin = hasattr(self.model, "relinearize")
        for i in range(self.max_itr):
            before = float(self.model.classic_elbo())
            e, n, r = self._optimize_sites_under_stable_prior()
            # (the reference swaps to the unclipped linearised prior here; the posterior, hence the ELBO under the SDE
            #  prior, is unchanged by the site transformation)
            after = float(self.model.classic_elbo())
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
        elbo_vals = [float(self.model.classic_elbo())]
        n0, r0 = self._nlpd_rmse()
        nlpd_vals, rmse_vals = [n0], [r0]
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
        elbos, nlpds, rmses = [float(mdl.elbo(mS))], [], []
        q_lr, x0_lr = self.q_lr, self.x0_lr
        for i in range(self.max_itr):
            # the marginals that close an iteration (for its ELBO) are those the next one starts from: the parameters do not change
            # in between, so the reference's second forward_pass per iteration (vi_markov_gp_trainer.py:60) is not repeated
            mdl.update_lagrange_and_param(mS, lr=q_lr)
            if i > self.warmup_x0_itr:
                mdl.update_initial_statistics(lr=x0_lr)
            mS = mdl._forward_packed()
            elbos.append(float(mdl.elbo(mS)))
            nl, rm = self._metrics(*mS)
            nlpds.append(nl)
            rmses.append(rm)
            if elbos[-2] > elbos[-1] or abs(elbos[-2] - elbos[-1]) < self.lr_tol:
                q_lr /= 10
                x0_lr /= 10
            if abs(elbos[-2] - elbos[-1]) < self.optim_tol:
                break
        return elbos[1:], nlpds, rmses

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
            P0 = torch.from_numpy(mdl.p0_cov).to(mdl.device)
            mean = mdl.p0_mu - self.prior_initial_state_lr * g_loc.cpu().numpy()
            M = (linalg.cholesky(P0) - self.prior_initial_state_lr * g_scale).cpu().numpy()
        mdl.set_prior_initial_state(mean, M @ M)

    def optimize_prior_sde(self):
        """vi_markov_gp_trainer.py:163-201: Adam on the drift parameters with dE_sde/d params at the current (m, S)."""
        mdl, sde = self.model, self.model.prior_sde
        elbo_vals, nlpd_vals, rmse_vals = [float(mdl.elbo())], [], []
        for _ in range(self.learning_max_itr):
            grads = mdl.grad_prior_sde_params()
            names = sde.trainable_variables
            for n, v in zip(names, self.prior_sde_optim.step([sde.get(n) for n in names], grads)):
                sde.assign(n, v)
            mdl._refresh_drift_params()
            if self.optimize_prior_initial_state:
                self.optimize_prior_x0()
            mS = mdl._forward_packed()
            elbo_vals.append(float(mdl.elbo(mS)))
            nl, rm = self._metrics(*mS)
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
        elbo_vals = [float(mdl.elbo(mS))]
        n0, r0 = self._metrics(*mS)
        nlpd_vals, rmse_vals = [n0], [r0]
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
