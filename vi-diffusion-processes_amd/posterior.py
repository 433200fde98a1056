"""
Host-side mirror of markovflow/posterior.py `ConditionalProcess` (posterior.py:166-260): the posterior process
q(s(.)) = int p(s(.) | s(Z)) q(s(Z)) ds(Z) evaluated at arbitrary sorted time points.
"""

import os

from .conditionals import conditional_predict, pairwise_marginals


class ConditionalProcess:
    def __init__(self, posterior_dist, kernel, conditioning_time_points, mean_function=None):
        self.gauss_markov_model = posterior_dist
        self.kernel = kernel
        self.conditioning_time_points = conditioning_time_points
        self.mean_function = mean_function

    def predict_state(self, new_time_points):
        """posterior.py:207-229.  One chain, 1-d query points: the pair gather and P S P^T run in one HIP kernel per query set
        (mfgm_cond_predict) on the marginal blocks of the posterior; otherwise the batched torch route."""
        q = self.gauss_markov_model
        if (new_time_points.dim() == 1 and self.conditioning_time_points.dim() == 1 and tuple(q.batch_shape) == () and new_time_points.is_cuda
                and q.d <= 32 and os.environ.get("VIDP_FUSED_PREDICT", "1") != "0"):
            return self._predict_state_fused(new_time_points)
        pw_mu, pw_cov = pairwise_marginals(q, self.kernel.initial_mean(q.batch_shape), self.kernel.initial_covariance_matrix())
        return conditional_predict(new_time_points, self.conditioning_time_points, self.kernel, pw_mu, pw_cov)

    def _predict_state_fused(self, new_time_points):
        import torch
        from . import _lib
        from ._lib import FULL, SYM, VEC
        from .conditionals import _conditional_statistics
        from .packed import _ptr, _stream
        q = self.gauss_markov_model
        pl, T, d = q.plan, q.T, q.d
        P, Tc, idx = _conditional_statistics(new_time_points, self.conditioning_time_points, self.kernel)
        s = q._posterior_packed()["s"]
        if pl.d > 8:          # wide plans: the packed arrays are the natural ones
            mu, Sig, Sub = s["x"].view(T, d), s["Sig"].view(T, d, d), s["Sub"].view(T, d, d)
        else:
            mu, Sig = pl.unpack(VEC, s["x"])[0], pl.unpack(SYM, s["Sig"])[0]
            Sub = torch.zeros((T, d, d), dtype=torch.float64, device=pl.device)
            if T > 1:
                Sub[:T - 1] = pl.unpack(FULL, s["Sub"], T - 1)[0]
        N = int(new_time_points.shape[0])
        dev = new_time_points.device
        pm = self.kernel.initial_mean(()).to(dev, torch.float64).contiguous()
        pc = self.kernel.initial_covariance_matrix().to(dev, torch.float64).contiguous()
        mean = torch.empty((N, d), dtype=torch.float64, device=dev)
        cov = torch.empty((N, d, d), dtype=torch.float64, device=dev)
        idx32 = idx.to(torch.int32).contiguous()
        _lib.check(pl.lib.mfgm_cond_predict(T, d, N, _ptr(idx32), _ptr(P.contiguous()), _ptr(Tc.contiguous()), _ptr(pm), _ptr(pc),
                                            _ptr(mu.contiguous()), _ptr(Sig.contiguous()), _ptr(Sub.contiguous()), _ptr(mean), _ptr(cov),
                                            _stream()), "mfgm_cond_predict")
        return mean, cov

    def predict_f(self, new_time_points, full_output_cov=False):
        """posterior.py:231-260 (zero mean function unless one is supplied)."""
        em = self.kernel.generate_emission_model(new_time_points)
        m, S = self.predict_state(new_time_points)
        f, fc = em.project_state_to_f(m), em.project_state_covariance_to_f(S, full_output_cov)
        if self.mean_function is not None:
            f = f + self.mean_function(new_time_points)
        return f, fc
