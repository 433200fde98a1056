"""
Host-side mirror of markovflow/models/sparse_variational_cvi.py `SparseCVIGaussianProcess`
(sparse_variational_cvi.py:38-292): CVI with sites on pairs of consecutive inducing states.  The overlap-add of the
[M+1, 2d, 2d] sites into the block-tri-diagonal natural parameters and the data -> site segment sums (a Python list of
M+1 reduce_sums in the reference, :199-213) are single index_add operations; the posterior refresh runs in the HIP sweeps.
"""
import torch

from ._lib import FULL, SYM, VEC
from .conditionals import conditional_statistics
from .posterior import ConditionalProcess
from .ssm_gaussian_transformations import naturals_to_ssm_params_packed
from .variational_cvi import back_project_nats


class SparseCVIGaussianProcess:
    def __init__(self, kernel, inducing_points, likelihood, mean_function=None, learning_rate=0.1):
        self._kernel, self._likelihood = kernel, likelihood
        self.learning_rate = learning_rate
        self.inducing_inputs = inducing_points
        M, sd = inducing_points.shape[-1], kernel.state_dim
        dev, dt = inducing_points.device, torch.float64
        self.nat1 = torch.zeros((M + 1, 2 * sd), dtype=dt, device=dev)
        self.nat2 = torch.zeros((M + 1, 2 * sd, 2 * sd), dtype=dt, device=dev)
        self._dist_p = None

    @property
    def kernel(self):
        return self._kernel

    @property
    def likelihood(self):
        return self._likelihood

    @property
    def dist_p(self):
        if self._dist_p is None:
            self._dist_p = self._kernel.state_space_model(self.inducing_inputs)
        return self._dist_p

    @property
    def dist_q(self):
        """Prior naturals + overlap-added site naturals (sparse_variational_cvi.py:140-174)."""
        p = self.dist_p
        pl, sd = p.plan, self._kernel.state_dim
        if getattr(self, "_p_nat", None) is None:      # the prior does not change between site updates
            self._p_nat = pl.ssm_to_naturals(p.packed.A, p.packed.off, p.packed.chol)
        nat = self._p_nat
        lin = self.nat1[1:, :sd] + self.nat1[:-1, sd:]
        diag = self.nat2[1:, :sd, :sd] + self.nat2[:-1, sd:, sd:]
        sub = 2.0 * self.nat2[1:-1, sd:, :sd]
        tl = pl.pack(VEC, lin[None].contiguous())
        td = pl.pack(SYM, diag[None].contiguous())
        ts = pl.pack(FULL, sub[None].contiguous()) if p.T > 1 else pl.zeros(FULL)
        pl.lincomb(td, 1.0, td, 1.0, nat["diag"])
        pl.lincomb(ts, 1.0, ts, 1.0, nat["sub"])
        q = naturals_to_ssm_params_packed(pl, tl, td, ts)
        q.batch_shape = p.batch_shape
        return q

    @property
    def posterior(self):
        return ConditionalProcess(self.dist_q, self._kernel, self.inducing_inputs)

    def local_objective_and_gradients(self, Fmu, Fvar, Y):
        obj = self._likelihood.variational_expectations(Fmu, Fvar, Y).sum()
        return obj, self._likelihood.ve_gradients_expectation(Fmu, Fvar, Y)

    def update_sites(self, input_data):
        """theta_m <- (1 - rho) theta_m + rho g_m, g_m = data gradients projected through p(f_k | v_m) (sparse_variational_cvi.py:176-221)."""
        time_points, observations = input_data
        fx_mus, fx_covs = self.posterior.predict_f(time_points)
        _, grads = self.local_objective_and_gradients(fx_mus, fx_covs, observations)
        H = self._kernel.generate_emission_model(time_points).emission_matrix
        P, _ = conditional_statistics(time_points, self.inducing_inputs, self._kernel)
        theta_linear, lik_nat2 = back_project_nats(grads[0], grads[1], H @ P)
        idx = torch.searchsorted(self.inducing_inputs.contiguous(), time_points.contiguous())
        s1 = torch.zeros_like(self.nat1).index_add_(0, idx, theta_linear)
        s2 = torch.zeros_like(self.nat2).index_add_(0, idx, lik_nat2)
        lr = self.learning_rate
        self.nat1 = (1 - lr) * self.nat1 + lr * s1
        self.nat2 = (1 - lr) * self.nat2 + lr * s2

    def classic_elbo(self, input_data):
        """sum_i E_q log p(y_i | f_i) - KL[q(s_Z) || p(s_Z)] (sparse_variational_cvi.py:270-292)."""
        time_points, observations = input_data
        q = self.dist_q
        fx_mus, fx_covs = ConditionalProcess(q, self._kernel, self.inducing_inputs).predict_f(time_points)
        ve = self._likelihood.variational_expectations(fx_mus, fx_covs, observations).sum()
        return ve - q.kl_divergence(self.dist_p).sum()

    def loss(self, input_data):
        return -self.classic_elbo(input_data)
