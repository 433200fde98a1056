"""
Host-side mirror of the stationary SDE kernels on the path (markovflow/kernels/matern.py: `Matern12`,
`OrnsteinUhlenbeck`, `Matern32`, `Matern52`; markovflow/kernels/sde_kernel.py: `StationaryKernel`, `Sum`).
`state_space_model(time_points)` evaluates the closed-form matrix exponentials, the process covariances and their
Cholesky factors in one HIP kernel (k_stationary_ssm) directly in the packed layout.
"""
import math

import torch

from . import _lib, linalg
from .emission_model import EmissionModel
from .packed import Plan
from .state_space_model import StateSpaceModel, _flat
from .variational_cvi_sde import _ssm_from_packed


class StationaryKernel:
    """kernels/sde_kernel.py:367-475."""

    state_dim = None

    def __init__(self, output_dim=1, jitter=0.0, state_mean=None):
        if output_dim != 1:
            raise ValueError("only output_dim == 1 kernels are on the hot path")
        self.output_dim = output_dim
        self.jitter = float(jitter)
        self._state_mean = state_mean

    # components: list of (order, lam, var) -------------------------------------------------------------
    def _components(self):
        raise NotImplementedError

    @property
    def state_mean(self):
        if self._state_mean is None:
            return torch.zeros(self.state_dim, dtype=torch.float64)
        return torch.as_tensor(self._state_mean, dtype=torch.float64).reshape(self.state_dim)

    def set_state_mean(self, state_mean, trainable=False):
        self._state_mean = state_mean

    def _spec(self):
        comps = self._components()
        if len(comps) > 8 or self.state_dim > 8:
            raise ValueError("the HIP path supports up to 8 components and state_dim <= 8")
        spec = _lib.KernelSpec()
        spec.ncomp = len(comps)
        off = 0
        for i, (order, lam, var) in enumerate(comps):
            spec.order[i], spec.offset[i], spec.lam[i], spec.var[i] = order, off, lam, var
            off += order
        m = self.state_mean
        for i in range(self.state_dim):
            spec.mean[i] = float(m[i])
        spec.jitter = self.jitter
        return spec

    def _block_diag(self, blocks):
        return torch.block_diag(*blocks)

    @property
    def steady_state_covariance(self):
        blocks = []
        for order, lam, var in self._components():
            if order == 1:
                blocks.append(torch.tensor([[var]], dtype=torch.float64))
            elif order == 2:
                blocks.append(var * torch.tensor([[1.0, 0.0], [0.0, lam ** 2]], dtype=torch.float64))
            else:
                l23 = lam ** 2 / 3.0
                blocks.append(var * torch.tensor([[1.0, 0.0, -l23], [0.0, l23, 0.0], [-l23, 0.0, lam ** 4]], dtype=torch.float64))
        return self._block_diag(blocks)

    @property
    def feedback_matrix(self):
        blocks = []
        for order, lam, var in self._components():
            if order == 1:
                blocks.append(torch.tensor([[-lam]], dtype=torch.float64))
            elif order == 2:
                blocks.append(torch.tensor([[0.0, 1.0], [-lam ** 2, -2.0 * lam]], dtype=torch.float64))
            else:
                blocks.append(torch.tensor([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-lam ** 3, -3.0 * lam ** 2, -3.0 * lam]],
                                           dtype=torch.float64))
        return self._block_diag(blocks)

    def state_space_model(self, time_points, plan=None):
        """SDEKernel.state_space_model (sde_kernel.py:153-171): prior SSM at the given (sorted) time points [..., T]."""
        t, bs = _flat(time_points, 1)
        B, T = t.shape
        if plan is None:
            plan = Plan(B, T, self.state_dim, device=t.device)
        dts = (t[:, 1:] - t[:, :-1]).contiguous()
        if self.state_dim > 8:
            return self._state_space_model_wide(dts, bs, plan)
        A, off, chol = plan.stationary_ssm(self._spec(), dts)
        plan.check_info()
        ssm = _ssm_from_packed(plan, A, off, chol)
        ssm.batch_shape = bs
        return ssm

    def _state_space_model_wide(self, dts, bs, plan):
        """
        state_dim > 8 (e.g. the reference's Sum of ten Matern-5/2, d = 30): the fused k_stationary_ssm kernel is
        specialised for d <= 8, so this one-off model construction uses the per-element closed forms in torch
        (block-diagonal A, Q = Pinf - A Pinf A^T + jitter, cholesky_or_zero, b = (I - A) m); the sweeps that follow
        run in the wide HIP kernels.
        """
        dev = dts.device
        d = self.state_dim
        A, Q = self.transition_statistics_local(dts)
        zero = (Q == 0).all(dim=-1).all(dim=-1)
        eye = torch.eye(d, dtype=Q.dtype, device=dev)
        chol = linalg.cholesky(torch.where(zero[..., None, None], eye, Q))
        chol = torch.where(zero[..., None, None], torch.zeros_like(chol), chol)
        m = self.state_mean.to(dev)
        off = m - torch.einsum("...ij,j->...i", A, m)
        B = dts.shape[0]
        P0 = self.initial_covariance_matrix().to(dev)
        ssm = StateSpaceModel(m.expand(B, d).contiguous(), linalg.cholesky(P0).expand(B, d, d).contiguous(), A, off, chol,
                              plan=plan)
        ssm.batch_shape = bs
        return ssm

    def transition_statistics(self, transition_times, time_deltas):
        """(A_k, Q_k) (sde_kernel.py:421-446), natural tensors."""
        td, bs = _flat(time_deltas, 1)
        B, N = td.shape
        t = torch.cat([torch.zeros((B, 1), dtype=td.dtype, device=td.device), torch.cumsum(td, dim=1)], dim=1)
        ssm = self.state_space_model(t)
        A = ssm.state_transitions
        c = ssm.cholesky_process_covariances
        return A.reshape(bs + tuple(A.shape[1:])), (c @ c.transpose(-1, -2)).reshape(bs + tuple(c.shape[1:]))

    def state_transitions(self, transition_times, time_deltas):
        return self.transition_statistics(transition_times, time_deltas)[0]

    def transition_statistics_local(self, time_deltas):
        """
        (A, Q) for arbitrary, unordered time gaps (any shape [...]) as per-element closed forms in torch: used by the
        conditionals (prediction between conditioning points, conditionals.py:207-256), which are embarrassingly parallel
        over query points.  Same formulas as k_stationary_ssm.
        """
        dt = time_deltas[..., None, None]
        blocks = []
        for order, lam, var in self._components():
            ex = torch.exp(-lam * dt)
            if order == 1:
                blocks.append(ex * torch.ones((1, 1), dtype=dt.dtype, device=dt.device))
            elif order == 2:
                N = torch.tensor([[lam, 1.0], [-lam ** 2, -lam]], dtype=dt.dtype, device=dt.device)
                blocks.append(ex * (torch.eye(2, dtype=dt.dtype, device=dt.device) + N * dt))
            else:
                N = torch.tensor([[lam, 1.0, 0.0], [0.0, lam, 1.0], [-lam ** 3, -3.0 * lam ** 2, -2.0 * lam]], dtype=dt.dtype,
                                 device=dt.device)
                blocks.append(ex * (torch.eye(3, dtype=dt.dtype, device=dt.device) + N * dt + (N @ N) * (0.5 * dt * dt)))
        d = self.state_dim
        A = torch.zeros(tuple(time_deltas.shape) + (d, d), dtype=dt.dtype, device=dt.device)
        o = 0
        for blk in blocks:
            k = blk.shape[-1]
            A[..., o:o + k, o:o + k] = blk
            o += k
        Pinf = self.steady_state_covariance.to(dt.device)
        Q = Pinf - A @ Pinf @ A.transpose(-1, -2) + self.jitter * torch.eye(d, dtype=dt.dtype, device=dt.device)
        return A, Q

    # -- hyper-parameters as leaves of a torch graph (the reference differentiates classic_elbo through the kernel's tf.Variables with a
    #    GradientTape, tests/integration/models/test_variational_cvi.py:93-110) ------------------------------------------------------------
    def hyperparameter_leaves(self, device="cpu"):
        """{name: 0-dim tensor with requires_grad} of this kernel's trainable hyper-parameters (a list of such dicts for a Sum)."""
        raise NotImplementedError

    def _components_t(self, leaves):
        """[(order, lam, var)] with lam / var torch expressions of the leaves (the differentiable twin of _components)."""
        raise NotImplementedError

    def differentiable_ssm(self, time_points, leaves=None, plan=None):
        """(tape.TapeSSM, leaves): the prior state-space model at the sorted time points [T] (one chain) as a differentiable function of
        the hyper-parameter leaves -- the closed forms of k_stationary_ssm (A = e^{-lam dt}(I + N dt + N^2 dt^2 / 2), Q = Pinf - A Pinf A^T
        + jitter) in torch, d <= 8; everything sequential in time downstream (marginals, log-determinants) goes through vidp_amd.tape,
        i.e. the HIP sweeps with exact backward passes."""
        from . import tape
        t = time_points.reshape(-1)
        dev = t.device
        if leaves is None:
            leaves = self.hyperparameter_leaves(dev)
        dt = (t[1:] - t[:-1])[:, None, None]
        d = self.state_dim
        A = torch.zeros((t.numel() - 1, d, d), dtype=torch.float64, device=dev)
        Pinf = torch.zeros((d, d), dtype=torch.float64, device=dev)
        o = 0
        for order, lam, var in self._components_t(leaves):
            ex = torch.exp(-lam * dt)
            eye = torch.eye(order, dtype=torch.float64, device=dev)
            one, zero = torch.ones_like(lam), torch.zeros_like(lam)
            if order == 1:
                blk, pinf = ex * eye, var.reshape(1, 1)
            elif order == 2:
                N = torch.stack([torch.stack([lam, one]), torch.stack([-lam ** 2, -lam])])
                blk = ex * (eye + N * dt)
                pinf = var * torch.stack([torch.stack([one, zero]), torch.stack([zero, lam ** 2])])
            else:
                N = torch.stack([torch.stack([lam, one, zero]), torch.stack([zero, lam, one]),
                                 torch.stack([-lam ** 3, -3.0 * lam ** 2, -2.0 * lam])])
                blk = ex * (eye + N * dt + (N @ N) * (0.5 * dt * dt))
                l23 = lam ** 2 / 3.0
                pinf = var * torch.stack([torch.stack([one, zero, -l23]), torch.stack([zero, l23, zero]), torch.stack([-l23, zero, lam ** 4])])
            A = A + torch.nn.functional.pad(blk, (o, d - o - order, o, d - o - order))
            Pinf = Pinf + torch.nn.functional.pad(pinf, (o, d - o - order, o, d - o - order))
            o += order
        jit = self.jitter * torch.eye(d, dtype=torch.float64, device=dev)
        Q = Pinf - A @ Pinf @ A.transpose(-1, -2) + jit
        m = self.state_mean.to(dev)
        b = m - (A @ m[:, None])[..., 0]
        ssm = tape.TapeSSM(m[None], tape.cholesky(Pinf + jit)[None], A[None], b[None], tape.cholesky(0.5 * (Q + Q.transpose(-1, -2)))[None], plan=plan)
        return ssm, leaves

    def initial_mean(self, batch_shape=()):
        return self.state_mean.expand(tuple(batch_shape) + (self.state_dim,))

    def initial_covariance_matrix(self):
        """Pinf + jitter (sde_kernel.py:402-419)."""
        return self.steady_state_covariance + self.jitter * torch.eye(self.state_dim, dtype=torch.float64)

    def generate_emission_model(self, time_points):
        """H = [1, 0, ...] per component, tiled over the time points (sde_kernel.py:173-211, 670-687)."""
        h = torch.zeros((1, self.state_dim), dtype=torch.float64, device=time_points.device)
        off = 0
        for order, _, _ in self._components():
            h[0, off] = 1.0
            off += order
        return EmissionModel(h.expand(tuple(time_points.shape) + (1, self.state_dim)).contiguous(), constant_matrix=h)


def _check(lengthscale, variance):
    if lengthscale <= 0.0 or variance <= 0.0:
        raise ValueError("lengthscale and variance must be positive")     # matern.py `_check_lengthscale_and_variance`


class Matern12(StationaryKernel):
    """matern.py:27-127."""
    state_dim = 1

    def __init__(self, lengthscale, variance, output_dim=1, jitter=0.0):
        super().__init__(output_dim, jitter)
        _check(lengthscale, variance)
        self.lengthscale, self.variance = float(lengthscale), float(variance)

    def _components(self):
        return [(1, 1.0 / self.lengthscale, self.variance)]

    def hyperparameter_leaves(self, device="cpu"):
        mk = lambda v: torch.tensor(float(v), dtype=torch.float64, device=device, requires_grad=True)
        return {"lengthscale": mk(self.lengthscale), "variance": mk(self.variance)}

    def _components_t(self, leaves):
        return [(1, 1.0 / leaves["lengthscale"], leaves["variance"])]


class OrnsteinUhlenbeck(StationaryKernel):
    """matern.py:130-234: decay lambda, diffusion q, Pinf = q / (2 lambda)."""
    state_dim = 1

    def __init__(self, decay, diffusion, output_dim=1, jitter=0.0):
        super().__init__(output_dim, jitter)
        _check(decay, diffusion)
        self.decay, self.diffusion = float(decay), float(diffusion)

    def _components(self):
        return [(1, self.decay, self.diffusion / (2.0 * self.decay))]

    def hyperparameter_leaves(self, device="cpu"):
        mk = lambda v: torch.tensor(float(v), dtype=torch.float64, device=device, requires_grad=True)
        return {"decay": mk(self.decay), "diffusion": mk(self.diffusion)}

    def _components_t(self, leaves):
        return [(1, leaves["decay"], leaves["diffusion"] / (2.0 * leaves["decay"]))]


class Matern32(StationaryKernel):
    """matern.py:237-373."""
    state_dim = 2

    def __init__(self, lengthscale, variance, output_dim=1, jitter=0.0):
        super().__init__(output_dim, jitter)
        _check(lengthscale, variance)
        self.lengthscale, self.variance = float(lengthscale), float(variance)

    def _components(self):
        return [(2, math.sqrt(3.0) / self.lengthscale, self.variance)]

    def hyperparameter_leaves(self, device="cpu"):
        mk = lambda v: torch.tensor(float(v), dtype=torch.float64, device=device, requires_grad=True)
        return {"lengthscale": mk(self.lengthscale), "variance": mk(self.variance)}

    def _components_t(self, leaves):
        return [(2, math.sqrt(3.0) / leaves["lengthscale"], leaves["variance"])]


class Matern52(StationaryKernel):
    """matern.py:376-520."""
    state_dim = 3

    def __init__(self, lengthscale, variance, output_dim=1, jitter=0.0):
        super().__init__(output_dim, jitter)
        _check(lengthscale, variance)
        self.lengthscale, self.variance = float(lengthscale), float(variance)

    def _components(self):
        return [(3, math.sqrt(5.0) / self.lengthscale, self.variance)]

    def hyperparameter_leaves(self, device="cpu"):
        mk = lambda v: torch.tensor(float(v), dtype=torch.float64, device=device, requires_grad=True)
        return {"lengthscale": mk(self.lengthscale), "variance": mk(self.variance)}

    def _components_t(self, leaves):
        return [(3, math.sqrt(5.0) / leaves["lengthscale"], leaves["variance"])]


class Sum(StationaryKernel):
    """sde_kernel.py:540-687 (ConcatKernel / Sum of stationary kernels): block-diagonal state, summed emissions."""

    def __init__(self, kernels, jitter=0.0):
        super().__init__(1, jitter)
        self.kernels = list(kernels)
        self.state_dim = sum(k.state_dim for k in self.kernels)

    @property
    def state_mean(self):
        return torch.cat([k.state_mean for k in self.kernels])

    def _components(self):
        out = []
        for k in self.kernels:
            out.extend(k._components())
        return out

    def hyperparameter_leaves(self, device="cpu"):
        return [k.hyperparameter_leaves(device) for k in self.kernels]

    def _components_t(self, leaves):
        out = []
        for k, lv in zip(self.kernels, leaves):
            out.extend(k._components_t(lv))
        return out

    def _spec(self):
        spec = super()._spec()
        # each component kernel adds its own jitter to its Q block; Sum adds its own on top (sde_kernel.py:640-656)
        if any(k.jitter != 0.0 for k in self.kernels):
            raise ValueError("per-component jitter inside Sum is not supported on the HIP path; set it on the Sum")
        return spec
