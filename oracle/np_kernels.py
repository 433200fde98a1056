"""
Oracle (test infrastructure, see oracle/__init__.py): stationary SDE kernels -> SSM parameters,
restating markovflow/kernels/{sde_kernel.py, matern.py} in NumPy (closed-form matrix exponentials).
"""
import numpy as np

from .np_ssm import state_space_model_from_covariances


class StationaryKernel:
    """sde_kernel.py:367-475 (StationaryKernel) restricted to what the hot path uses."""

    state_dim = None

    def __init__(self, jitter=0.0, state_mean=None):
        self.jitter = jitter
        self._state_mean = None if state_mean is None else np.asarray(state_mean, dtype=np.float64)

    @property
    def state_mean(self):
        return np.zeros(self.state_dim) if self._state_mean is None else self._state_mean

    def feedback_matrix(self):
        raise NotImplementedError

    def steady_state_covariance(self):
        raise NotImplementedError

    def state_transitions(self, time_deltas):
        raise NotImplementedError

    def transition_statistics(self, time_deltas):
        """sde_kernel.py:421-446: Q_k = Pinf - A_k Pinf A_k^T (+ jitter)."""
        A = self.state_transitions(time_deltas)
        Pinf = self.steady_state_covariance()
        Q = Pinf - A @ Pinf @ np.swapaxes(A, -1, -2)
        return A, Q + self.jitter * np.eye(self.state_dim)

    def state_offsets(self, time_deltas):
        """sde_kernel.py:460-475: b_k = (I - A_k) m."""
        A = self.state_transitions(time_deltas)
        return np.einsum("...ij,j->...i", -(A - np.eye(self.state_dim)), self.state_mean)

    def initial_covariance(self, batch_shape=()):
        Pinf = self.steady_state_covariance() + self.jitter * np.eye(self.state_dim)
        return np.broadcast_to(Pinf, tuple(batch_shape) + Pinf.shape).copy()

    def initial_mean(self, batch_shape=()):
        return np.broadcast_to(self.state_mean, tuple(batch_shape) + (self.state_dim,)).copy()

    def state_space_model(self, time_points):
        """sde_kernel.py:153-171."""
        t = np.asarray(time_points, dtype=np.float64)
        dt = t[..., 1:] - t[..., :-1]
        A, Q = self.transition_statistics(dt)
        return state_space_model_from_covariances(
            self.initial_mean(t.shape[:-1]), self.initial_covariance(t.shape[:-1]), A,
            self.state_offsets(dt), Q)

    def emission_vector(self):
        """sde_kernel.py:173-211: H = [1, 0, ...] (output_dim 1)."""
        h = np.zeros((1, self.state_dim))
        h[0, 0] = 1.0
        return h

    def emission_matrix(self, time_points):
        t = np.asarray(time_points)
        return np.broadcast_to(self.emission_vector(), t.shape + (1, self.state_dim)).copy()


class Matern12(StationaryKernel):
    """matern.py:27-127."""
    state_dim = 1

    def __init__(self, lengthscale, variance, jitter=0.0):
        super().__init__(jitter)
        self.lengthscale, self.variance = float(lengthscale), float(variance)

    def state_transitions(self, time_deltas):
        return np.exp(-np.asarray(time_deltas) / self.lengthscale)[..., None, None]

    def feedback_matrix(self):
        return np.array([[-1.0 / self.lengthscale]])

    def steady_state_covariance(self):
        return np.array([[self.variance]])


class OrnsteinUhlenbeck(StationaryKernel):
    """matern.py:130-234: decay lambda, diffusion q; Pinf = q / (2 lambda)."""
    state_dim = 1

    def __init__(self, decay, diffusion, jitter=0.0):
        super().__init__(jitter)
        self.decay, self.diffusion = float(decay), float(diffusion)

    def state_transitions(self, time_deltas):
        return np.exp(-np.asarray(time_deltas) * self.decay)[..., None, None]

    def feedback_matrix(self):
        return np.array([[-self.decay]])

    def steady_state_covariance(self):
        return np.array([[self.diffusion / (2.0 * self.decay)]])


class Matern32(StationaryKernel):
    """matern.py:237-373."""
    state_dim = 2

    def __init__(self, lengthscale, variance, jitter=0.0):
        super().__init__(jitter)
        self.lengthscale, self.variance = float(lengthscale), float(variance)
        self.lam = np.sqrt(3.0) / self.lengthscale

    def feedback_matrix(self):
        return np.array([[0.0, 1.0], [-self.lam ** 2, -2.0 * self.lam]])

    def state_transitions(self, time_deltas):
        dt = np.asarray(time_deltas)[..., None, None]
        N = (self.feedback_matrix() + self.lam * np.eye(2)) * dt
        return np.exp(-self.lam * dt) * (np.eye(2) + N)

    def steady_state_covariance(self):
        return self.variance * np.array([[1.0, 0.0], [0.0, self.lam ** 2]])


class Matern52(StationaryKernel):
    """matern.py:376-520."""
    state_dim = 3

    def __init__(self, lengthscale, variance, jitter=0.0):
        super().__init__(jitter)
        self.lengthscale, self.variance = float(lengthscale), float(variance)
        self.lam = np.sqrt(5.0) / self.lengthscale

    def feedback_matrix(self):
        l = self.lam
        return np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-l ** 3, -3.0 * l ** 2, -3.0 * l]])

    def state_transitions(self, time_deltas):
        dt = np.asarray(time_deltas)[..., None, None]
        N = (self.feedback_matrix() + self.lam * np.eye(3)) * dt
        return np.exp(-self.lam * dt) * (np.eye(3) + N + N @ N / 2.0)

    def steady_state_covariance(self):
        l23 = self.lam ** 2 / 3.0
        return self.variance * np.array([[1.0, 0.0, -l23], [0.0, l23, 0.0], [-l23, 0.0, self.lam ** 4]])


class Sum(StationaryKernel):
    """sde_kernel.py:540-687 (ConcatKernel / Sum): block-diagonal state, H = concatenated emissions."""

    def __init__(self, kernels, jitter=0.0):
        super().__init__(jitter)
        self.kernels = list(kernels)
        self.state_dim = sum(k.state_dim for k in self.kernels)

    @staticmethod
    def _block_diag(mats):
        batch = np.broadcast_shapes(*[m.shape[:-2] for m in mats])
        n = sum(m.shape[-1] for m in mats)
        out = np.zeros(batch + (n, n))
        i = 0
        for m in mats:
            k = m.shape[-1]
            out[..., i:i + k, i:i + k] = m
            i += k
        return out

    @property
    def state_mean(self):
        return np.concatenate([k.state_mean for k in self.kernels])

    def state_transitions(self, time_deltas):
        return self._block_diag([k.state_transitions(time_deltas) for k in self.kernels])

    def steady_state_covariance(self):
        return self._block_diag([k.steady_state_covariance() for k in self.kernels])

    def feedback_matrix(self):
        return self._block_diag([k.feedback_matrix() for k in self.kernels])

    def transition_statistics(self, time_deltas):
        stats = [k.transition_statistics(time_deltas) for k in self.kernels]
        A = self._block_diag([s[0] for s in stats])
        Q = self._block_diag([s[1] for s in stats])
        return A, Q + self.jitter * np.eye(self.state_dim)

    def emission_vector(self):
        return np.concatenate([k.emission_vector() for k in self.kernels], axis=-1)
