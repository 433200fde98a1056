"""
Host-side mirror of markovflow/sde/sde.py for the drifts on the path: `OrnsteinUhlenbeckSDE` (sde.py:134-176) and
`DoubleWellSDE` (sde.py:179-224).  Both drifts act per dimension and their Euler map is a cubic
u(x) = x + dt f(x) = alpha x - beta x^3, which is what the HIP kernels consume (closed-form Gaussian moments in
place of the reference's 10^d / 20^d-point Gauss-Hermite grids).  The diffusion matrix q must be diagonal for d > 1.
"""
import math

import numpy as np
import torch

from . import _lib


class SDE:
    """sde.py:24-131."""

    kind = 0        # drift family understood by the HIP kernels (mfgm_sde_params.kind): 0 = per-dimension cubic

    def __init__(self, q):
        q = torch.as_tensor(q, dtype=torch.float64)
        if q.dim() == 0:
            q = q.reshape(1, 1)
        self.q = q
        self.state_dim = q.shape[0]
        qd = torch.diagonal(q)
        # a full (non-diagonal) diffusion matrix is served by the tensor-product quadrature kernels (quad_params,
        # variational_cvi_sde.CVISitesSDEQuadrature); the closed-form kernels (params) need a diagonal one
        self.q_is_diagonal = bool(self.state_dim == 1 or torch.equal(torch.diag(qd), q.cpu() if q.is_cuda else q))
        self.q_diag = [float(v) for v in qd]

    def drift(self, x, t=None):
        raise NotImplementedError

    def gradient_drift(self, x, t=None):
        raise NotImplementedError

    def diffusion(self, x, t=None):
        """l(x, t) = sqrt(q) (sde.py:165-176)."""
        return torch.ones_like(x[..., None]) * torch.sqrt(self.q.to(x.device))     # q is diagonal (see __init__)

    def cubic(self, dt):
        """(alpha, beta) of the Euler map u(x) = x + dt f(x) = alpha x - beta x^3."""
        raise NotImplementedError

    # -- trainable drift parameters (tf.Variable(trainable=...) in the reference, sde.py:150, 197-198) ---------------------
    _param_names = ()

    @property
    def trainable_variables(self):
        """Names of the drift parameters being learnt, in the reference's variable order."""
        return [n for n in self._param_names if getattr(self, "_trainable", {}).get(n, False)]

    def get(self, name):
        return float(getattr(self, name))

    def assign(self, name, value):
        setattr(self, name, float(value))

    def cubic_jacobian(self, dt):
        """{parameter: (d alpha / d parameter, d beta / d parameter)}."""
        raise NotImplementedError

    def drift_cubic_jacobian(self):
        """{parameter: (d af / d parameter, d bf / d parameter)} of f(x) = af x - bf x^3."""
        raise NotImplementedError

    # -- the tensor-product quadrature kernels (csrc/mfgm_quad.h): coupled / non-polynomial drifts, full diffusion matrices ------------
    quad_kind = 12      # per-dimension cubic f = af x - bf x^3

    def quad_theta(self):
        """Drift parameters as the quadrature kernels take them (mfgm_quad_drift.theta) and the hidden width (kind 11)."""
        return list(self.drift_cubic()), 0

    def quad_param_jacobian(self):
        """{trainable parameter: [d theta_k / d parameter]}: how the kernels' parameter gradient maps onto `trainable_variables`."""
        jac = self.drift_cubic_jacobian()
        return {n: list(jac[n]) for n in self.trainable_variables}

    def quad_params(self, dt, init_mu, init_cov, clip=None):
        """Fill mfgm_quad_drift."""
        d = self.state_dim
        if d > 3:
            raise ValueError("the quadrature kernels cover state dimensions up to 3 (20^d nodes per time step)")
        th, nh = self.quad_theta()
        if len(th) > _lib.QUAD_NTHETA:
            raise ValueError("too many drift parameters for the quadrature kernels")
        prm = _lib.QuadDrift()
        prm.kind, prm.d, prm.nh, prm.dt = int(self.quad_kind), d, int(nh), float(dt)
        for k, v in enumerate(th):
            prm.theta[k] = float(v)
        Qp = dt * self.q.cpu().numpy().astype(np.float64)
        W = np.linalg.inv(Qp)
        P0 = np.asarray(init_cov, dtype=np.float64).reshape(d, d)
        P0inv = np.linalg.inv(P0)
        for i in range(d):
            prm.mu0[i] = float(np.asarray(init_mu).reshape(-1)[i])
            for j in range(i + 1):
                prm.W[i * (i + 1) // 2 + j] = 0.5 * (W[i, j] + W[j, i])
                prm.P0inv[i * (i + 1) // 2 + j] = P0inv[i, j]
        prm.logdetQp = float(np.linalg.slogdet(Qp)[1])
        prm.logdetP0 = float(np.linalg.slogdet(P0)[1])
        prm.clip_lo, prm.clip_hi = (1.0, 0.0) if clip is None else (float(clip[0]), float(clip[1]))
        return prm

    def params(self, dt, init_mu, init_cov, lr=0.0, clip=None):
        """Fill the C parameter block (mfgm_sde_params)."""
        d = self.state_dim
        if not self.q_is_diagonal:
            raise ValueError("the closed-form HIP kernels need a diagonal diffusion matrix q for state_dim > 1: a full q runs on the "
                             "quadrature kernels (CVISitesSDEQuadrature, VariationalMarkovGP with d <= 3)")
        al, be = self.cubic(dt) if self.kind == 0 else (1.0, 0.0)
        prm = _lib.SdeParams()
        prm.kind, prm.dt = int(self.kind), float(dt)
        for i in range(d):
            prm.theta[i] = float(getattr(self, "theta", 0.0))
        P0 = np.asarray(init_cov, dtype=np.float64).reshape(d, d)
        P0inv = np.linalg.inv(P0)
        cP0 = np.linalg.cholesky(P0)
        for i in range(d):
            prm.alpha[i], prm.beta[i] = al, be
            prm.W[i] = 1.0 / (dt * self.q_diag[i])
            prm.sq_dtq[i] = math.sqrt(dt * self.q_diag[i])
            prm.mu0[i] = float(np.asarray(init_mu).reshape(-1)[i])
            for j in range(i + 1):
                prm.P0inv[i * (i + 1) // 2 + j] = P0inv[i, j]
                prm.cholP0[i * (i + 1) // 2 + j] = cP0[i, j]
        prm.logdetQp = sum(math.log(dt * v) for v in self.q_diag)
        prm.logdetP0 = float(np.linalg.slogdet(P0)[1])
        prm.lr = float(lr)
        if clip is None:
            prm.clip_lo, prm.clip_hi = 1.0, 0.0
        else:
            prm.clip_lo, prm.clip_hi = float(clip[0]), float(clip[1])
        return prm


class OrnsteinUhlenbeckSDE(SDE):
    """dx = -decay x dt + dB, spectral density q (sde.py:134-176)."""

    _param_names = ("decay",)

    def __init__(self, decay=1.0, q=None, trainable=False):
        super().__init__(torch.ones((1, 1), dtype=torch.float64) if q is None else q)
        self.decay = float(decay)
        self._trainable = {"decay": bool(trainable)}

    def cubic_jacobian(self, dt):
        return {"decay": (-dt, 0.0)}

    def drift_cubic_jacobian(self):
        return {"decay": (-1.0, 0.0)}

    def drift(self, x, t=None):
        return -self.decay * x

    def gradient_drift(self, x, t=None):
        return -self.decay * torch.ones_like(x)

    def cubic(self, dt):
        return 1.0 - dt * self.decay, 0.0

    def drift_cubic(self):
        """f(x) = af x - bf x^3."""
        return -self.decay, 0.0


class DoubleWellSDE(SDE):
    """dx = scale x (c - x^2) dt + dB (sde.py:179-224)."""

    _param_names = ("scale", "c")

    def __init__(self, q=None, scale_trainable=False, c_trainable=False, scale=4.0, c=1.0):
        super().__init__(torch.ones((1, 1), dtype=torch.float64) if q is None else q)
        self.scale, self.c = float(scale), float(c)
        self._trainable = {"scale": bool(scale_trainable), "c": bool(c_trainable)}

    def cubic_jacobian(self, dt):
        return {"scale": (dt * self.c, dt), "c": (dt * self.scale, 0.0)}

    def drift_cubic_jacobian(self):
        return {"scale": (self.c, 1.0), "c": (self.scale, 0.0)}

    def drift(self, x, t=None):
        return self.scale * x * (self.c - x * x)

    def gradient_drift(self, x, t=None):
        return self.scale * (self.c - 3.0 * x * x)

    def cubic(self, dt):
        return 1.0 + dt * self.scale * self.c, dt * self.scale

    def drift_cubic(self):
        return self.scale * self.c, self.scale


class _ThetaSDE(SDE):
    """Per-dimension non-polynomial drifts with one parameter theta (sde.py:227-356): Gaussian expectations by the reference's
    Gauss-Hermite rules inside the kernels (state_dim <= 4).  CVI-DP inference runs on the moment-array kernels (CVISitesSDE); the VDP
    model and prior learning on the tensor-product quadrature kernels (VariationalMarkovGPQuadrature, CVISitesSDEQuadrature; d <= 3)."""

    def quad_theta(self):
        return [self.theta, 0.0], 0

    def quad_param_jacobian(self):
        return {n: [1.0, 0.0] for n in self.trainable_variables}

    _param_names = ("theta",)

    def __init__(self, theta, q=None, trainable=False):
        super().__init__(torch.ones((1, 1), dtype=torch.float64) if q is None else q)
        if self.state_dim > 4:
            raise ValueError("the quadrature drifts are built for state_dim <= 4")
        self.theta = float(theta)
        self._trainable = {"theta": bool(trainable)}

    def cubic(self, dt):
        raise NotImplementedError(f"{type(self).__name__} is not a cubic drift")

    drift_cubic = cubic


class BenesSDE(_ThetaSDE):
    """dx = theta tanh(x) dt + dB (sde.py:227-268)."""
    kind = 1
    quad_kind = 13

    def __init__(self, theta=1.0, q=None, trainable=False):
        super().__init__(theta, q, trainable)

    def drift(self, x, t=None):
        return self.theta * torch.tanh(x)

    def gradient_drift(self, x, t=None):
        return self.theta * (1.0 - torch.tanh(x) ** 2)


class SineDiffusionSDE(_ThetaSDE):
    """dx = sin(x - theta) dt + dB (sde.py:271-312)."""
    kind = 2
    quad_kind = 14

    def __init__(self, theta=0.0, q=None, trainable=False):
        super().__init__(theta, q, trainable)

    def drift(self, x, t=None):
        return torch.sin(x - self.theta)

    def gradient_drift(self, x, t=None):
        return torch.cos(x - self.theta)


class SqrtDiffusionSDE(_ThetaSDE):
    """dx = sqrt(theta |x|) dt + dB (sde.py:315-356)."""
    kind = 3
    quad_kind = 15

    def __init__(self, theta=1.0, q=None, trainable=False):
        super().__init__(theta, q, trainable)

    def drift(self, x, t=None):
        return torch.sqrt(self.theta * torch.abs(x))

    def gradient_drift(self, x, t=None):
        return 0.5 * torch.sign(x) * torch.sqrt(self.theta / torch.abs(x))


# ---- drifts that couple the state dimensions or are not given in closed form (sde.py:359-518) ----------------------------------------------
def _hermgauss_grid(H, D, device):
    """Tensor-product Gauss-Hermite nodes [H^D, D] and weights [H^D] of GPflow's mvnquad (weights include pi^{-D/2})."""
    import itertools
    gx, gw = np.polynomial.hermite.hermgauss(H)
    x = np.array(list(itertools.product(*(gx,) * D)))
    w = np.prod(np.array(list(itertools.product(*(gw,) * D))), 1) * np.pi ** (-0.5 * D)
    return torch.from_numpy(x).to(device), torch.from_numpy(w).to(device)


def mvnquad(func, means, chols, H, Dout=()):
    """E_{N(m, L L^T)} func(x) by the H^D-point tensor Gauss-Hermite rule (gpflow.quadrature.mvnquad as the reference calls it,
    sde.py:92-131, sde_utils.py:262-359) for means [..., D] and Cholesky factors [..., D, D]; func maps [H^D, ..., D] -> [H^D, ..., *Dout].
    Differentiable in (means, chols): the nodes are m + sqrt(2) L z."""
    D = means.shape[-1]
    z, w = _hermgauss_grid(H, D, means.device)
    X = means[None] + math.sqrt(2.0) * torch.einsum("...ij,hj->h...i", chols, z)
    fX = func(X)
    return (fX * w.reshape((-1,) + (1,) * (fX.dim() - 1))).sum(0)


class QuadratureSDE(SDE):
    """An SDE whose drift is a torch function of the whole state: E_q f and E_q df/dx by the reference's 10-point-per-dimension
    Gauss-Hermite rules (sde.py:92-131; `expected_gradient_drift` returns the full Jacobian [.., D, D], sde.py:484-518 for the coupled
    drifts).  Served by variational_cvi_sde.CVISitesSDEQuadrature: small models (the tensor grid has 10^D / 20^D points per step), the
    posterior refresh itself stays in the HIP sweeps."""

    kind = -1       # not a drift family of the HIP kernels

    def jacobian_drift(self, x, t=None):
        """d f_i / d x_j at x [..., D] -> [..., D, D]; default: reverse-mode through `drift`, one pass per output dimension."""
        with torch.enable_grad():
            xx = x.detach().requires_grad_(True)
            f = self.drift(xx, t)
            rows = [torch.autograd.grad(f[..., i].sum(), xx, retain_graph=True)[0] for i in range(self.state_dim)]
        return torch.stack(rows, dim=-2)

    gradient_drift = jacobian_drift

    def expected_drift(self, q_mean, q_chol):
        return mvnquad(lambda x: self.drift(x), q_mean, q_chol, 10, (self.state_dim,))

    def expected_gradient_drift(self, q_mean, q_chol):
        return mvnquad(lambda x: self.jacobian_drift(x), q_mean, q_chol, 10, (self.state_dim, self.state_dim))

    def cubic(self, dt):
        raise NotImplementedError("not a per-dimension cubic drift: use CVISitesSDEQuadrature")

    def params(self, *a, **k):
        raise NotImplementedError("not a per-dimension cubic drift: use CVISitesSDEQuadrature")


class VanderPolOscillatorSDE(QuadratureSDE):
    """Van der Pol oscillator, state (x1, x2): f = tau (a (x1 - x1^3 / 3 - x2), x1 / a) (sde.py:432-482)."""

    _param_names = ("a", "tau")

    def __init__(self, a=1.0, tau=1.0, q=None, trainable=False):
        super().__init__(torch.eye(2, dtype=torch.float64) if q is None else q)
        if self.state_dim != 2:
            raise ValueError("the Van der Pol oscillator has two state dimensions")
        self.a, self.tau = float(a), float(tau)
        self._trainable = {"a": bool(trainable), "tau": bool(trainable)}

    quad_kind = 10

    def quad_theta(self):
        return [self.a, self.tau], 0

    def quad_param_jacobian(self):
        return {n: [1.0 if n == "a" else 0.0, 1.0 if n == "tau" else 0.0] for n in self.trainable_variables}

    def drift(self, x, t=None):
        x1, x2 = x[..., 0], x[..., 1]
        return self.tau * torch.stack([self.a * (x1 - x1 ** 3 / 3.0 - x2), x1 / self.a], dim=-1)

    def jacobian_drift(self, x, t=None):
        x1 = x[..., 0]
        one, zero = torch.ones_like(x1), torch.zeros_like(x1)
        return self.tau * torch.stack([torch.stack([self.a * (1.0 - x1 * x1), -self.a * one], dim=-1),
                                       torch.stack([one / self.a, zero], dim=-1)], dim=-2)

    gradient_drift = jacobian_drift


class MLPDrift(QuadratureSDE):
    """One-dimensional drift given by a 1 -> 3 -> 1 ReLU network (sde.py:359-429: two Dense layers, standard-normal initial weights, zero
    biases).  `weights` = (W1 [1, 3], b1 [3], W2 [3, 1], b2 [1]); default: drawn with torch's generator `seed`."""

    def __init__(self, q=None, weights=None, seed=0):
        super().__init__(torch.ones((1, 1), dtype=torch.float64) if q is None else q)
        if self.state_dim != 1:
            raise ValueError("MLPDrift is the reference's one-dimensional network drift")
        if weights is None:
            g = torch.Generator().manual_seed(int(seed))
            weights = (torch.randn((1, 3), generator=g, dtype=torch.float64), torch.zeros(3, dtype=torch.float64),
                       torch.randn((3, 1), generator=g, dtype=torch.float64), torch.zeros(1, dtype=torch.float64))
        self.weights = tuple(torch.as_tensor(w, dtype=torch.float64) for w in weights)
        self._trainable_weights = False

    quad_kind = 11

    def quad_theta(self):
        W1, b1, W2, b2 = self.weights
        return [float(v) for v in torch.cat([W1.reshape(-1), b1.reshape(-1), W2.reshape(-1), b2.reshape(-1)])], int(b1.numel())

    # the network's weights as ONE trainable vector "weights" (the reference trains MLP.trainable_variables, sde.py:375-381)
    @property
    def trainable_variables(self):
        return ["weights"] if self._trainable_weights else []

    def get(self, name):
        return np.array(self.quad_theta()[0])

    def assign(self, name, value):
        v = torch.as_tensor(np.asarray(value, dtype=np.float64))
        nh = int(self.weights[1].numel())
        self.weights = (v[:nh].reshape(1, nh).clone(), v[nh:2 * nh].clone(), v[2 * nh:3 * nh].reshape(nh, 1).clone(), v[3 * nh:3 * nh + 1].clone())

    def quad_param_jacobian(self):
        return {"weights": None}         # the kernels' parameter gradient IS the gradient of the weight vector

    def drift(self, x, t=None):
        W1, b1, W2, b2 = (w.to(x.device) for w in self.weights)
        h = torch.relu(x.reshape(-1, 1) @ W1 + b1)
        return (h @ W2 + b2).reshape(x.shape)

    def jacobian_drift(self, x, t=None):
        W1, b1, W2, b2 = (w.to(x.device) for w in self.weights)
        act = ((x.reshape(-1, 1) @ W1 + b1) > 0).to(x.dtype)
        return ((act * W1) @ W2).reshape(x.shape + (1,))

    gradient_drift = jacobian_drift
