"""
Oracle (test infrastructure, see oracle/__init__.py): non-linear SDE priors, Gauss-Hermite expectations,
linearisation and the Girsanov KL terms, restating markovflow/sde/{sde.py, sde_utils.py, drift.py}.

Third-party arithmetic absent from /root/reference: `gpflow.quadrature.mvnquad` (GPflow 2.2.1,
gpflow/quadrature/deprecated.py; call sites sde.py:109,128, sde_utils.py:247,354).  Its published algorithm is
restated in `mvnquad` below: tensor-product Gauss-Hermite nodes, X = sqrt(2) chol(S) xi + m, weights
pi^{-D/2} prod w, quadrature-point-major flattening.

PARITY UNPINNED beyond d = 1: the reference never tests these functions (tests/unit/test_sde.py covers only the
1-D OU process) and several of them only run for state_dim 1 (expected_gradient_drift returns the Jacobian
diagonal, sde.py:114-131; KL_q_p squares Cholesky factors element-wise, variational_cvi_sde.py:227-228).
For d > 1 this oracle defines A = diag(E[f'_i]) and Q = chol chol^T, and says so here.
"""
import itertools

import numpy as np

from . import np_transforms
from .np_ssm import StateSpaceModel, chol_solve

_T = lambda x: np.swapaxes(x, -1, -2)


def mvhermgauss(H, D):
    gx, gw = np.polynomial.hermite.hermgauss(H)
    x = np.array(list(itertools.product(*(gx,) * D)))
    w = np.prod(np.array(list(itertools.product(*(gw,) * D))), 1)
    return x, w


def mvnquad(func, means, covs, H, Din, Dout=()):
    """GPflow 2.2.1 mvnquad.  func maps [H^D * N, D] -> [H^D * N, *Dout]; returns [N, *Dout]."""
    xn, wn = mvhermgauss(H, Din)
    N = means.shape[0]
    chol = np.linalg.cholesky(covs)
    Xt = chol @ np.broadcast_to(xn.T, (N,) + xn.T.shape)          # N x D x H^D
    X = np.sqrt(2.0) * Xt + means[:, :, None]
    Xr = np.transpose(X, (2, 0, 1)).reshape(-1, Din)                # (H^D * N) x D
    fX = func(Xr).reshape((H ** Din, N) + tuple(Dout))
    wr = (wn * np.pi ** (-Din * 0.5)).reshape((-1,) + (1,) * (1 + len(Dout)))
    return np.sum(fX * wr, 0)


class SDE:
    """sde.py:24-131."""

    def __init__(self, q):
        self.q = np.atleast_2d(np.asarray(q, dtype=np.float64))
        self.state_dim = self.q.shape[0]

    def drift(self, x, t=None):
        raise NotImplementedError

    def gradient_drift(self, x, t=None):
        """d(sum_i f_i)/dx (sde.py:76-90): the Jacobian diagonal for per-dimension drifts."""
        raise NotImplementedError

    def diffusion(self, x, t=None):
        return np.ones_like(x[..., None]) * np.linalg.cholesky(self.q)

    def expected_drift(self, q_mean, q_covar):
        """sde.py:92-112 (H = 10)."""
        B, N, D = q_mean.shape
        val = mvnquad(lambda x: self.drift(x), q_mean.reshape(-1, D), q_covar.reshape(-1, D, D), 10, D, (D,))
        return val.reshape(B, N, D)

    def expected_gradient_drift(self, q_mean, q_covar):
        """sde.py:114-131 (H = 10); returns the expected Jacobian diagonal [B, N, D]."""
        B, N, D = q_mean.shape
        val = mvnquad(lambda x: self.gradient_drift(x), q_mean.reshape(-1, D), q_covar.reshape(-1, D, D), 10, D, (D,))
        return val.reshape(B, N, D)


def expected_drift_closed_form(sde, q_mean, q_covar):
    """
    E_q f and E_q f' of a per-dimension cubic drift f_i(x) = af x_i - bf x_i^3 under N(m, S), from the Gaussian moments of each
    dimension's own marginal (only diag S enters).  Equals SDE.expected_drift / expected_gradient_drift exactly (a 10-point
    Gauss-Hermite rule integrates the cubic and its derivative exactly: tests/test_oracle_sde.py pins the two at d <= 2); this
    form is what makes d = 6 tractable for the oracle (the tensor-product rule needs 10^6 points per time step there).
    """
    af, bf = drift_cubic(sde)
    v = np.einsum("...ii->...i", q_covar)
    Ef, Jf, _, _ = cubic_moments(af, bf, q_mean, v)
    return Ef, Jf


class OrnsteinUhlenbeckSDE(SDE):
    """sde.py:134-176: f(x) = -decay x."""

    def __init__(self, decay=1.0, q=np.ones((1, 1))):
        super().__init__(q)
        self.decay = float(decay)

    def drift(self, x, t=None):
        return -self.decay * x

    def gradient_drift(self, x, t=None):
        return -self.decay * np.ones_like(x)

    def cubic(self, dt):
        """u(x) = x + dt f(x) = alpha x - beta x^3."""
        return 1.0 - dt * self.decay, 0.0


class DoubleWellSDE(SDE):
    """sde.py:179-224: f(x) = scale x (c - x^2)."""

    def __init__(self, q=np.ones((1, 1)), scale=4.0, c=1.0):
        super().__init__(q)
        self.scale, self.c = float(scale), float(c)

    def drift(self, x, t=None):
        return self.scale * x * (self.c - np.square(x))

    def gradient_drift(self, x, t=None):
        return self.scale * (self.c - 3.0 * np.square(x))

    def cubic(self, dt):
        return 1.0 + dt * self.scale * self.c, dt * self.scale


class BenesSDE(SDE):
    """sde.py:227-268: f(x) = theta tanh(x)."""

    def __init__(self, theta=1.0, q=np.ones((1, 1))):
        super().__init__(q)
        self.theta = float(theta)

    def drift(self, x, t=None):
        return self.theta * np.tanh(x)

    def gradient_drift(self, x, t=None):
        return self.theta * (1.0 - np.tanh(x) ** 2)


class SineDiffusionSDE(SDE):
    """sde.py:271-312: f(x) = sin(x - theta)."""

    def __init__(self, theta=0.0, q=np.ones((1, 1))):
        super().__init__(q)
        self.theta = float(theta)

    def drift(self, x, t=None):
        return np.sin(x - self.theta)

    def gradient_drift(self, x, t=None):
        return np.cos(x - self.theta)


class SqrtDiffusionSDE(SDE):
    """sde.py:315-356: f(x) = sqrt(theta |x|)."""

    def __init__(self, theta=1.0, q=np.ones((1, 1))):
        super().__init__(q)
        self.theta = float(theta)

    def drift(self, x, t=None):
        return np.sqrt(self.theta * np.abs(x))

    def gradient_drift(self, x, t=None):
        return 0.5 * np.sign(x) * np.sqrt(self.theta / np.abs(x))


class VanderPolSDE(SDE):
    """sde.py:432-482: f = tau (a (x1 - x1^3 / 3 - x2), x1 / a); gradient_drift is the full Jacobian (batch_jacobian, sde.py:484-498)."""

    def __init__(self, a=1.0, tau=1.0, q=np.eye(2)):
        super().__init__(q)
        self.a, self.tau = float(a), float(tau)

    def drift(self, x, t=None):
        x1, x2 = x[..., 0], x[..., 1]
        return self.tau * np.stack([self.a * (x1 - x1 ** 3 / 3.0 - x2), x1 / self.a], axis=-1)

    def jacobian_drift(self, x, t=None):
        x1 = x[..., 0]
        one, zero = np.ones_like(x1), np.zeros_like(x1)
        return self.tau * np.stack([np.stack([self.a * (1.0 - x1 * x1), -self.a * one], axis=-1), np.stack([one / self.a, zero], axis=-1)], axis=-2)


def _vanderpol_hessian(self, x, t=None):
    """d^2 f_k / dx_i dx_j [..., k, i, j]: only f_1 = tau a (x1 - x1^3 / 3 - x2) is non-linear."""
    H = np.zeros(x.shape[:-1] + (2, 2, 2))
    H[..., 0, 0, 0] = -2.0 * self.tau * self.a * x[..., 0]
    return H


VanderPolSDE.hessian_drift = _vanderpol_hessian


class MLPDriftSDE(SDE):
    """sde.py:359-429: the one-dimensional 1 -> 3 -> 1 ReLU network drift, weights (W1 [1,3], b1 [3], W2 [3,1], b2 [1]) given."""

    def __init__(self, weights, q=np.ones((1, 1))):
        super().__init__(q)
        self.weights = [np.asarray(w, dtype=np.float64) for w in weights]

    def drift(self, x, t=None):
        W1, b1, W2, b2 = self.weights
        h = np.maximum(x.reshape(-1, 1) @ W1 + b1, 0.0)
        return (h @ W2 + b2).reshape(x.shape)

    def jacobian_drift(self, x, t=None):
        W1, b1, W2, b2 = self.weights
        act = ((x.reshape(-1, 1) @ W1 + b1) > 0).astype(np.float64)
        return ((act * W1) @ W2).reshape(x.shape + (1,))


def linear_drift_to_ssm(A, b, q, transition_times, initial_mean, initial_chol_covariance):
    """LinearDrift.to_ssm (drift.py:66-117): A_k = A dt + I, b_k = b dt, Q_k = q dt.  A [N,D,D], b [N,D], q [N,D,D]."""
    dts = (transition_times[1:] - transition_times[:-1])
    At = A * dts[:, None, None] + np.eye(A.shape[-1])
    bt = b * dts[:, None]
    cholQ = np.linalg.cholesky(q * dts[:, None, None])
    return StateSpaceModel(initial_mean, initial_chol_covariance, At, bt, cholQ)


def linearize_sde(sde, transition_times, path_mu, path_cov, init_mu, init_cov, closed_form=False, exact_q=False):
    """
    sde_utils.py:119-179.  path_mu [N, D], path_cov [N, D, D] (N = num transitions).
    A_i = E[f'] (as diag for D > 1), b_i = E[f] - A_i E[x]; then LinearDrift.to_ssm.
    closed_form: the two expectations from the cubic's Gaussian moments instead of the 10^D-point quadrature.
    exact_q: the process noise of the linearised prior is  chol_q chol_q^T = q  instead of the reference's  chol_q @ chol_q
    (sde_utils.py:173, no transpose).  The two agree for a diagonal q -- everything the reference ever runs --; for a full matrix the
    reference's product is not even symmetric, so a full q is only meaningful with this flag (the build under test uses q).
    """
    if exact_q:
        class _Sym:                                           # the same SDE with diffusion() @ diffusion() == q
            def __getattr__(self, name):
                return getattr(sde, name)

            def diffusion(self, x, t=None):
                w, v = np.linalg.eigh(np.asarray(sde.q, dtype=np.float64))
                return np.ones_like(x[..., None]) * ((v * np.sqrt(w)) @ v.T)
        return linearize_sde(_Sym(), transition_times, path_mu, path_cov, init_mu, init_cov, closed_form=closed_form)
    N, D = path_mu.shape
    if hasattr(sde, "jacobian_drift"):
        # drifts that couple the dimensions: E[f'] is the full Jacobian [N, D, D] (sde.py:500-518 with batch_jacobian, :484-498)
        E_f = sde.expected_drift(path_mu[None], path_cov[None])[0]
        A = mvnquad(lambda x: sde.jacobian_drift(x), path_mu, path_cov, 10, D, (D, D))
        b = E_f - (A @ path_mu[..., None])[..., 0]
        cq = sde.diffusion(path_mu, None)
        return linear_drift_to_ssm(A, b, cq @ cq, transition_times, init_mu, np.linalg.cholesky(init_cov))
    if closed_form:
        E_f, Adiag = expected_drift_closed_form(sde, path_mu, path_cov)
    else:
        E_f = sde.expected_drift(path_mu[None], path_cov[None])[0]
        Adiag = sde.expected_gradient_drift(path_mu[None], path_cov[None])[0]
    A = Adiag[:, :, None] * np.eye(D)
    b = E_f - (A @ path_mu[..., None])[..., 0]
    cq = sde.diffusion(path_mu, None)
    q = cq @ cq                                           # sde_utils.py:173 (chol_q @ chol_q, no transpose)
    return linear_drift_to_ssm(A, b, q, transition_times, init_mu, np.linalg.cholesky(init_cov))


def ssm_kl_along_gaussian_path(func_q, func_p, Qq, Qp, m, S, H=20):
    """
    SSM_KL_along_Gaussian_path (sde_utils.py:262-359):
      1/2 sum_t { E_{N(m_t,S_t)} |f_p(x) - f_q(x)|^2_{Qp^{-1}} + tr(Qp^{-1} * Qq) - D - logdet Qq + logdet Qp }
    m [N+1, D], S [N+1, D, D]; the quadrature runs over the first N marginals.
    """
    N, D = Qp.shape[0], Qp.shape[-1]
    Qp_inv = chol_solve(np.linalg.cholesky(Qp), np.broadcast_to(np.eye(D), Qp.shape))
    C = -(np.linalg.slogdet(Qq)[1] - np.linalg.slogdet(Qp)[1]) - D + np.sum(Qp_inv * Qq, axis=(-1, -2))

    def func(x):
        x = x.reshape(-1, N, D)
        diff = (func_p(x) - func_q(x))[..., None]
        return (_T(diff) @ Qp_inv[None] @ diff).reshape(-1)

    fn = mvnquad(func, m[:-1], S[:-1], H, D)
    return 0.5 * np.sum(fn + C)


def gauss_kl(m0, S0, m1, S1):
    """KL(N(m0,S0) || N(m1,S1))."""
    D = m0.shape[-1]
    S1inv = np.linalg.inv(S1)
    dm = m1 - m0
    return 0.5 * (np.trace(S1inv @ S0) + dm @ S1inv @ dm - D + np.linalg.slogdet(S1)[1] - np.linalg.slogdet(S0)[1])


def sde_ssm_kl_from_expectations(eta1, eta_d, eta_s, sde, dt, init_mu, init_cov, H=20):
    """The scalar differentiated by SDE_SSM_KL_with_grads_wrt_exp_params (sde_utils.py:473-547)."""
    A, b, cholP0, cholQ, mu0 = np_transforms.expectations_to_ssm_params(eta1, eta_d, eta_s)
    covar = eta_d - eta1[..., None] @ _T(eta1[..., None])
    Qq = cholQ @ _T(cholQ)
    N, D = b.shape
    Qp = np.broadcast_to(dt * sde.q, (N, D, D))
    f_q = lambda x: (A[None] @ x[..., None])[..., 0] + b[None]
    f_p = lambda x: x + dt * sde.drift(x)
    kl = ssm_kl_along_gaussian_path(f_q, f_p, Qq, Qp, eta1, covar, H)
    return kl + gauss_kl(mu0, cholP0 @ cholP0.T, init_mu, init_cov)


def sde_ssm_kl_grads_fd(eta1, eta_d, eta_s, sde, dt, init_mu, init_cov, eps=1e-6, H=20, richardson=False):
    """
    Central finite differences standing in for the reference's GradientTape (sde_utils.py:496-545).  Symmetric
    perturbations are used for eta_d so the result is the gradient with respect to the symmetric block.
    richardson: fourth-order quotient (4 D(eps / 2) - D(eps)) / 3 -- with eps ~ 1e-4 the result is good to ~1e-10 of the gradient's
    scale for a smooth drift (a ReLU drift keeps O(eps) kinks wherever a quadrature node crosses a unit's threshold).
    """
    f = lambda a, b_, c: sde_ssm_kl_from_expectations(a, b_, c, sde, dt, init_mu, init_cov, H)

    def quot(up, dn, h):
        d1 = (up(h) - dn(h)) / (2 * h)
        if not richardson:
            return d1
        return (4.0 * (up(h / 2) - dn(h / 2)) / h - d1) / 3.0
    g1, gd, gs = np.zeros_like(eta1), np.zeros_like(eta_d), np.zeros_like(eta_s)
    for idx in np.ndindex(eta1.shape):
        e = np.zeros_like(eta1); e[idx] = 1.0
        g1[idx] = quot(lambda h: f(eta1 + h * e, eta_d, eta_s), lambda h: f(eta1 - h * e, eta_d, eta_s), eps)
    for idx in np.ndindex(eta_d.shape):
        t, i, j = idx
        if j > i:
            continue
        e = np.zeros_like(eta_d); e[t, i, j] = 1.0; e[t, j, i] = 1.0
        v = quot(lambda h: f(eta1, eta_d + h * e, eta_s), lambda h: f(eta1, eta_d - h * e, eta_s), eps)
        # d/d(sym entry): for i != j both mirrored entries move, so the per-entry gradient is half of it
        gd[t, i, j] = gd[t, j, i] = v if i == j else 0.5 * v
    for idx in np.ndindex(eta_s.shape):
        e = np.zeros_like(eta_s); e[idx] = 1.0
        gs[idx] = quot(lambda h: f(eta1, eta_d, eta_s + h * e), lambda h: f(eta1, eta_d, eta_s - h * e), eps)
    return g1, gd, gs


# ---- closed forms for per-dimension cubic maps u(x) = alpha x - beta x^3 (OU: beta = 0; double-well) ----------
def cubic_moments(alpha, beta, m, v):
    """E u, E u', Var u for x ~ N(m, v) (element-wise), and their partial derivatives in (m, v)."""
    a = m * m + v
    ubar = alpha * m - beta * (m ** 3 + 3 * m * v)
    J = alpha - 3 * beta * a
    V = alpha ** 2 * v - 6 * alpha * beta * v * a + beta ** 2 * (9 * m ** 4 * v + 36 * m ** 2 * v ** 2 + 15 * v ** 3)
    d = dict(ubar_m=J, ubar_v=-3 * beta * m, J_m=-6 * beta * m, J_v=-3 * beta * np.ones_like(m),
             V_m=-12 * alpha * beta * m * v + beta ** 2 * (36 * m ** 3 * v + 72 * m * v ** 2),
             V_v=alpha ** 2 - 6 * alpha * beta * (m * m + 2 * v) + beta ** 2 * (9 * m ** 4 + 72 * m * m * v + 45 * v * v))
    return ubar, J, V, d


def sde_ssm_kl_closed_form(mu, Sig, Sub, alpha, beta, qdiag, dt, init_mu, init_cov, want_grads=True):
    """
    KL[q || p_SDE] and its gradient with respect to the expectation parameters, in closed form, for a q given by
    its marginal blocks (mu [T,D], Sig [T,D,D], Sub [T-1,D,D] = Cov(x_{t+1}, x_t)) and a prior with per-dimension
    cubic Euler map and DIAGONAL diffusion q.  Per transition (m,S,C,m',S'):
       KL_t = 1/2 { tr(W [V - J C^T - C J^T + S']) + |ubar - m'|^2_W - D - logdet(S' - C S^{-1} C^T) + logdet Qp }
    """
    T, D = mu.shape
    W = np.diag(1.0 / (dt * qdiag))
    logdetQp = np.sum(np.log(dt * qdiag))
    P0inv = np.linalg.inv(init_cov)
    kl = 0.5 * (np.trace(P0inv @ Sig[0]) + (mu[0] - init_mu) @ P0inv @ (mu[0] - init_mu) - D
                + np.linalg.slogdet(init_cov)[1] - np.linalg.slogdet(Sig[0])[1])
    Gm, GS, GC = np.zeros_like(mu), np.zeros_like(Sig), np.zeros_like(Sub)
    Gm[0] += P0inv @ (mu[0] - init_mu)
    GS[0] += 0.5 * (P0inv - np.linalg.inv(Sig[0]))
    for t in range(T - 1):
        m, S, C, mn, Sn = mu[t], Sig[t], Sub[t], mu[t + 1], Sig[t + 1]
        v = np.diag(S)
        ubar, J, V, d = cubic_moments(alpha, beta, m, v)
        A = np.linalg.solve(S, C.T).T
        Qq = Sn - A @ C.T
        P = np.linalg.inv(Qq)
        e = ubar - mn
        Wd = np.diag(W)
        kvec = np.einsum("ab,ab->b", W, C)
        kl += 0.5 * (np.sum(Wd * V) - 2.0 * np.sum(J * kvec) + np.trace(W @ Sn) + e @ W @ e - D
                     - np.linalg.slogdet(Qq)[1] + logdetQp)
        if want_grads:
            We = W @ e
            Gm[t + 1] += -We
            GS[t + 1] += 0.5 * (W - P)
            GC[t] += -W @ np.diag(J) + P @ A
            GS[t] += -0.5 * A.T @ P @ A + np.diag(0.5 * Wd * d["V_v"] - kvec * d["J_v"] + We * d["ubar_v"])
            Gm[t] += 0.5 * Wd * d["V_m"] - kvec * d["J_m"] + We * d["ubar_m"]
    if not want_grads:
        return kl
    g1 = Gm - 2.0 * (GS @ mu[..., None])[..., 0]
    g1[:-1] -= (_T(GC) @ mu[1:, :, None])[..., 0]
    g1[1:] -= (GC @ mu[:-1, :, None])[..., 0]
    return kl, (g1, GS, GC)


# ---- VDP (vi_sde.py) drift-difference energy ------------------------------------------------------------------
def squared_drift_difference_along_gaussian_path(sde, A, b, m, S, dt, H=20):
    """
    squared_drift_difference_along_Gaussian_path (sde_utils.py:182-249):
        1/2 dt sum_t E_{N(m_t, S_t)} (f_L(x) - f(x))^T q^{-1} (f_L(x) - f(x)),   f_L(x) = A_t x + b_t
    A [N, D, D], b [N, D] are the LINEAR DRIFT's parameters (the VDP model passes -A_model, b_model).
    """
    N, D = m.shape
    qinv = np.linalg.inv(sde.q)

    def func(x):
        x = x.reshape(-1, N, D)
        tmp = ((A[None] @ x[..., None])[..., 0] + b[None]) - sde.drift(x)
        return np.einsum("pni,ij,pnj->pn", tmp, qinv, tmp).reshape(-1)

    val = mvnquad(func, m, S, H, D)
    return 0.5 * np.sum(val) * dt


def squared_drift_difference_terms(sde, A, b, m, S, H=20):
    """The per-node terms of squared_drift_difference_along_gaussian_path WITHOUT the Riemann factor dt: [N]."""
    N, D = m.shape
    qinv = np.linalg.inv(sde.q)

    def func(x):
        x = x.reshape(-1, N, D)
        tmp = ((A[None] @ x[..., None])[..., 0] + b[None]) - sde.drift(x)
        return np.einsum("pni,ij,pnj->pn", tmp, qinv, tmp).reshape(-1)

    return 0.5 * mvnquad(func, m, S, H, D)


def e_sde_grads_fd(sde, A, b, m, S, eps=2e-4):
    """d E_sde / d (m, S) per node without the factor dt (what vi_sde.py:205-243 tapes and divides by dt), by fourth-order difference
    quotients of the reference's quadrature.  E_sde is a SUM of per-node terms, each a function of its own (m_t, S_t): one perturbed
    evaluation moves the same entry of every node at once."""
    N, D = m.shape
    f = lambda mm, SS: squared_drift_difference_terms(sde, A, b, mm, SS)

    def quot(up, dn):
        d1 = (up(eps) - dn(eps)) / (2 * eps)
        return (4.0 * (up(eps / 2) - dn(eps / 2)) / eps - d1) / 3.0
    dm, dS = np.zeros_like(m), np.zeros_like(S)
    for i in range(D):
        e = np.zeros(D); e[i] = 1.0
        dm[:, i] = quot(lambda h: f(m + h * e, S), lambda h: f(m - h * e, S))
        for j in range(i + 1):
            E = np.zeros((D, D)); E[i, j] = 1.0; E[j, i] = 1.0
            v = quot(lambda h: f(m, S + h * E), lambda h: f(m, S - h * E))
            dS[:, i, j] = dS[:, j, i] = v if i == j else 0.5 * v
    return dm, dS


def e_sde_grads_stein(sde, A, b, m, S, H=20):
    """d E_sde / d (m, S) per node without the factor dt, EXACT for a polynomial drift: with h(x) = 1/2 |f(x) - f_L(x)|^2_{q^-1},
    d/dm E h = E grad h and d/dS E h = 1/2 E hess h (Gaussian identities), both polynomials of degree <= 6 for the Van der Pol drift,
    which the 20-point rule integrates exactly.  (A, b) are the LINEAR DRIFT's parameters, as in squared_drift_difference_along_gaussian_path.)"""
    N, D = m.shape
    qinv = np.linalg.inv(sde.q)

    def parts(x):
        x = x.reshape(-1, N, D)
        r = sde.drift(x) - ((A[None] @ x[..., None])[..., 0] + b[None])            # f - f_L
        G = sde.jacobian_drift(x) - A[None]                                        # d r / d x
        qr = r @ qinv
        return x, G, qr

    def grad(x):
        x, G, qr = parts(x)
        return np.einsum("pnki,pnk->pni", G, qr).reshape(-1, D)

    def hess(x):
        x, G, qr = parts(x)
        Hf = sde.hessian_drift(x)
        return (np.einsum("pnki,kl,pnlj->pnij", G, qinv, G) + np.einsum("pnk,pnkij->pnij", qr, Hf)).reshape(-1, D, D)

    return mvnquad(grad, m, S, H, D, (D,)), 0.5 * mvnquad(hess, m, S, H, D, (D, D))


def drift_cubic(sde):
    """f_i(x) = af x - bf x^3 for the two per-dimension drifts on the path."""
    if isinstance(sde, OrnsteinUhlenbeckSDE):
        return -sde.decay, 0.0
    return sde.scale * sde.c, sde.scale


def e_sde_closed_form(af, bf, qdiag, A, b, m, S, dt, want_grads=True):
    """Closed form of the energy above for a per-dimension cubic drift and DIAGONAL q, with dE/dm [N,D] and the
    symmetric-convention dE/dS [N,D,D]."""
    N, D = m.shape
    w = 1.0 / qdiag
    v = np.einsum("nii->ni", S)
    Ef, Jf, Vf, d = cubic_moments(af, bf, m, v)
    El = (A @ m[..., None])[..., 0] + b
    LS = A @ S
    LSL = np.einsum("nik,nik->ni", LS, A)
    LSii = np.einsum("nii->ni", LS)
    r = El - Ef
    Tm = LSL - 2.0 * LSii * Jf + Vf + r * r
    E = 0.5 * dt * np.sum(w * Tm)
    if not want_grads:
        return E
    dm = np.zeros_like(m)
    dS = np.zeros_like(S)
    for i in range(D):
        l = A[:, i, :]                                        # [N, D]
        ei = np.zeros(D); ei[i] = 1.0
        gi_m = 2.0 * r[:, i, None] * (l - ei[None] * Jf[:, i, None])
        gi_m[:, i] += -2.0 * LSii[:, i] * d["J_m"][:, i] + d["V_m"][:, i]
        dm += 0.5 * dt * w[i] * gi_m
        gi_S = l[:, :, None] * l[:, None, :] - Jf[:, i, None, None] * (l[:, :, None] * ei[None, None, :] + ei[None, :, None] * l[:, None, :])
        gi_S[:, i, i] += -2.0 * LSii[:, i] * d["J_v"][:, i] + d["V_v"][:, i] - 2.0 * r[:, i] * d["ubar_v"][:, i]
        dS += 0.5 * dt * w[i] * gi_S
    return E, dm, dS
