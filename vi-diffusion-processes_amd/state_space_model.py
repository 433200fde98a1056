"""
Host-side mirror of markovflow/state_space_model.py (`StateSpaceModel`) over the HIP kernels.

Same constructor arguments, property and method names as the reference class
(state_space_model.py:35-664).  Tensors are torch fp64 CUDA tensors with the reference's shapes
``batch_shape + [T-1, d, d]`` etc.; all leading batch dimensions are flattened into the chain axis B of
the packed layout.  Every heavy quantity is computed once and cached (the reference recomputes the
precision, its Cholesky and the sparse inverse on every property access).
"""
import torch

from . import linalg
from ._lib import FULL, SYM, TRI, VEC
from .packed import Plan


def _flat(x, tail):
    """Flatten leading batch dims: returns ([B, *tail_shape], batch_shape)."""
    bs = tuple(x.shape[: x.dim() - tail])
    B = 1
    for v in bs:
        B *= int(v)
    return x.reshape((B,) + tuple(x.shape[x.dim() - tail:])).contiguous(), bs


class PackedSSM:
    """Packed storage of SSM parameters: A (FULL at node t: t -> t+1), off (VEC), chol (TRI)."""

    def __init__(self, plan, A, off, chol):
        self.plan, self.A, self.off, self.chol = plan, A, off, chol


class StateSpaceModel:
    """x0 ~ N(mu0, P0); x_{k+1} = A_k x_k + b_k + q_k, q_k ~ N(0, Q_k)  (state_space_model.py:35-130)."""

    def __init__(self, initial_mean, chol_initial_covariance, state_transitions, state_offsets,
                 chol_process_covariances, plan=None):
        A, bs = _flat(state_transitions, 3)
        if A.shape[1] == 0:
            # reference: tf.errors.InvalidArgumentError (tests/unit/test_state_space_model.py:58-60)
            raise ValueError("StateSpaceModel requires at least one transition")
        self.batch_shape = bs
        self._A = A
        self._b = _flat(state_offsets, 2)[0]
        self._cholQ = _flat(chol_process_covariances, 3)[0]
        self._mu0 = _flat(initial_mean, 1)[0]
        self._cholP0 = _flat(chol_initial_covariance, 2)[0]
        B, Tm1, d, _ = A.shape
        for name, t, shp in (("state_offsets", self._b, (B, Tm1, d)), ("chol_process_covariances", self._cholQ, (B, Tm1, d, d)),
                             ("initial_mean", self._mu0, (B, d)), ("chol_initial_covariance", self._cholP0, (B, d, d))):
            if tuple(t.shape) != shp:
                raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {shp}")
        self.B, self.T, self.d = B, Tm1 + 1, d
        self.plan = plan if plan is not None else Plan(B, self.T, d, device=A.device)
        self._packed = None
        self._prec = None
        self._post = None

    def __getattr__(self, name):
        # natural-layout parameter tensors of a model created from packed arrays are unpacked on first use
        if name in ("_A", "_b", "_cholQ", "_mu0", "_cholP0") and self.__dict__.get("_lazy_natural"):
            pl, pk = self.plan, self._packed
            off = pl.unpack(VEC, pk.off)
            chol = pl.unpack(TRI, pk.chol)
            self.__dict__.update(_A=pl.unpack(FULL, pk.A, self.T - 1), _b=off[:, 1:].contiguous(), _mu0=off[:, 0].contiguous(),
                                 _cholQ=chol[:, 1:].contiguous(), _cholP0=chol[:, 0].contiguous(), _lazy_natural=False)
            return self.__dict__[name]
        raise AttributeError(name)

    # -- reference-named accessors -----------------------------------------------------------
    @property
    def state_dim(self):
        return self.d

    @property
    def num_transitions(self):
        return self.T - 1

    def _unflat(self, x):
        return x.reshape(self.batch_shape + tuple(x.shape[1:]))

    @property
    def state_transitions(self):
        return self._unflat(self._A)

    @property
    def state_offsets(self):
        return self._unflat(self._b)

    @property
    def cholesky_process_covariances(self):
        return self._unflat(self._cholQ)

    @property
    def initial_mean(self):
        return self._unflat(self._mu0)

    @property
    def cholesky_initial_covariance(self):
        return self._unflat(self._cholP0)

    @property
    def concatenated_state_offsets(self):
        return self._unflat(torch.cat([self._mu0[:, None, :], self._b], dim=1))

    @property
    def concatenated_cholesky_process_covariance(self):
        return self._unflat(torch.cat([self._cholP0[:, None], self._cholQ], dim=1))

    # -- packed parameter storage ---------------------------------------------------------------
    @property
    def packed(self):
        if self._packed is None:
            pl = self.plan
            off = torch.cat([self._mu0[:, None, :], self._b], dim=1)
            chol = torch.cat([self._cholP0[:, None], self._cholQ], dim=1)
            self._packed = PackedSSM(pl, pl.pack(FULL, self._A), pl.pack(VEC, off), pl.pack(TRI, chol))
        return self._packed

    def _precision_packed(self):
        """precision blocks + K^{-1} mu (packed) + sum log chol (state_space_model.py:431-483, 343-373)."""
        if self._prec is None:
            pk = self.packed
            self._prec = self.plan.ssm_to_naturals(pk.A, pk.off, pk.chol, precision=True, want_logdet=True)
        return self._prec

    def _posterior_packed(self):
        """factor + selected inverse of the precision: marginal means / covariances / cross covariances."""
        if self._post is None:
            pr = self._precision_packed()
            pl = self.plan
            f = pl.factor(pr["diag"], pr["sub"], pr["lin"], want_logdet=True)
            s = pl.selinv(f["L"], f["G"], f["y"], want_sub=True)
            pl.check_info()
            self._post = dict(f=f, s=s)
        return self._post

    # -- reference API ---------------------------------------------------------------------------
    @property
    def precision(self):
        """SymmetricBlockTriDiagonal of K^{-1} (state_space_model.py:431-483)."""
        from .block_tri_diag import SymmetricBlockTriDiagonal
        pr = self._precision_packed()
        pl = self.plan
        diag = self._unflat(pl.unpack(SYM, pr["diag"]))
        sub = self._unflat(pl.unpack(FULL, pr["sub"], self.T - 1))
        return SymmetricBlockTriDiagonal(diag, sub)

    @property
    def marginal_means(self):
        """state_space_model.py:232-251 (mu = K (K^{-1} mu), one forward + one backward sweep)."""
        return self._unflat(self.plan.unpack(VEC, self._posterior_packed()["s"]["x"]))

    @property
    def marginal_covariances(self):
        """state_space_model.py:254-262."""
        return self._unflat(self.plan.unpack(SYM, self._posterior_packed()["s"]["Sig"]))

    @property
    def marginals(self):
        return self.marginal_means, self.marginal_covariances

    def subsequent_covariances(self, marginal_covariances=None):
        """Cov(x_{k+1}, x_k) (state_space_model.py:326-341); read off the selected inverse."""
        return self._unflat(self.plan.unpack(FULL, self._posterior_packed()["s"]["Sub"], self.T - 1))

    def covariance_blocks(self):
        return self.marginal_covariances, self.subsequent_covariances()

    def log_det_precision(self):
        """-2 (log|chol P0| + sum log|chol Q_k|) (state_space_model.py:343-373)."""
        return self._unflat(-2.0 * self._precision_packed()["sumlogchol"])

    def kl_divergence(self, dist):
        """KL(self || dist) (state_space_model.py:528-593); shape batch_shape."""
        if (dist.B, dist.T, dist.d) != (self.B, self.T, self.d):
            raise ValueError("kl_divergence: incompatible state space models")
        q = self._posterior_packed()["s"]
        pp = dist._precision_packed()
        mup = dist._posterior_packed()["s"]["x"]
        if dist.plan is not self.plan and (dist.plan.R, dist.plan.P) != (self.plan.R, self.plan.P):
            raise ValueError("kl_divergence: the two models must share a partition plan")
        tr, mh = self.plan.kl_terms(q["Sig"], q["Sub"], q["x"], pp["diag"], pp["sub"], mup)
        dim = float(self.T * self.d)
        kl = 0.5 * (tr + mh - dim + 2.0 * pp["sumlogchol"] - 2.0 * self._precision_packed()["sumlogchol"])
        return self._unflat(kl)


def _ssm_sample(self, sample_shape, generator=None):
    """
    Sample trajectories (state_space_model.py:298-324): x = (A^{-1})^{-1} (m + chol eps), i.e. a solve against the unit
    lower block-bidiagonal A^{-1} (partitioned on the device).  Returns sample_shape + batch_shape + [T, d].
    """
    from .block_tri_diag import LowerTriangularBlockTriDiagonal
    if isinstance(sample_shape, int):
        sample_shape = (sample_shape,)
    sample_shape = tuple(sample_shape)
    S = 1
    for v in sample_shape:
        S *= int(v)
    B, T, d = self.B, self.T, self.d
    dev, dt = self._A.device, self._A.dtype
    if S == 0:
        return torch.zeros(sample_shape + self.batch_shape + (T, d), dtype=dt, device=dev)
    eps = torch.randn((S, B, T, d), dtype=dt, device=dev, generator=generator)
    chols = torch.cat([self._cholP0[:, None], self._cholQ], dim=1)
    off = torch.cat([self._mu0[:, None, :], self._b], dim=1)
    z = (chols[None] @ eps[..., None])[..., 0] + off[None]
    eye = torch.eye(d, dtype=dt, device=dev).expand(S * B, T, d, d).contiguous()
    negA = (-self._A)[None].expand(S, B, T - 1, d, d).reshape(S * B, T - 1, d, d).contiguous()
    x = LowerTriangularBlockTriDiagonal(eye, negA).solve(z.reshape(S * B, T, d))
    return x.reshape(sample_shape + self.batch_shape + (T, d))


def _ssm_log_pdf(self, states):
    """log p(x) = log p(x0) + sum_k log p(x_{k+1} | x_k) (state_space_model.py:485-526); shape sample_shape + batch_shape."""
    import math
    d, T = self.d, self.T
    x = states.reshape((-1, self.B, T, d))

    def mvn(xx, mean, chol):
        z = linalg.solve_lower(chol, xx - mean)
        logdet = torch.log(torch.abs(torch.diagonal(chol, dim1=-2, dim2=-1))).sum(-1)
        return -0.5 * (z * z).sum(-1) - logdet - 0.5 * d * math.log(2 * math.pi)

    first = mvn(x[:, :, 0], self._mu0[None], self._cholP0[None])
    cond = (self._A[None] @ x[:, :, :-1, :, None])[..., 0] + self._b[None]
    rest = mvn(x[:, :, 1:], cond, self._cholQ[None]).sum(-1)
    return (first + rest).reshape(tuple(states.shape[:-2 - len(self.batch_shape)]) + self.batch_shape)


StateSpaceModel.sample = _ssm_sample
StateSpaceModel.log_pdf = _ssm_log_pdf


def state_space_model_from_covariances(initial_mean, initial_covariance, state_transitions, state_offsets,
                                       process_covariances):
    """state_space_model.py:613-664 (per-block Cholesky factorisations through vidp_amd.linalg)."""
    def chol_or_zero(cov):
        mask = (cov == 0).all(dim=-1).all(dim=-1)
        eye = torch.eye(cov.shape[-1], dtype=cov.dtype, device=cov.device)
        fix = torch.where(mask[..., None, None], eye, torch.zeros_like(eye))
        c = linalg.cholesky(cov + fix)
        return torch.where(mask[..., None, None], torch.zeros_like(c), c)

    return StateSpaceModel(initial_mean, chol_or_zero(initial_covariance), state_transitions, state_offsets,
                           chol_or_zero(process_covariances))
