"""
Host-side mirror of markovflow/block_tri_diag.py over the HIP kernels: same class and method names
(`SymmetricBlockTriDiagonal`, `LowerTriangularBlockTriDiagonal`; block_tri_diag.py:37-596), torch fp64 CUDA
tensors with the reference's ``[... outer_dim, inner_dim, inner_dim]`` shapes.
"""
import torch

from . import linalg
from ._lib import FULL, SYM, TRI, VEC
from .packed import Plan
from .state_space_model import _flat


class BlockTriDiagonal:
    """block_tri_diag.py:37-289."""

    _symmetric = False

    def __init__(self, diagonal, sub_diagonal=None, plan=None):
        d3, bs = _flat(diagonal, 3)
        self.batch_shape = bs
        self._diag = d3
        self.B, self.outer_dim, self.inner_dim = d3.shape[0], d3.shape[1], d3.shape[2]
        if d3.shape[-1] != d3.shape[-2]:
            raise ValueError("diagonal blocks must be square")
        self._sub = None
        if sub_diagonal is not None:
            s3, _ = _flat(sub_diagonal, 3)
            if tuple(s3.shape) != (self.B, self.outer_dim - 1, self.inner_dim, self.inner_dim):
                raise ValueError(f"sub_diagonal has shape {tuple(sub_diagonal.shape)}, incompatible with the diagonal")
            if self.outer_dim > 1:
                self._sub = s3
        self._plan = plan

    @property
    def plan(self):
        if self._plan is None:
            self._plan = Plan(self.B, self.outer_dim, self.inner_dim, device=self._diag.device)
        return self._plan

    def _unflat(self, x):
        return x.reshape(self.batch_shape + tuple(x.shape[1:]))

    @property
    def block_diagonal(self):
        return self._unflat(self._diag)

    @property
    def block_sub_diagonal(self):
        return None if self._sub is None else self._unflat(self._sub)

    @property
    def bandwidth(self):
        """Lower bandwidth excluding the main diagonal (block_tri_diag.py:117-127)."""
        return (2 if self._sub is not None else 1) * self.inner_dim - 1

    def to_dense(self):
        """block_tri_diag.py:150-158 (debug densification; small sizes only)."""
        T, d = self.outer_dim, self.inner_dim
        out = torch.zeros((self.B, T * d, T * d), dtype=self._diag.dtype, device=self._diag.device)
        dg = self._diag
        if self._symmetric:
            low = torch.tril(dg)
            dg = low + torch.tril(dg, -1).transpose(-1, -2)
        for k in range(T):
            out[:, k * d:(k + 1) * d, k * d:(k + 1) * d] = dg[:, k]
            if self._sub is not None and k < T - 1:
                out[:, (k + 1) * d:(k + 2) * d, k * d:(k + 1) * d] = self._sub[:, k]
                if self._symmetric:
                    out[:, k * d:(k + 1) * d, (k + 1) * d:(k + 2) * d] = self._sub[:, k].transpose(-1, -2)
        return self._unflat(out)

    def dense_mult(self, right, transpose_left=False):
        """
        M x (or M^T x) for right [..., outer_dim, inner_dim] (block_tri_diag.py:175-199 -> product_band_mat).
        Embarrassingly parallel in the outer dimension: one HIP kernel (mfgm_btd_matvec), one thread per output element.
        """
        x, _ = _flat(right, 2)
        if tuple(x.shape) != (self.B, self.outer_dim, self.inner_dim):
            raise ValueError("dense_mult: incompatible right-hand side")
        from . import _lib
        from .packed import _ptr, _stream
        lib = _lib.load()
        x = x.contiguous()
        out = torch.empty_like(x)
        dg = self._diag.contiguous()
        sub = self._sub.contiguous() if self._sub is not None else None
        _lib.check(lib.mfgm_btd_matvec(self.B, self.outer_dim, self.inner_dim, _ptr(dg), _ptr(sub), _ptr(x), _ptr(out),
                                       1 if self._symmetric else 0, 1 if transpose_left else 0, _stream()), "mfgm_btd_matvec")
        return self._unflat(out)


class LowerTriangularBlockTriDiagonal(BlockTriDiagonal):
    """block_tri_diag.py:291-381."""

    _symmetric = False

    def __init__(self, diagonal, sub_diagonal=None, plan=None, _packed=None):
        super().__init__(diagonal, sub_diagonal, plan)
        self._pk = _packed   # (L tri, G full) packed, when this factor came out of `cholesky`

    def abs_log_det(self):
        """block_tri_diag.py:353-366."""
        dg = torch.diagonal(self._diag, dim1=-2, dim2=-1)
        return self._unflat(0.5 * torch.log(dg * dg).sum(dim=(-1, -2)))

    def _gram(self):
        """K = L L^T as a SymmetricBlockTriDiagonal (block products, parallel in the outer dimension)."""
        Ld, Ls = self._diag, self._sub
        diag = Ld @ Ld.transpose(-1, -2)
        sub = None
        if Ls is not None:
            diag[:, 1:] += Ls @ Ls.transpose(-1, -2)
            sub = Ls @ Ld[:, :-1].transpose(-1, -2)
        return SymmetricBlockTriDiagonal(diag, sub, plan=self.plan)

    def block_diagonal_of_inverse(self):
        """Diagonal blocks of (L L^T)^{-1} (block_tri_diag.py:318-337 -> inverse_from_cholesky_band)."""
        pl = self.plan
        if self._pk is not None and self._pk[2] == pl.epoch:
            # this factor is the plan's most recent factorisation: its coarse levels are still in the workspace
            s = pl.selinv(self._pk[0], self._pk[1], None, want_sub=False)
            return self._unflat(pl.unpack(SYM, s["Sig"]))
        K = self._gram()
        return self._unflat(_flat(K.solve_and_marginals(None)[2], 3)[0])

    def solve(self, right, transpose_left=False):
        """
        L^{-1} x or L^{-T} x (block_tri_diag.py:339-351 -> solve_triang_mat) for an arbitrary lower block-bidiagonal factor: the
        plain substitution, parallelised exactly over time segments as an affine recurrence (mfgm_bidiag_solve); no Gram matrix is
        formed, so the error is that of a sequential triangular solve.
        """
        x, _ = _flat(right, 2)
        if tuple(x.shape) != (self.B, self.outer_dim, self.inner_dim):
            raise ValueError("solve: incompatible right-hand side")
        from . import _lib
        from .packed import _ptr, _stream
        if self.inner_dim > 32:
            raise ValueError("solve: inner_dim > 32 is not supported")
        lib = _lib.load()
        B, T, d = self.B, self.outer_dim, self.inner_dim
        x = x.contiguous()
        out = torch.empty_like(x)
        Ld = self._diag.contiguous()
        Ls = self._sub.contiguous() if self._sub is not None else (torch.zeros((B, max(T - 1, 1), d, d), dtype=x.dtype, device=x.device))
        scratch = torch.empty(max(lib.mfgm_bidiag_scratch_doubles(B, T, d), 1), dtype=torch.float64, device=x.device)
        _lib.check(lib.mfgm_bidiag_solve(B, T, d, _ptr(Ld), _ptr(Ls), _ptr(x), _ptr(out), 1 if transpose_left else 0, _ptr(scratch),
                                         _stream()), "mfgm_bidiag_solve")
        return self._unflat(out)


class SymmetricBlockTriDiagonal(BlockTriDiagonal):
    """block_tri_diag.py:384-549."""

    _symmetric = True

    def __add__(self, other):
        sub = self._sub
        if other._sub is not None:
            sub = other._sub if sub is None else sub + other._sub
        return SymmetricBlockTriDiagonal(self._unflat(self._diag + other._diag),
                                         None if sub is None else self._unflat(sub), plan=self._plan)

    @property
    def cholesky(self):
        """Block Cholesky (block_tri_diag.py:428-440 -> cholesky_band).  Raises ArithmeticError if not PD."""
        pl = self.plan
        D = pl.pack(SYM, self._diag)
        S = pl.pack(FULL, self._sub) if self._sub is not None else pl.zeros(FULL)
        f = pl.factor(D, S, None, want_logdet=False)
        pl.check_info()
        Ld = self._unflat(pl.unpack(TRI, f["L"]))
        Ls = None
        if self._sub is not None:
            Ls = self._unflat(pl.unpack(FULL, f["G"], self.outer_dim - 1))
        return LowerTriangularBlockTriDiagonal(Ld, Ls, plan=pl, _packed=(f["L"], f["G"], pl.epoch))

    def upper_diagonal_lower(self):
        """
        U D U^T factorisation (block_tri_diag.py:442-549): returns (U^T as LowerTriangularBlockTriDiagonal with identity
        diagonal, chol(D) as LowerTriangularBlockTriDiagonal).  With K^{-1} = this matrix, A_k = -U_k^T and D_k = Q_k^{-1}
        (D_0 = P_0^{-1}) are the parameters of the chain whose precision this is, so they come out of the same
        forward/backward sweeps as naturals_to_ssm_params instead of the reference's O(T^2) tf.while_loop.
        """
        if self._sub is None:
            raise ValueError("upper_diagonal_lower needs a sub-diagonal")
        from .ssm_gaussian_transformations import naturals_to_ssm_params_packed
        pl = self.plan
        td = pl.lincomb(pl.empty(SYM), -0.5, pl.pack(SYM, self._diag))
        ts = pl.lincomb(pl.empty(FULL), -1.0, pl.pack(FULL, self._sub))
        ssm = naturals_to_ssm_params_packed(pl, pl.zeros(VEC), td, ts)
        chols = torch.cat([ssm._cholP0[:, None], ssm._cholQ], dim=1)
        eye = torch.eye(self.inner_dim, dtype=chols.dtype, device=chols.device).expand(chols.shape)
        chol_d = linalg.cholesky(linalg.spd_inverse(chol=chols))
        u_s = -ssm._A
        identities = eye.contiguous()
        return (LowerTriangularBlockTriDiagonal(self._unflat(identities), self._unflat(u_s), plan=pl),
                LowerTriangularBlockTriDiagonal(self._unflat(chol_d), plan=pl))

    def solve_and_marginals(self, rhs=None):
        """
        Fused path used by the models: one factor + one selected inverse giving
        (log|L|, K^{-1} rhs, diag blocks of K^{-1}, sub-diagonal blocks of K^{-1}).
        """
        pl = self.plan
        D = pl.pack(SYM, self._diag)
        S = pl.pack(FULL, self._sub) if self._sub is not None else pl.zeros(FULL)
        r = None if rhs is None else pl.pack(VEC, _flat(rhs, 2)[0])
        f = pl.factor(D, S, r, want_logdet=True)
        s = pl.selinv(f["L"], f["G"], f["y"], want_sub=True)
        pl.check_info()
        x = None if rhs is None else self._unflat(pl.unpack(VEC, s["x"]))
        sub = self._unflat(pl.unpack(FULL, s["Sub"], self.outer_dim - 1)) if self.outer_dim > 1 else None
        return self._unflat(f["logdet"]), x, self._unflat(pl.unpack(SYM, s["Sig"])), sub
