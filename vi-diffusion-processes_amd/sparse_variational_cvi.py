"""
Host-side mirror of markovflow/models/sparse_variational_cvi.py `SparseCVIGaussianProcess`
(sparse_variational_cvi.py:38-292): CVI with sites on pairs of consecutive inducing states.  The overlap-add of the
[M+1, 2d, 2d] sites into the block-tri-diagonal natural parameters and the data -> site segment sums (a Python list of
M+1 reduce_sums in the reference, :199-213) are single index_add operations; the posterior refresh runs in the HIP sweeps.
"""
import os

import torch

from ._lib import FULL, SYM, VEC
from .conditionals import _conditional_statistics, conditional_statistics
from .posterior import ConditionalProcess
from .ssm_gaussian_transformations import naturals_to_ssm_params_packed
from .variational_cvi import back_project_nats


class SparseCVIGaussianProcess:
    """`shard=(rank, world)` (or `(rank, world, allreduce, allgather)` with stand-ins for the collectives, or a dict with these keys and
    optionally the partition `R0`, `Rup`): ONE chain shared between
    `world` processes along time (SURVEY 8e second row, config 5; distributed.ChainShard).  Process k owns the inducing states
    [node_lo, node_hi), the sites / intervals m in [node_lo, node_hi) (interval m lies between states m - 1 and m; the last process also
    takes interval M) and the data points inside them.  Per step three small collectives cross processes: the site on the right edge
    (one [2d + 4d^2] block per neighbour pair, all-gathered after `update_sites`), the exchange level of the factorisation
    (n_X (3d^2 + 2d) doubles) and the four scalars of the ELBO.  Every process is handed the same (time_points, observations) and keeps
    its own slice; the arrays over inducing states keep global indices and are valid on the owned range.  State dimension > 8 only."""

    def __init__(self, kernel, inducing_points, likelihood, mean_function=None, learning_rate=0.1, shard=None):
        self._kernel, self._likelihood = kernel, likelihood
        self.learning_rate = learning_rate
        self.inducing_inputs = inducing_points
        M, sd = inducing_points.shape[-1], kernel.state_dim
        dev, dt = inducing_points.device, torch.float64
        self._version = 0          # bumped by every site update: keys the cached posterior marginals
        # Wide path (state dimension > 8): the RESIDENT form of nat2 is the quadrant-packed tensor _nat2q [M + 1, d (d + 1) + d^2]
        # (include/mfgm.h: the lower triangle of each symmetric [2d, 2d] site, 528 instead of 1 024 doubles at d = 16) that the site
        # update and the factor passes work on; `nat2` materialises the reference's [M + 1, 2d, 2d] tensor on demand and takes
        # assignments / in-place edits back (VIDP_PACKED_SITES=0: the dense tensor stays the resident one)
        self._packed = sd > 8 and inducing_points.dim() == 1 and os.environ.get("VIDP_PACKED_SITES", "1") != "0"
        self._nat2q, self._nat2, self._nat2_seen = None, None, None
        self.nat1 = torch.zeros((M + 1, 2 * sd), dtype=dt, device=dev)
        if self._packed:
            self._nat2q = torch.zeros((M + 1, sd * (sd + 1) + sd * sd), dtype=dt, device=dev)
        else:
            self.nat2 = torch.zeros((M + 1, 2 * sd, 2 * sd), dtype=dt, device=dev)
        self._dist_p = None
        self._shard = None
        if shard is not None:
            from .distributed import ChainShard
            from .packed import Plan
            if sd <= 8 or inducing_points.dim() != 1:
                raise ValueError("a chain is shared between processes on the wide path only (state dimension > 8, one chain)")
            if not isinstance(shard, dict):
                shard = dict(zip(("rank", "world", "allreduce", "allgather"), shard))
            rank, world = int(shard["rank"]), int(shard["world"])
            # a plan of its own: the prior's plan keeps serving whole-chain calls (dist_p, dist_q of gathered sites)
            self._shard = ChainShard(Plan(1, M, sd, R0=shard.get("R0", 0), Rup=shard.get("Rup", 0), device=dev), rank, world,
                                     shard.get("allreduce"), shard.get("allgather"))
            self._m_lo = self._shard.node_lo
            self._m_hi = self._shard.node_hi + (1 if rank == world - 1 else 0)

    # the sites are public tensors, as the reference's `sites.nat1 / nat2`: assigning them, or editing them in place through torch, moves
    # the key of every cached posterior quantity (the library's own in-place updates bump `_version`)
    @property
    def nat1(self):
        return self._nat1

    @nat1.setter
    def nat1(self, value):
        self._nat1 = value
        self._version += 1

    def _pack_index(self):
        """Flat indices into a [2d, 2d] site of the entries the packed form keeps, in its order: upper-left lower triangle, lower-left
        block, lower-right lower triangle."""
        idx = getattr(self, "_pidx", None)
        if idx is None:
            d = self._kernel.state_dim
            d2 = 2 * d
            i, j = torch.tril_indices(d, d)
            r, c = torch.meshgrid(torch.arange(d), torch.arange(d), indexing="ij")
            idx = torch.cat([i * d2 + j, ((d + r) * d2 + c).reshape(-1), (d + i) * d2 + d + j]).to(self._nat1.device)
            self._pidx = idx
        return idx

    def _sync_sites(self):
        """Packed mode: take back a dense `nat2` that was edited in place since it was handed out."""
        if self._packed and self._nat2 is not None and self._nat2._version != self._nat2_seen:
            self.nat2 = self._nat2

    @property
    def nat2(self):
        if self._packed:
            self._sync_sites()
            if self._nat2 is None:
                d2 = self._nat1.shape[-1]
                flat = torch.zeros((self._nat2q.shape[0], d2 * d2), dtype=torch.float64, device=self._nat2q.device)
                flat[:, self._pack_index()] = self._nat2q
                low = flat.view(-1, d2, d2)
                self._nat2 = low + low.transpose(-1, -2) - torch.diag_embed(torch.diagonal(low, dim1=-2, dim2=-1))
                self._nat2_seen = self._nat2._version
        return self._nat2

    @nat2.setter
    def nat2(self, value):
        self._nat2 = value
        self._version += 1
        if self._packed:
            self._nat2_seen = value._version
            self._nat2q = value.reshape(value.shape[0], -1)[:, self._pack_index()].contiguous()

    def _key(self):
        self._sync_sites()
        n2 = self._nat2q if self._packed else self._nat2
        return (self._version, id(self._nat1), self._nat1._version, id(n2), n2._version)

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
        """Prior naturals + overlap-added site naturals (sparse_variational_cvi.py:140-174) as a StateSpaceModel."""
        p = self.dist_p
        if self._shard is not None:
            self._gather_sites()
        lin, diag, sub = (x.clone() for x in self._theta())      # _theta() reuses its buffers
        q = naturals_to_ssm_params_packed(p.plan, lin, diag, sub)
        q.batch_shape = p.batch_shape
        return q

    @property
    def posterior(self):
        return ConditionalProcess(self.dist_q, self._kernel, self.inducing_inputs)

    def local_objective_and_gradients(self, Fmu, Fvar, Y):
        obj = self._likelihood.variational_expectations(Fmu, Fvar, Y).sum()
        return obj, self._likelihood.ve_gradients_expectation(Fmu, Fvar, Y)

    # ---- fused route (csrc/mfgm_sparse.h): sorted data points, one chain ---------------------------------------------------------------
    def _data(self, input_data):
        """Per-data-set constants of the fused route: CSR offsets of the data points per inter-inducing interval, w_i = H P_i, c_i = H T_i H^T
        (conditionals.py:207-256: functions of the time points and the kernel only), the kernel's initial state; or None when the
        route does not apply (unsorted or batched time points)."""
        import ctypes
        import weakref
        from . import _lib
        time_points, observations = input_data
        c = getattr(self, "_data_cache", None)
        if c is not None and c["ref"]() is time_points and c["ver"] == time_points._version:
            return c["val"]
        val = None
        z = self.inducing_inputs
        if (time_points.dim() == 1 and z.dim() == 1 and time_points.is_cuda and os.environ.get("VIDP_FUSED_SPARSE", "1") != "0"
                and (time_points.numel() < 2 or bool((time_points[1:] >= time_points[:-1]).all()))):
            M, d = int(z.shape[0]), self._kernel.state_dim
            m_lo, m_hi, own = 0, M + 1, slice(None)
            if self._shard is not None:
                # the data points of the owned intervals: a contiguous slice of the sorted time points
                m_lo, m_hi = self._m_lo, self._m_hi
                edges = torch.searchsorted(torch.searchsorted(z.contiguous(), time_points.contiguous()),
                                           torch.tensor([m_lo, m_hi], device=z.device), right=False)
                own = slice(int(edges[0]), int(edges[1]))
                time_points = time_points[own]
            N = int(time_points.shape[0])
            if N > 0:
                P, Tc, idx = _conditional_statistics(time_points, z, self._kernel)
                H = self._kernel.generate_emission_model(time_points[:1]).emission_matrix[0]      # [1, d], time-invariant
                w = (H @ P)[:, 0, :].contiguous()                                                  # [N, 2d]
                cc = (H @ Tc @ H.transpose(-1, -2))[:, 0, 0].contiguous()                          # [N]
            else:
                idx = torch.zeros(0, dtype=torch.int64, device=z.device)
                w, cc = (torch.zeros((1, 2 * d), dtype=torch.float64, device=z.device), torch.zeros(1, dtype=torch.float64, device=z.device))
            seg = torch.zeros(m_hi - m_lo + 1, dtype=torch.int32, device=z.device)
            seg[1:] = torch.cumsum(torch.bincount(idx - m_lo, minlength=m_hi - m_lo), 0).to(torch.int32)
            pm = self._kernel.initial_mean(()).to(z.device, torch.float64).contiguous()
            pc = self._kernel.initial_covariance_matrix().to(z.device, torch.float64).contiguous()
            sd = _lib.SparseData()
            sd.M, sd.d, sd.N, sd.m_lo, sd.m_hi = M, d, N, m_lo, m_hi
            sd.seg, sd.w, sd.c, sd.prior_mean, sd.prior_cov = seg.data_ptr(), w.data_ptr(), cc.data_ptr(), pm.data_ptr(), pc.data_ptr()
            val = dict(struct=sd, keep=(seg, w, cc, pm, pc), N=N, own=own)
        if val is None and self._shard is not None:
            raise ValueError("a shared chain needs sorted, un-batched time points on the device")
        self._data_cache = dict(ref=weakref.ref(input_data[0]), ver=input_data[0]._version, val=val)
        return val

    def _prior_natural(self):
        """Prior naturals in natural layout ([T, d], [T, d, d] x 2) + their packed form and sum log chol (cached: the prior is fixed)."""
        if getattr(self, "_pn", None) is None:
            p = self.dist_p
            pl = p.plan
            nat = pl.ssm_to_naturals(p.packed.A, p.packed.off, p.packed.chol, want_logdet=True)
            T, d = p.T, p.d
            if pl.d > 8:      # wide plans: the packed arrays are the natural ones
                lin, diag, sub = nat["lin"].view(T, d), nat["diag"].view(T, d, d), nat["sub"].view(T, d, d)
            else:
                lin, diag = pl.unpack(VEC, nat["lin"])[0], pl.unpack(SYM, nat["diag"])[0]
                sub = torch.zeros((T, d, d), dtype=torch.float64, device=pl.device)
                sub[:T - 1] = pl.unpack(FULL, nat["sub"], T - 1)[0]
            self._pn = dict(nat=nat, lin=lin.contiguous(), diag=diag.contiguous(), sub=sub.contiguous(), zeros=pl.zeros(VEC))
        return self._pn

    # the inverse form is used while the prior's precision blocks are conditioned better than this (tests/test_gpu_accuracy.py: at
    # 3.7e10, config 5's grid, both forms hold 1e-10 on the ELBO; at 1.1e13 the inverse form is 10-1000 x worse than the Cholesky form)
    INVERSE_FORM_MAX_COND = 1e11

    def _inverse_form(self):
        """The marginals come from the inverse-form sweeps (Plan.factor(moments_only=True), d > 8).  They lose ~10 eps cond(F_t) where
        the Cholesky form loses ~0.05 eps cond: on config 5's kernel and grid (rho = dz / lengthscale >= 0.05) the ELBO agrees with the
        oracle to 1e-10 in either form; on grids so fine that the prior precision is numerically singular (rho ~ 0.005, cond 1e15)
        neither form nor the NumPy oracle is meaningful in fp64.  So the form follows the conditioning of the prior's precision blocks
        (a sample of them, once per prior): above INVERSE_FORM_MAX_COND the Cholesky form is used, with a warning.
        VIDP_SPARSE_INVERSE_FORM=0 / 1 forces the Cholesky / inverse form."""
        if not self.dist_p.plan.wide:
            return False
        forced = os.environ.get("VIDP_SPARSE_INVERSE_FORM")
        if forced is not None:
            return forced != "0"
        pn = self._prior_natural()
        if "inverse_ok" not in pn:
            cond = self._prior_block_condition(pn["diag"])
            pn["inverse_ok"] = cond <= self.INVERSE_FORM_MAX_COND
            if not pn["inverse_ok"]:
                import warnings
                warnings.warn(f"SparseCVIGaussianProcess: the prior's precision blocks have condition number ~{cond:.1e} on this grid; "
                              "using the Cholesky-form sweeps (slower, ~200x more accurate) instead of the inverse form", RuntimeWarning)
        return pn["inverse_ok"]

    @staticmethod
    def _prior_block_condition(diag, samples=128):
        """Largest 2-norm condition number over a sample of the prior's diagonal precision blocks ([T, d, d], lower triangle valid): evenly
        spaced nodes plus both ends (on a uniform grid every interior block is the same).  Host side, once per prior."""
        T = diag.shape[0]
        idx = torch.unique(torch.cat([torch.linspace(0, T - 1, min(samples, T)).round().long(), torch.tensor([0, max(T - 2, 0), T - 1])]))
        blk = diag[idx.to(diag.device)].detach().cpu()
        low = torch.tril(blk)
        blk = low + torch.tril(blk, -1).transpose(-1, -2)
        ev = torch.linalg.eigvalsh(blk).abs()
        return float((ev.max(-1).values / ev.min(-1).values.clamp_min(1e-300)).max())

    def _fused_theta(self):
        import os
        return (self._inverse_form() and os.environ.get("VIDP_FUSED_THETA", "1") != "0" and self.nat1.is_contiguous()
                and (self._packed or self.nat2.is_contiguous()))

    def _theta(self):
        """Posterior naturals of the current sites, packed (lin, diag, sub): one pass over the sites (mfgm_sparse_theta)."""
        from . import _lib
        from .packed import _ptr, _stream
        p = self.dist_p
        pl, T, d = p.plan, p.T, p.d
        pn = self._prior_natural()
        wide = pl.d > 8
        b = self.__dict__.setdefault("_theta_bufs", {})
        if not b:
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=pl.device)
            b.update(lin=mk(T, d), diag=mk(T, d, d), sub=mk(T, d, d))
        _lib.check(pl.lib.mfgm_sparse_theta(T, d, _ptr(self.nat1), _ptr(self.nat2), _ptr(pn["lin"]), _ptr(pn["diag"]), _ptr(pn["sub"]),
                                            _ptr(b["lin"]), _ptr(b["diag"]), _ptr(b["sub"]), _stream()), "mfgm_sparse_theta")
        if wide:
            return b["lin"].view(-1), b["diag"].view(-1), b["sub"].view(-1)
        return (pl.pack(VEC, b["lin"][None]), pl.pack(SYM, b["diag"][None]),
                pl.pack(FULL, b["sub"][None, :T - 1].contiguous()) if T > 1 else pl.zeros(FULL))

    def _marginals(self):
        """Posterior marginals of the inducing states for the current sites: one factorisation + selected inverse, cached until the
        sites move.  dict(mu [T, d], Sig [T, d, d], Sub [T, d, d] natural; packed Sig / Sub / x; log|L|)."""
        m = getattr(self, "_marg", None)
        if m is not None and m["version"] == self._key():
            return m
        p = self.dist_p
        pl, T, d = p.plan, p.T, p.d
        bufs = self.__dict__.setdefault("_sweep_bufs", dict(f={}, s={}))
        if self._shard is not None:
            # one chain shared between processes: the same passes on the owned segments, one exchange (distributed.ChainShard)
            sh, pn = self._shard, self._prior_natural()
            pl = sh.plan
            f = sh.sparse_factor(self._nat1, self._nat2q if self._packed else self._nat2, pn["lin"], pn["diag"], pn["sub"], out=bufs["f"],
                                 packed=self._packed)
        elif self._fused_theta():
            # the level-0 passes of the factorisation form  prior + overlap-added sites  while loading: no posterior naturals in memory
            pn = self._prior_natural()
            f = pl.sparse_factor(self._nat1, self._nat2q if self._packed else self._nat2, pn["lin"], pn["diag"], pn["sub"], want_logdet=True,
                                 out=bufs["f"], packed=self._packed)
        else:
            lin, diag, sub = self._theta()
            f = pl.factor(diag, sub, lin, aD=-2.0, aS=-1.0, aR=1.0, want_logdet=True, out=bufs["f"], moments_only=self._inverse_form())
        bufs["f"].update(L=f["L"], G=f["G"], y=f["y"], form=f["form"])
        s = pl.selinv(f["L"], f["G"], f["y"], want_sub=True, out=bufs["s"], form=f["form"])
        bufs["s"].update(Sig=s["Sig"], Sub=s["Sub"], x=s["x"])
        if self._shard is not None:
            self._shard.left_marginal(s["Sig"], s["x"])      # the pair marginal of the first owned interval reaches one node to the left
        if pl.d > 8:
            mu, Sig, Sub = s["x"].view(T, d), s["Sig"].view(T, d, d), s["Sub"].view(T, d, d)
        else:
            mu, Sig = pl.unpack(VEC, s["x"])[0], pl.unpack(SYM, s["Sig"])[0]
            Sub = torch.zeros((T, d, d), dtype=torch.float64, device=pl.device)
            if T > 1:
                Sub[:T - 1] = pl.unpack(FULL, s["Sub"], T - 1)[0]
        self._marg = dict(version=self._key(), mu=mu, Sig=Sig, Sub=Sub, packed=s, logdetL=f["logdet"])
        return self._marg

    @staticmethod
    def _own_observations(data, observations):
        """observations[data["own"]] as ONE tensor object per (data set, observation tensor, version)."""
        c = data.get("y_own")
        if c is None or c[0] is not observations or c[1] != observations._version:
            c = data["y_own"] = (observations, observations._version, observations[data["own"]])
        return c[2]

    def _predict_f_data(self, data):
        """(fmu, fvar) [N, 1] at the data points from the cached marginals (mfgm_sparse_predict)."""
        import ctypes
        from . import _lib
        from .packed import _ptr, _stream
        c = getattr(self, "_pred_cache", None)
        if c is not None and c[0] == self._key() and c[1] is data:
            return c[2]          # the sites have not moved since these were computed (the ELBO of the previous iteration)
        m = self._marginals()
        pl = self.dist_p.plan if self._shard is None else self._shard.plan
        N = data["N"]
        out = torch.empty((2, max(N, 1)), dtype=torch.float64, device=pl.device)[:, :N]
        import os
        if pl.wide and (self._shard is not None or os.environ.get("VIDP_FUSED_SPARSE_KL", "1") != "0"):
            # the pass over the pair covariances also takes the trace / Mahalanobis terms of KL[q || p] (what classic_elbo asks for next)
            pn = self._prior_natural()
            kt = torch.empty(2, dtype=torch.float64, device=pl.device)
            _lib.check(pl.lib.mfgm_sparse_predict_kl(ctypes.byref(data["struct"]), _ptr(m["mu"]), _ptr(m["Sig"]), _ptr(m["Sub"]), _ptr(out[0]),
                                                     _ptr(out[1]), pl.h, _ptr(pn["nat"]["diag"]), _ptr(pn["nat"]["sub"]), -2.0, -1.0,
                                                     _ptr(self._prior_mean_packed()), _ptr(kt[0:1]), _ptr(kt[1:2]), _ptr(pl.ws), _stream()),
                       "mfgm_sparse_predict_kl")
            self._kl_cache = (self._key(), kt[0:1], kt[1:2])
        else:
            _lib.check(pl.lib.mfgm_sparse_predict(ctypes.byref(data["struct"]), _ptr(m["mu"]), _ptr(m["Sig"]), _ptr(m["Sub"]), _ptr(out[0]),
                                                  _ptr(out[1]), _stream()), "mfgm_sparse_predict")
        res = (out[0][:, None], out[1][:, None])
        self._pred_cache = (self._key(), data, res)
        return res

    def update_sites(self, input_data):
        """theta_m <- (1 - rho) theta_m + rho g_m, g_m = data gradients projected through p(f_k | v_m) (sparse_variational_cvi.py:176-221)."""
        data = self._data(input_data)
        if data is None:
            return self._update_sites_generic(input_data)
        import ctypes
        from . import _lib
        from .packed import _ptr, _stream
        time_points, observations = input_data
        fx_mus, fx_covs = self._predict_f_data(data)
        # the gradients alone: the reference's local_objective_and_gradients also returns the objective value, which update_sites drops
        # (:187-189) -- eight element-wise launches over the observations per step here; and the SAME tensor object for the owned
        # observations every step, so that a likelihood's per-observation-tensor caches hit (a fresh slice object never would)
        grads = self._likelihood.ve_gradients_expectation(fx_mus, fx_covs, self._own_observations(data, observations))
        g1, g2 = grads[0].reshape(-1).contiguous(), grads[1].reshape(-1).contiguous()
        pl = self.dist_p.plan
        self._sync_sites()
        if self._packed:
            _lib.check(pl.lib.mfgm_sparse_site_update_q(ctypes.byref(data["struct"]), _ptr(g1), _ptr(g2), float(self.learning_rate),
                                                        _ptr(self._nat1), _ptr(self._nat2q), _stream()), "mfgm_sparse_site_update_q")
            self._nat2 = None            # a dense copy handed out earlier is stale now
        else:
            _lib.check(pl.lib.mfgm_sparse_site_update(ctypes.byref(data["struct"]), _ptr(g1), _ptr(g2), float(self.learning_rate),
                                                      _ptr(self._nat1), _ptr(self._nat2), _stream()), "mfgm_sparse_site_update")
        self._version += 1
        if self._shard is not None:
            self._pass_edge_site()

    def _pass_edge_site(self):
        """A shared chain: the factorisation of the owned nodes reads the site node_hi, which the right neighbour owns and has just
        updated -- every process publishes its first owned site, one all-gather of [2d + 4d^2] doubles per process."""
        sh = self._shard
        lo, hi = sh.node_lo, sh.node_hi
        n2 = self._nat2q if self._packed else self._nat2
        mine = torch.cat([self._nat1[lo], n2[lo].reshape(-1)])
        every = sh.allgather(mine)
        if sh.rank + 1 < sh.world:
            d2 = self._nat1.shape[-1]
            self._nat1[hi].copy_(every[sh.rank + 1, :d2])
            n2[hi].copy_(every[sh.rank + 1, d2:].view(n2[hi].shape))
            if self._packed:
                self._nat2 = None
            self._version += 1

    def _gather_sites(self):
        """A shared chain: the sites of every process on every process (dist_q / posterior of the whole chain; not on the training path)."""
        sh = self._shard
        self._sync_sites()
        for t in (self._nat1, self._nat2q if self._packed else self._nat2):
            own = torch.zeros_like(t)
            own[self._m_lo:self._m_hi] = t[self._m_lo:self._m_hi]
            t.copy_(sh.allreduce(own))
        if self._packed:
            self._nat2 = None
        self._version += 1

    def _update_sites_generic(self, input_data):
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
        self._version += 1

    def classic_elbo(self, input_data):
        """sum_i E_q log p(y_i | f_i) - KL[q(s_Z) || p(s_Z)] (sparse_variational_cvi.py:270-292)."""
        time_points, observations = input_data
        data = self._data(input_data)
        if data is None:
            q = self.dist_q
            fx_mus, fx_covs = ConditionalProcess(q, self._kernel, self.inducing_inputs).predict_f(time_points)
            ve = self._likelihood.variational_expectations(fx_mus, fx_covs, observations).sum()
            return ve - q.kl_divergence(self.dist_p).sum()
        fx_mus, fx_covs = self._predict_f_data(data)
        y_own = self._own_observations(data, observations)
        ve_terms = getattr(self._likelihood, "variational_expectations_terms", None)
        kc = getattr(self, "_kl_cache", None)
        if ve_terms is not None and self._shard is None and kc is not None and kc[0] == self._key():
            # every piece of the bound is a device scalar by now: assemble  VE - KL  in one launch (a dozen scalar torch kernels otherwise),
            # NaN-poisoned when a pivot block of the factorisation was not positive definite
            import ctypes
            from . import _lib
            from .packed import _ptr, _stream
            m, pn, p = self._marginals(), self._prior_natural(), self.dist_p
            c, w, (t1, t2) = ve_terms(fx_mus, fx_covs, y_own)
            terms = torch.cat([t1, t2, kc[1].reshape(1), kc[2].reshape(1), pn["nat"]["sumlogchol"].reshape(1), m["logdetL"].reshape(1)])
            wts = (ctypes.c_double * 6)(w, w, -0.5, -0.5, -1.0, -1.0)
            out = torch.empty(1, dtype=torch.float64, device=terms.device)
            _lib.check(p.plan.lib.mfgm_combine_terms(6, 1, _ptr(terms), wts, c + 0.5 * float(p.T * p.d), None, 0.0, _ptr(p.plan.info), None,
                                                     _ptr(out), _stream()), "mfgm_combine_terms")
            return out[0]
        ve_sum = getattr(self._likelihood, "variational_expectations_sum", None)
        ve = ve_sum(fx_mus, fx_covs, y_own) if ve_sum is not None else self._likelihood.variational_expectations(fx_mus, fx_covs, y_own).sum()
        # KL[q || p] from the marginal blocks of q and the prior's naturals (state_space_model.py:528-593); the prior mean is zero
        m, pn = self._marginals(), self._prior_natural()
        p = self.dist_p
        pl = p.plan
        s = m["packed"]
        kc = getattr(self, "_kl_cache", None)
        if self._shard is not None:
            # partial sums over the owned data points, intervals and nodes: one all-reduce of four doubles
            kc = self._kl_cache
            tot = self._shard.allreduce(torch.cat([ve.reshape(1), kc[1], kc[2], m["logdetL"].reshape(1)]))
            kl = 0.5 * (tot[1] + tot[2] - float(p.T * p.d) + 2.0 * pn["nat"]["sumlogchol"].sum() + 2.0 * tot[3])
            return tot[0] - kl
        if kc is not None and kc[0] == self._key():
            tr, mh = kc[1], kc[2]          # taken together with the predictions (mfgm_sparse_predict_kl)
        else:
            tr, mh = pl.kl_terms(s["Sig"], s["Sub"], s["x"], pn["nat"]["diag"], pn["nat"]["sub"], self._prior_mean_packed(), aD=-2.0, aS=-1.0)
        kl = 0.5 * (tr + mh - float(p.T * p.d) + 2.0 * pn["nat"]["sumlogchol"] + 2.0 * m["logdetL"])
        return ve - kl.sum()

    def _prior_mean_packed(self):
        """Marginal means of the prior chain, packed (zero for the kernel priors)."""
        if getattr(self, "_pmu", None) is None:
            p = self.dist_p
            zero = bool((p._mu0 == 0).all() and (p._b == 0).all())
            self._pmu = p.plan.zeros(VEC) if zero else p._posterior_packed()["s"]["x"]
        return self._pmu

    def loss(self, input_data):
        return -self.classic_elbo(input_data)
