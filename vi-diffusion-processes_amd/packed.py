"""
Device-resident packed tensors and the partition plan (thin wrapper over the C ABI).

PyTorch is used only as the owner of device memory and of the HIP stream.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import FULL, SYM, TRI, VEC


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def observation_period(time_index):
    """The common spacing of the observation indices on the grid when they are equally spaced (tensor or array [n] or [B, n], the same
    for every trajectory), else 0."""
    ti = torch.as_tensor(time_index)
    if ti.dim() > 1:
        # [B, n]: the spacing along the LAST axis, and only when every trajectory has the same grid (flattening would see a negative
        # step at each row boundary and report "no period" for every batched index tensor)
        ti = ti.reshape(-1, ti.shape[-1])
        if ti.shape[0] > 1 and not bool((ti == ti[:1]).all()):
            return 0
        ti = ti[0]
    if ti.numel() < 2:
        return 0
    diffs = (ti[1:] - ti[:-1]).cpu()
    p = int(diffs[0])
    return p if p > 0 and bool((diffs == p).all()) else 0


def aligned_segment_length(B, T, d, period):
    """Level-0 segment length for chains whose observations sit every `period` grid nodes: the automatic length ceil(B T / 65536)
    (csrc/mfgm_api_core.hip) moved up to the next multiple of the period, so that the 64 segments of a wavefront meet their observation
    nodes at the SAME steps -- the observation branch of the sweeps (site added on load, marginals written at the node) is then taken
    by whole wavefronts on period-aligned steps instead of by one or two lanes on almost every step.  Headline (period 50): 98 -> 100,
    KL backward sweep 0.40 -> 0.35 ms, forward 0.43 -> 0.41, step -1.5 ... -2.3 %; config 3: 49 -> 50, -3 %; lengths sharing only a
    factor 10 with the period (110, 120) get part of it, co-prime ones (99, 101) are the worst (tools/ragged_probe.sh).  0 (= automatic)
    when there is no common period, d > 8, or the multiple would cost more than a fifth of the lanes."""
    if period <= 1 or d > 8 or os.environ.get("MFGM_R0"):      # MFGM_R0: the partition is being forced from outside
        return 0
    base = max(8, -(-B * T // 65536))
    if base <= 8:
        return 0
    r = -(-base // period) * period
    return r if r <= 1.2 * base and r <= T else 0


class Plan:
    """Partition plan for B chains of T nodes with d x d blocks (mfgm_plan_create)."""

    def __init__(self, B, T, d, R0=0, Rup=0, device="cuda"):
        self.lib = _lib.load()
        self.B, self.T, self.d = int(B), int(T), int(d)
        self.wide = self.d > 8      # one wavefront per segment (MFMA tiles) instead of one lane per segment
        h = ctypes.c_void_p()
        _lib.check(self.lib.mfgm_plan_create(self.B, self.T, self.d, int(R0), int(Rup), ctypes.byref(h)),
                   f"mfgm_plan_create(B={B}, T={T}, d={d})")
        self.h = h
        desc = (ctypes.c_int * 6)()
        self.lib.mfgm_plan_describe(self.h, desc)
        self.nlevels, self.R, self.P, self.Lpad = desc[0], desc[1], desc[2], desc[3]
        self.device = torch.device(device)
        nbytes = self.lib.mfgm_plan_workspace_bytes(self.h)
        self.ws = torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=self.device)
        self.info = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.epoch = 0   # bumped by every factorisation (the coarse-level factors in `ws` belong to the latest one)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.mfgm_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- storage ------------------------------------------------------------------------------
    def empty(self, kind):
        n = self.lib.mfgm_packed_doubles(self.h, kind)
        return torch.empty(n, dtype=torch.float64, device=self.device)

    def zeros(self, kind):
        n = self.lib.mfgm_packed_doubles(self.h, kind)
        return torch.zeros(n, dtype=torch.float64, device=self.device)

    def pack(self, kind, natural, out=None):
        """natural: [B, n_nodes, ...] contiguous fp64 device tensor; n_nodes = T or T-1."""
        natural = natural.contiguous()
        assert natural.dtype == torch.float64 and natural.is_cuda
        assert natural.shape[0] == self.B, (natural.shape, self.B)
        n_nodes = natural.shape[1]
        given = out is not None
        out = self.zeros(kind) if out is None else out
        _lib.check(self.lib.mfgm_pack(self.h, kind, _ptr(natural), n_nodes, _ptr(out), _stream()), "mfgm_pack")
        if given:
            torch.autograd.graph.increment_version(out)      # written behind torch's back: caches keyed on ._version must notice
        return out

    def unpack(self, kind, packed, n_nodes=None):
        n_nodes = self.T if n_nodes is None else n_nodes
        shape = (self.B, n_nodes, self.d) if kind == VEC else (self.B, n_nodes, self.d, self.d)
        out = torch.empty(shape, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_unpack(self.h, kind, _ptr(packed), _ptr(out), n_nodes, _stream()), "mfgm_unpack")
        return out

    # -- sweeps -------------------------------------------------------------------------------
    def factor(self, D, S, r=None, aD=1.0, aS=1.0, aR=1.0, want_logdet=True, want_quad=False, out=None, store_G=True,
               moments_only=False):
        """Block Cholesky (+ forward substitution).  Returns dict(L, G, y, logdet, quad, form).  store_G=False (d <= 8) skips the
        L_{t+1,t} output; the selected inverse then has to be taken with selinv_mom(..., S=S, aS=aS).
        moments_only=True says the factor arrays will only be handed to `selinv(..., form=f["form"])`: plans with d > 8 then use the
        inverse form (mfgm_packed_factor_form, form 1), whose arrays are not Cholesky blocks."""
        out = {} if out is None else out
        form = 1 if (moments_only and self.wide and store_G and os.environ.get("MFGM_INVERSE_FORM", "1") != "0") else 0
        L = out.get("L") if out.get("L") is not None else self.empty(TRI)
        G = None
        if store_G:
            G = out.get("G") if out.get("G") is not None else self.empty(FULL)
        y = None
        if r is not None:
            y = out.get("y") if out.get("y") is not None else self.empty(VEC)
        self.epoch += 1
        logdet = torch.empty(self.B, dtype=torch.float64, device=self.device) if want_logdet else None
        quad = torch.empty(self.B, dtype=torch.float64, device=self.device) if want_quad else None
        _lib.check(self.lib.mfgm_packed_factor_form(self.h, form, _ptr(D), _ptr(S), _ptr(r), float(aD), float(aS), float(aR),
                                                    _ptr(L), _ptr(G), _ptr(y), _ptr(logdet), _ptr(quad), _ptr(self.ws),
                                                    _ptr(self.info), _stream()), "mfgm_packed_factor_form")
        return dict(L=L, G=G, y=y, logdet=logdet, quad=quad, form=form)

    def sparse_factor(self, nat1, nat2, plin, pdiag, psub, want_logdet=True, out=None, packed=False):
        """Inverse-form factorisation of the sparse-CVI posterior straight from the sites (mfgm_sparse_factor; plans with d > 8, one
        chain): the posterior naturals  prior + overlap-added sites  are formed by the level-0 passes while loading.
        packed: nat2 is the quadrant-packed tensor [M + 1, d (d + 1) + d^2] (mfgm_sparse_factor_q)."""
        out = {} if out is None else out
        L = out.get("L") if out.get("L") is not None else self.empty(TRI)
        G = out.get("G") if out.get("G") is not None else self.empty(FULL)
        y = out.get("y") if out.get("y") is not None else self.empty(VEC)
        self.epoch += 1
        logdet = torch.empty(self.B, dtype=torch.float64, device=self.device) if want_logdet else None
        if packed:
            _lib.check(self.lib.mfgm_sparse_factor_q(self.h, -1, _ptr(nat1), _ptr(nat2), _ptr(plin), _ptr(pdiag), _ptr(psub), _ptr(L), _ptr(G),
                                                     _ptr(y), _ptr(logdet), None, _ptr(self.ws), _ptr(self.info), _stream()),
                       "mfgm_sparse_factor_q")
        else:
            _lib.check(self.lib.mfgm_sparse_factor(self.h, _ptr(nat1), _ptr(nat2), _ptr(plin), _ptr(pdiag), _ptr(psub), _ptr(L), _ptr(G),
                                                   _ptr(y), _ptr(logdet), None, _ptr(self.ws), _ptr(self.info), _stream()), "mfgm_sparse_factor")
        return dict(L=L, G=G, y=y, logdet=logdet, quad=None, form=1)

    def selinv(self, L, G, y=None, want_sub=True, out=None, form=0):
        """Selected inverse (+ backward substitution).  Returns dict(Sig, Sub, x).  `form`: the form of the factor arrays (`factor`)."""
        out = {} if out is None else out
        Sig = out.get("Sig") if out.get("Sig") is not None else self.empty(SYM)
        Sub = None
        if want_sub:
            Sub = out.get("Sub") if out.get("Sub") is not None else self.empty(FULL)
        x = None
        if y is not None:
            x = out.get("x") if out.get("x") is not None else self.empty(VEC)
        _lib.check(self.lib.mfgm_packed_selinv_form(self.h, int(form), _ptr(L), _ptr(G), _ptr(y), _ptr(Sig), _ptr(Sub), _ptr(x),
                                                    _ptr(self.ws), _stream()), "mfgm_packed_selinv_form")
        return dict(Sig=Sig, Sub=Sub, x=x)

    # -- local kernels --------------------------------------------------------------------------
    def lincomb(self, out, a, x, b=0.0, y=None, c=0.0, z=None):
        """out = a*x + b*y + c*z on flat packed arrays (in place allowed)."""
        n = x.numel()
        assert out.numel() == n and (y is None or y.numel() == n) and (z is None or z.numel() == n)
        _lib.check(self.lib.mfgm_lincomb(n, _ptr(out), float(a), _ptr(x), float(b), _ptr(y), float(c), _ptr(z), _stream()),
                   "mfgm_lincomb")
        return out

    def congruence_scan(self, Phi, Q, out=None):
        """X_t = Phi_t X_{t-1} Phi_t^T + Q_t, X_{-1} = 0, on packed arrays (Phi FULL, Q SYM over all T nodes; d <= 8): packed SYM X."""
        if getattr(self, "_scan_ws", None) is None:
            self._scan_ws = torch.empty(max(int(self.lib.mfgm_congruence_scan_workspace_doubles(self.h)), 1), dtype=torch.float64, device=self.device)
        X = out if out is not None else self.empty(SYM)
        _lib.check(self.lib.mfgm_congruence_scan(self.h, _ptr(Phi), _ptr(Q), _ptr(X), _ptr(self._scan_ws), _stream()), "mfgm_congruence_scan")
        return X

    def band_of_sigma_dP_sigma(self, Sig, Sub, dPd, dPs):
        """Packed (X_tt SYM, X_{t+1,t} FULL) of X = Sigma dP Sigma from the packed band of Sigma (Sig SYM, Sub FULL) and of the symmetric
        block-tri-diagonal dP (dPd SYM, dPs FULL); d <= 8 (mfgm_band_sigma_dP_sigma)."""
        if getattr(self, "_band_ws", None) is None:
            self._band_ws = torch.empty(max(int(self.lib.mfgm_band_workspace_doubles(self.h)), 1), dtype=torch.float64, device=self.device)
        Xd, Xs = self.empty(SYM), self.zeros(FULL)
        _lib.check(self.lib.mfgm_band_sigma_dP_sigma(self.h, _ptr(Sig), _ptr(Sub), _ptr(dPd), _ptr(dPs), _ptr(Xd), _ptr(Xs), _ptr(self._band_ws),
                                                     _stream()), "mfgm_band_sigma_dP_sigma")
        return Xd, Xs

    def gather_nodes(self, kind, packed, node_ids, out=None):
        """node_ids: int64 device tensor of b*T + t.  Returns natural [n, d] or [n, d, d] (written into `out` when given)."""
        n = node_ids.numel()
        shape = (n, self.d) if kind == VEC else (n, self.d, self.d)
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=self.device)
        assert tuple(out.shape) == shape and out.is_contiguous()
        _lib.check(self.lib.mfgm_node_io(self.h, kind, _ptr(packed), None, _ptr(node_ids), n, _ptr(out), 0, 1.0, _stream()),
                   "mfgm_node_io(gather)")
        return out

    def scatter_nodes(self, kind, packed, node_ids, values, accumulate=False, scale=1.0, packed2=None):
        """packed (+= | =) scale * values at the listed nodes; with `packed2` the same increment goes to a second array."""
        values = values.contiguous()
        n = node_ids.numel()
        if not accumulate:
            assert scale == 1.0 and packed2 is None
        _lib.check(self.lib.mfgm_node_io(self.h, kind, _ptr(packed), _ptr(packed2), _ptr(node_ids), n, _ptr(values),
                                         2 if accumulate else 1, float(scale), _stream()), "mfgm_node_io(scatter)")
        return packed

    def gather_nodes_pair(self, packed_vec, packed_sym, node_ids, out_vec, out_sym):
        """gather_nodes of a VEC and a SYM array at the same nodes in one launch (d <= 8), into the given buffers."""
        if self.d > 8:
            self.gather_nodes(VEC, packed_vec, node_ids, out=out_vec)
            self.gather_nodes(SYM, packed_sym, node_ids, out=out_sym)
            return out_vec, out_sym
        _lib.check(self.lib.mfgm_node_io_pair(self.h, _ptr(packed_vec), _ptr(packed_sym), _ptr(node_ids), node_ids.numel(),
                                              _ptr(out_vec), _ptr(out_sym), 0, 1.0, _stream()), "mfgm_node_io_pair(gather)")
        return out_vec, out_sym

    def scatter_nodes_pair(self, packed_vec, packed_sym, node_ids, values_vec, values_sym, scale=1.0):
        """packed_vec += scale * values_vec and packed_sym += scale * values_sym at the listed nodes, in one launch (d <= 8)."""
        values_vec, values_sym = values_vec.contiguous(), values_sym.contiguous()
        if self.d > 8:
            self.scatter_nodes(VEC, packed_vec, node_ids, values_vec, accumulate=True, scale=scale)
            self.scatter_nodes(SYM, packed_sym, node_ids, values_sym, accumulate=True, scale=scale)
            return
        _lib.check(self.lib.mfgm_node_io_pair(self.h, _ptr(packed_vec), _ptr(packed_sym), _ptr(node_ids), node_ids.numel(),
                                              _ptr(values_vec), _ptr(values_sym), 2, float(scale), _stream()), "mfgm_node_io_pair(scatter)")

    def site_update_pair(self, packed_vec, packed_sym, node_ids, sites_vec, sites_sym, g_vec, g_sym, lr):
        """sites <- (1 - lr) sites + lr g (in place) and packed += new - old at the listed nodes, in one launch (d <= 8)."""
        for t in (sites_vec, sites_sym, g_vec, g_sym):
            assert t.is_contiguous()
        _lib.check(self.lib.mfgm_site_update_pair(self.h, _ptr(packed_vec), _ptr(packed_sym), _ptr(node_ids), node_ids.numel(),
                                                  _ptr(sites_vec), _ptr(sites_sym), _ptr(g_vec), _ptr(g_sym), float(lr), _stream()),
                   "mfgm_site_update_pair")

    def mvn_obs_ve(self, mu, Sig, node_ids, n_per, y, Sinv, cst, out_mu=None, out_cov=None):
        """Per-trajectory variational expectations of a multivariate Gaussian likelihood at the observation nodes, gathered from the
        packed marginals in the same launch (d <= 8).  y: [B * n_per, d] trajectory-major; returns ve [B]."""
        ve = torch.empty((self.B, (int(n_per) + 255) // 256), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_mvn_obs_ve(self.h, _ptr(mu), _ptr(Sig), _ptr(node_ids), int(n_per), _ptr(y), _ptr(Sinv), float(cst),
                                            _ptr(out_mu), _ptr(out_cov), _ptr(ve), _stream()), "mfgm_mvn_obs_ve")
        return ve.sum(-1)

    def ssm_to_naturals(self, A, off, chol, precision=False, want_logdet=False, out=None):
        """Packed SSM parameters -> naturals (or precision blocks).  Returns dict(lin, diag, sub, sumlogchol)."""
        out = {} if out is None else out
        lin = None
        if off is not None:
            lin = out.get("lin") if out.get("lin") is not None else self.empty(VEC)
        diag = out.get("diag") if out.get("diag") is not None else self.empty(SYM)
        sub = out.get("sub") if out.get("sub") is not None else self.empty(FULL)
        slc = torch.empty(self.B, dtype=torch.float64, device=self.device) if want_logdet else None
        cD, cS = (1.0, -1.0) if precision else (-0.5, 1.0)
        _lib.check(self.lib.mfgm_packed_ssm_to_naturals(self.h, _ptr(A), _ptr(off), _ptr(chol), cD, cS, _ptr(lin),
                                                        _ptr(diag), _ptr(sub), _ptr(slc), _ptr(self.ws), _stream()),
                   "mfgm_packed_ssm_to_naturals")
        return dict(lin=lin, diag=diag, sub=sub, sumlogchol=slc)

    def kl_terms(self, Sig, Sub, mu, Pd, Ps, mup, aD=1.0, aS=1.0):
        """Trace and Mahalanobis terms of KL(q||p) per chain."""
        tr = torch.empty(self.B, dtype=torch.float64, device=self.device)
        mh = torch.empty(self.B, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_packed_kl_terms(self.h, _ptr(Sig), _ptr(Sub), _ptr(mu), _ptr(Pd), _ptr(Ps), float(aD),
                                                 float(aS), _ptr(mup), _ptr(tr), _ptr(mh), _ptr(self.ws), _stream()),
                   "mfgm_packed_kl_terms")
        return tr, mh

    def sde_kl(self, prm, mu, Sig, Sub, mode=0, grads=None, theta_q=None, want_kl=True):
        """Closed-form KL[q || p_SDE] (mode 0), + gradient wrt eta (mode 1, into `grads`), or fused Girsanov update (mode 2)."""
        kl = torch.empty(self.B, dtype=torch.float64, device=self.device) if want_kl else None
        g = grads if grads is not None else (None, None, None)
        q = theta_q if theta_q is not None else (None, None, None)
        _lib.check(self.lib.mfgm_packed_sde_kl(self.h, int(mode), ctypes.byref(prm), _ptr(mu), _ptr(Sig), _ptr(Sub), _ptr(kl),
                                               _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(q[0]), _ptr(q[1]), _ptr(q[2]),
                                               _ptr(self.ws), _ptr(self.info), _stream()), "mfgm_packed_sde_kl")
        return kl

    def linearize_cubic(self, prm, mu, Sig, out=None):
        """SDE linearised on the path (mu, Sig) as packed SSM parameters (A, off, chol)."""
        A = out[0] if out else self.empty(FULL)
        off = out[1] if out else self.empty(VEC)
        chol = out[2] if out else self.empty(TRI)
        _lib.check(self.lib.mfgm_packed_linearize_cubic(self.h, ctypes.byref(prm), _ptr(mu), _ptr(Sig), _ptr(A), _ptr(off),
                                                        _ptr(chol), _stream()), "mfgm_packed_linearize_cubic")
        return A, off, chol

    def stationary_ssm(self, spec, time_deltas):
        """Stationary kernel -> packed SSM parameters (A, off, chol); time_deltas natural [B, T-1]."""
        A, off, chol = self.empty(FULL), self.empty(VEC), self.empty(TRI)
        td = time_deltas.contiguous() if time_deltas is not None else None
        _lib.check(self.lib.mfgm_packed_stationary_ssm(self.h, ctypes.byref(spec), _ptr(td), _ptr(A), _ptr(off), _ptr(chol),
                                                       _ptr(self.info), _stream()), "mfgm_packed_stationary_ssm")
        return A, off, chol

    def node_ids(self, time_index):
        """int64 device tensor b*T + t for every chain and every index in `time_index` ([n] or [B, n])."""
        ti = torch.as_tensor(time_index, dtype=torch.int64, device=self.device)
        if ti.dim() == 1:
            ti = ti.unsqueeze(0).expand(self.B, -1)
        base = torch.arange(self.B, dtype=torch.int64, device=self.device).unsqueeze(1) * self.T
        return (base + ti).reshape(-1).contiguous()

    def selinv_mom(self, L, G, y, want_sub=False, out=None, S=None, aS=1.0):
        """Selected inverse that also writes the moment array (mu, diag Sigma, diag Sigma_sub).  Returns dict(Sig, Sub, x, mom).
        With G = None the factorisation was made with store_G=False and (S, aS) are its sub-diagonal input and scale."""
        out = {} if out is None else out
        Sig = out.get("Sig") if out.get("Sig") is not None else self.empty(SYM)
        Sub = None
        if want_sub:
            Sub = out.get("Sub") if out.get("Sub") is not None else self.empty(FULL)
        x = out.get("x") if out.get("x") is not None else self.empty(VEC)
        mom = out.get("mom")
        if mom is None:
            mom = torch.empty(3 * x.numel(), dtype=torch.float64, device=self.device)
        if G is None:
            if want_sub or S is None:
                raise ValueError("selinv_mom without L_{t+1,t} needs (S, aS) and cannot return the full cross-covariance blocks")
            _lib.check(self.lib.mfgm_packed_selinv_mom_s(self.h, -1, _ptr(L), _ptr(S), float(aS), _ptr(y), _ptr(Sig), _ptr(x), _ptr(mom),
                                                         _ptr(self.ws), _stream()), "mfgm_packed_selinv_mom_s")
        else:
            _lib.check(self.lib.mfgm_packed_selinv_mom(self.h, -1, _ptr(L), _ptr(G), _ptr(y), _ptr(Sig), _ptr(Sub), _ptr(x), _ptr(mom),
                                                       _ptr(self.ws), _stream()), "mfgm_packed_selinv_mom")
        return dict(Sig=Sig, Sub=Sub, x=x, mom=mom)

    def unpack_moments(self, mom):
        """Natural [B, T, 3d] view (mu, diag Sigma_tt, diag Sigma_{t+1,t}) of a packed moment array."""
        out = torch.empty((self.B, self.T, 3 * self.d), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_unpack_moments(self.h, _ptr(mom), _ptr(out), _stream()), "mfgm_unpack_moments")
        return out

    def sde_lean(self, prm, mom, Sig=None, mode=0, theta_q=None):
        """Moment-array CVI-DP kernel: mode 0 -> per-chain KL partial (add log|L_q| - T d / 2), mode 3 -> theta_q update."""
        q = theta_q if theta_q is not None else (None, None, None)
        out = torch.empty(self.B, dtype=torch.float64, device=self.device) if mode == 0 else None
        _lib.check(self.lib.mfgm_packed_sde_lean(self.h, int(mode), ctypes.byref(prm), _ptr(mom), _ptr(Sig), _ptr(out), _ptr(q[0]),
                                                 _ptr(q[1]), _ptr(q[2]), _ptr(self.ws), _stream()), "mfgm_packed_sde_lean")
        return out

    def selinv_s(self, L, S, aS, y, out=None):
        """Selected inverse of a store_G=False factorisation: dict(Sig, x) (marginals only; (S, aS) as given to `factor`)."""
        out = {} if out is None else out
        Sig = out.get("Sig") if out.get("Sig") is not None else self.empty(SYM)
        x = out.get("x") if out.get("x") is not None else self.empty(VEC)
        _lib.check(self.lib.mfgm_packed_selinv_mom_s(self.h, -1, _ptr(L), _ptr(S), float(aS), _ptr(y), _ptr(Sig), _ptr(x), None,
                                                     _ptr(self.ws), _stream()), "mfgm_packed_selinv_mom_s")
        return dict(Sig=Sig, x=x)

    def selinv_kl(self, L, S, aS, y, prm, out=None):
        """Backward sweep of a store_G=False factorisation returning dict(Sig, x, klpart): marginals and the per-chain moment-array
        KL sum of `sde_lean(mode=0)` (add log|L_q| - T d / 2), without ever writing the moment array."""
        out = {} if out is None else out
        Sig = out.get("Sig") if out.get("Sig") is not None else self.empty(SYM)
        x = out.get("x") if out.get("x") is not None else self.empty(VEC)
        kl = torch.empty(self.B, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_packed_selinv_kl(self.h, -1, _ptr(L), _ptr(S), float(aS), _ptr(y), ctypes.byref(prm), _ptr(Sig), _ptr(x),
                                                  _ptr(kl), _ptr(self.ws), _stream()), "mfgm_packed_selinv_kl")
        return dict(Sig=Sig, x=x, klpart=kl)

    def selinv_girsanov(self, L, S, aS, y, prm, theta_q, out, only_level=-1):
        """Backward sweep of a store_G=False factorisation fused with the Girsanov-site update: `out` = (lin, diag, sub) receives
        (1 - lr) theta_q + lr theta~ (out of place; theta_q = (lin, diag, sub) with sub = S)."""
        _lib.check(self.lib.mfgm_packed_selinv_girsanov(self.h, int(only_level), _ptr(L), _ptr(S), float(aS), _ptr(y), ctypes.byref(prm),
                                                        _ptr(theta_q[0]), _ptr(theta_q[1]), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                                        _ptr(self.ws), _stream()), "mfgm_packed_selinv_girsanov")
        return out

    # -- CVI-DP sweeps on the structured ("cq") state (include/mfgm.h, csrc/mfgm_cq.h) ------------------------------------------------------
    def cq_pack(self, lin, diag, sub):
        """Dense packed naturals -> (dyn, (dmin, dmax), (smin, smax)): the diagonals and the range of the off-diagonal entries."""
        dyn = torch.empty(self.lib.mfgm_cq_dyn_doubles(self.h), dtype=torch.float64, device=self.device)
        rng = torch.empty((4, self.Lpad), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_cq_pack(self.h, _ptr(lin), _ptr(diag), _ptr(sub), _ptr(dyn), _ptr(rng), _stream()), "mfgm_cq_pack")
        lo, hi = rng[0::2].min(dim=1).values.tolist(), rng[1::2].max(dim=1).values.tolist()
        return dyn, (lo[0], hi[0]), (lo[1], hi[1])

    def cq_unpack(self, cq):
        """cq state -> dense packed (lin, diag, sub), p0_off included, observation sites not."""
        lin, diag, sub = self.empty(VEC), self.empty(SYM), self.empty(FULL)
        _lib.check(self.lib.mfgm_cq_unpack(self.h, ctypes.byref(cq.struct()), _ptr(lin), _ptr(diag), _ptr(sub), _stream()), "mfgm_cq_unpack")
        return lin, diag, sub

    def cq_slots(self, node_ids):
        """int32 slot array of the observation nodes, or None when two observations share a node."""
        slot = torch.empty(self.lib.mfgm_cq_slot_ints(self.h), dtype=torch.int32, device=self.device)
        dup = torch.zeros(1, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.mfgm_cq_slots(self.h, _ptr(node_ids), node_ids.numel(), _ptr(slot), _ptr(dup), _stream()), "mfgm_cq_slots")
        return None if int(dup.item()) else slot

    def cq_factor(self, cq, want_logdet=True, out=None, use_ahead=False, next_sites=None, side=None):
        """Block Cholesky + forward substitution of the posterior precision given as a cq state: dict(L, y, logdet).
        Pipelined across the steps of the CVI-DP loop (mfgm_cq_factor_pipelined): `use_ahead` says the previous pipelined call made the
        separator system of exactly this state (the level-0 reduce is skipped); `next_sites` = (site_lin, site_sym) of the state whose
        separator system is to be made now, next to this call's level-0 forward sweep: by the second wavefront of the same kernel
        (side=None) or as a kernel of its own on the torch stream `side`."""
        out = {} if out is None else out
        L = out.get("L") if out.get("L") is not None else self.empty(TRI)
        y = out.get("y") if out.get("y") is not None else self.empty(VEC)
        self.epoch += 1
        logdet = torch.empty(self.B, dtype=torch.float64, device=self.device) if want_logdet else None
        if not use_ahead and next_sites is None:
            _lib.check(self.lib.mfgm_cq_factor(self.h, ctypes.byref(cq.struct()), _ptr(L), _ptr(y), _ptr(logdet), None, _ptr(self.ws),
                                               _ptr(self.info), _stream()), "mfgm_cq_factor")
            return dict(L=L, y=y, logdet=logdet)
        nxt = None
        if next_sites is not None:
            nxt = cq.struct()
            nxt.site_lin, nxt.site_sym = next_sites[0].data_ptr(), next_sites[1].data_ptr()
        _lib.check(self.lib.mfgm_cq_factor_pipelined(self.h, ctypes.byref(cq.struct()), _ptr(L), _ptr(y), _ptr(logdet), None, _ptr(self.ws),
                                                     _ptr(self.info), 1 if use_ahead else 0, ctypes.byref(nxt) if nxt is not None else None,
                                                     ctypes.c_void_p(side.cuda_stream) if side is not None else None, _stream()),
                   "mfgm_cq_factor_pipelined")
        return dict(L=L, y=y, logdet=logdet)

    def cq_selinv_girsanov(self, cq, L, y, prm, dyn_out, only_level=-1):
        _lib.check(self.lib.mfgm_cq_selinv_girsanov(self.h, int(only_level), ctypes.byref(cq.struct()), _ptr(L), _ptr(y), ctypes.byref(prm),
                                                    _ptr(dyn_out), _ptr(self.ws), _stream()), "mfgm_cq_selinv_girsanov")
        return dyn_out

    def cq_selinv_kl(self, cq, L, y, prm, out=None, obs_mu=None, obs_cov=None, only_level=-1, want_marginals=True):
        """want_marginals=False: the marginal arrays are not written (Sig, x = None in the result); the KL sum and the marginals at
        the observation nodes are all the ELBO needs."""
        out = {} if out is None else out
        Sig = x = None
        if want_marginals:
            Sig = out.get("Sig") if out.get("Sig") is not None else self.empty(SYM)
            x = out.get("x") if out.get("x") is not None else self.empty(VEC)
        kl = torch.empty(self.B, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_cq_selinv_kl(self.h, int(only_level), ctypes.byref(cq.struct()), _ptr(L), _ptr(y), ctypes.byref(prm), _ptr(Sig),
                                              _ptr(x), _ptr(kl), _ptr(obs_mu), _ptr(obs_cov), _ptr(self.ws), _stream()), "mfgm_cq_selinv_kl")
        return dict(Sig=Sig, x=x, klpart=kl)

    def mvn_ve_compact(self, mu, cov, n_per, y, Sinv, cst, partials=False):
        """mvn_obs_ve on marginals already gathered in observation order; returns ve [B] (partials=True: the per-block sums
        [B, ceil(n_per / 256)], for cq_elbo)."""
        ve = torch.empty((self.B, (int(n_per) + 255) // 256), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_mvn_ve_compact(self.B, int(n_per), self.d, _ptr(mu), _ptr(cov), _ptr(y), _ptr(Sinv), float(cst), _ptr(ve),
                                                _stream()), "mfgm_mvn_ve_compact")
        return ve if partials else ve.sum(-1)

    def cq_elbo(self, ve_part, kl_part, logdet, c):
        """(elbo [B], total) = per-chain sum_j ve_part[b, j] - (kl_part[b] + logdet[b] + c) and its sum, in one launch (mfgm_cq_elbo)."""
        out = torch.empty(self.B + 1, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_cq_elbo(self.B, int(ve_part.shape[-1]), _ptr(ve_part), _ptr(kl_part), _ptr(logdet), float(c), _ptr(self.info),
                                         _ptr(out), ctypes.c_void_p(out.data_ptr() + 8 * self.B), _stream()), "mfgm_cq_elbo")
        return out[:self.B], out[self.B]

    def site_lerp_to(self, out1, x1, g1, out2, x2, g2, w):
        """out = x + w (g - x) on two flat arrays in one launch (mfgm_site_lerp_to); out may be x."""
        _lib.check(self.lib.mfgm_site_lerp_to(_ptr(out1), _ptr(x1), _ptr(g1), x1.numel(), _ptr(out2), _ptr(x2), _ptr(g2), x2.numel(), float(w),
                                              _stream()), "mfgm_site_lerp_to")

    def check_info(self):
        """Raise ArithmeticError if a pivot block was not positive definite (synchronises); the message names the first failing chain and
        node range (mfgm_plan_check_info: status 2 with the location the kernels left in the info word)."""
        out = (ctypes.c_int * 4)()
        rc = self.lib.mfgm_plan_check_info(self.h, _ptr(self.info), out, _stream())
        if rc == 2:
            self.info.zero_()
            where = (f": chain {out[0]}, a node in [{out[1]}, {out[2]})" + (f" (separator system of level {out[3]})" if out[3] > 0 else "")
                     if out[0] >= 0 else "")
            err = ArithmeticError("block-tri-diagonal matrix is not positive definite" + where)
            err.location = tuple(out) if out[0] >= 0 else None
            raise err
        _lib.check(rc, "mfgm_plan_check_info")


class CqState:
    """The structured posterior naturals of CVI-DP (mfgm_cq_state): dyn (+ a spare buffer the Girsanov sweep writes), the uniform
    off-diagonal values, the node-0 block and the observation sites.  Tensors are referenced here so that the pointers handed to the
    library stay alive."""

    def __init__(self, dyn, d_off, s_off, p0_off=None, slot=None, site_lin=None, site_sym=None):
        self.dyn, self.spare = dyn, None
        self.version = 0          # bumped by whoever moves dyn / d_off / s_off / p0_off (a separator system made ahead is keyed on it)
        self.d_off, self.s_off = float(d_off), float(s_off)
        self.p0_off, self.slot, self.site_lin, self.site_sym = p0_off, slot, site_lin, site_sym

    def struct(self):
        st = _lib.CqState()
        st.dyn, st.d_off, st.s_off = self.dyn.data_ptr(), self.d_off, self.s_off
        st.p0_off = self.p0_off.data_ptr() if self.p0_off is not None else None
        if self.slot is not None:
            st.slot, st.site_lin, st.site_sym = self.slot.data_ptr(), self.site_lin.data_ptr(), self.site_sym.data_ptr()
        return st
