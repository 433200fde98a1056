"""
Multi-GPU layer: trajectories (the leading batch axis of every tensor on the path) are independent, so they are
sharded contiguously over ranks and nothing per-time-step is ever communicated.  The only exchange is the sum that the
reference takes over its batch axis (`tf.reduce_sum`, kalman_filter.py:255; variational_cvi.py:402): one all-reduce of
the scalar ELBO (plus any per-step scalars / small gradient vectors) over RCCL ("nccl" backend on ROCm) or gloo on CPU.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous, balanced shard [lo, hi) of `total` trajectories for `rank` (first `total % world` ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        elif torch.cuda.is_available():
            # gloo rehearsal of the multi-rank path: ranks share the GPUs that exist (several ranks per card on a one-GPU box)
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend, **kw)
    return rank, world


def allreduce_sum_(t):
    """In-place sum over ranks (no-op without a process group): the ELBO / gradient reduction."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max_(t):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def sum_over_ranks(values):
    """Sum each of a short list of Python floats over the ranks in ONE all-reduce and hand the totals back as floats: the
    ELBO / metric sums that steer a trainer (every rank must take the same learning-rate and convergence decisions) and the
    hyper-parameter gradients [d/d hyper ...] before an optimiser step (docs/diffusion_processes/cvi_dp_trainer.py:207-235,
    vi_markov_gp_trainer.py:163-216 sum over the batch axis; here the batch is spread over processes).  Identity without a group."""
    values = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return values
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor(values, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def sum_tensor_over_ranks(t):
    """Sum a small device tensor over the ranks (one all-reduce, no host synchronisation); the tensor itself without a group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() != "nccl" and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            return c.to(t.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allgather_(t):
    """[world, *t.shape] tensor of every rank's `t` (the tensor itself, with a leading axis of 1, without a process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t.contiguous())
        return torch.stack(parts)
    return t[None]


class ChainShard:
    """
    One long chain over several GPUs (SURVEY 8e second row, config 5): the time axis is cut at segment boundaries of a COARSE level
    X of the partition (the highest level that still has at least 4 x `world` nodes per chain, else `world`); rank k owns a contiguous range of that
    level's nodes and, below it, the segments they stand for.  Every rank eliminates the interior of its range by itself (the
    reduces of the levels 0 .. X-1 on its own segments); the one exchange per factorisation is the sum of the level-X inputs --
    n_X x (3 d^2 + 2 d) doubles per chain, every entry written by one rank (32 x 800 doubles = 205 KB at config 5 on 8 ranks) -- after which the
    levels >= X are solved redundantly on every rank and the sweeps back down touch only the owned segments.  Besides that exchange
    only the scalar log-determinant / quadratic-form sums cross ranks.
    Wide plans (8 < d <= 32) only.  Arrays are addressed with global node indices: a rank needs the inputs of its own nodes
    [node_lo, node_hi) and the sub-diagonal block at node_lo - 1; its outputs are valid on its own nodes (the cross-covariances
    Sigma_{t+1,t} on [node_lo - 1, node_hi - 1): each is produced together with Sigma_{t+1}).

    `allreduce` defaults to the in-place sum over the torch.distributed group, `allgather` to its all-gather; tests on a single GPU
    inject ones that work over several shard objects living in the same process.
    """

    def __init__(self, plan, rank, world, allreduce=None, allgather=None):
        import ctypes
        from . import _lib
        self.plan, self.rank, self.world = plan, int(rank), int(world)
        lev = (ctypes.c_int * 4)()
        levels = []
        for l in range(plan.nlevels):
            _lib.check(plan.lib.mfgm_plan_level(plan.h, l, lev), "mfgm_plan_level")
            levels.append(tuple(lev))                       # (n, R, P, Lpad)
        cand = [l for l in range(1, plan.nlevels) if levels[l][0] >= self.world]
        if not cand:
            raise ValueError(f"no level of this plan has {world} nodes per chain to share between the ranks: use a smaller R0")
        # balance: a level with about `world` nodes deals them out 2 : 1 in the worst case (3 ranks on 8 nodes own 3 / 3 / 2, and a node of a
        # high level is thousands of level-0 nodes), so prefer the highest level with >= 4 nodes per rank -- the levels from X upwards, solved
        # on every rank, and the exchange (n_X blocks) stay tiny either way
        roomy = [l for l in cand if levels[l][0] >= 4 * self.world]
        self.level = max(roomy) if roomy else max(cand)
        n_x = levels[self.level][0]
        lo, hi = shard_bounds(n_x, self.rank, self.world)
        _lib.check(plan.lib.mfgm_plan_set_shard_level(plan.h, self.level, lo, hi), "mfgm_plan_set_shard_level (wide plans with >= 2 levels only)")
        off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(plan.lib.mfgm_plan_exchange_region(plan.h, ctypes.byref(off), ctypes.byref(cnt)), "mfgm_plan_exchange_region")
        self.exchange = plan.ws[off.value:off.value + cnt.value]
        # level-0 nodes of the range: one node of level l stands for R_{l-1} nodes of level l-1
        span = 1
        for l in range(self.level):
            span *= levels[l][1]
        self.node_lo, self.node_hi = lo * span, min(hi * span, plan.T)
        self.seg_lo, self.seg_hi = self.node_lo // levels[0][1], -(-self.node_hi // levels[0][1])
        self._allreduce = allreduce if allreduce is not None else allreduce_sum_
        self._allgather = allgather if allgather is not None else allgather_

    def allreduce(self, t):
        return self._allreduce(t)

    def allgather(self, t):
        return self._allgather(t)

    def factor(self, D, S, r=None, aD=1.0, aS=1.0, aR=1.0, want_logdet=True, want_quad=False, moments_only=False):
        """Block Cholesky of the sharded chain.  Returns dict(L, G, y, logdet, quad, form) with logdet / quad already summed over ranks.
        moments_only: the factor arrays are only handed to `selinv(..., form=f["form"])` (inverse form, see Plan.factor)."""
        from ._lib import FULL, TRI, VEC, check
        from .packed import _ptr, _stream
        pl = self.plan
        L, G = pl.empty(TRI), pl.empty(FULL)
        y = pl.empty(VEC) if r is not None else None
        pl.epoch += 1
        logdet = torch.empty(pl.B, dtype=torch.float64, device=pl.device) if want_logdet else None
        quad = torch.empty(pl.B, dtype=torch.float64, device=pl.device) if want_quad else None
        args = (_ptr(D), _ptr(S), _ptr(r), float(aD), float(aS), float(aR), _ptr(L), _ptr(G), _ptr(y), _ptr(logdet), _ptr(quad),
                _ptr(pl.ws), _ptr(pl.info), _stream())
        form = 1 if moments_only else 0
        check(pl.lib.mfgm_packed_factor_phase_form(pl.h, form, 0, *args), "mfgm_packed_factor_phase_form(0)")
        self._allreduce(self.exchange)
        check(pl.lib.mfgm_packed_factor_phase_form(pl.h, form, 1, *args), "mfgm_packed_factor_phase_form(1)")
        if logdet is not None:
            self._allreduce(logdet)
        if quad is not None:
            self._allreduce(quad)
        return dict(L=L, G=G, y=y, logdet=logdet, quad=quad, form=form)

    def selinv(self, L, G, y=None, want_sub=True, form=0):
        """Selected inverse on the owned nodes (no communication: the levels from the exchange level up are replicated)."""
        return self.plan.selinv(L, G, y, want_sub=want_sub, form=form)

    def sparse_factor(self, nat1, nat2, plin, pdiag, psub, out=None, packed=False):
        """Inverse-form factorisation of the sparse-CVI posterior of the shared chain straight from the sites (mfgm_sparse_factor_phase;
        sparse_variational_cvi.py:140-174): phase 0 on the owned segments, ONE all-reduce of the exchange region, phase 1.  The sites
        node_lo .. node_hi must be current (site node_hi is the right neighbour's: SparseCVIGaussianProcess passes it on after every
        update).  `logdet` is the PARTIAL sum over the owned nodes -- the model adds it to the other scalars of the ELBO and reduces
        them together."""
        from ._lib import FULL, TRI, VEC, check
        from .packed import _ptr, _stream
        pl = self.plan
        out = {} if out is None else out
        L = out.get("L") if out.get("L") is not None else pl.empty(TRI)
        G = out.get("G") if out.get("G") is not None else pl.empty(FULL)
        y = out.get("y") if out.get("y") is not None else pl.empty(VEC)
        pl.epoch += 1
        logdet = torch.empty(pl.B, dtype=torch.float64, device=pl.device)
        args = (_ptr(nat1), _ptr(nat2), _ptr(plin), _ptr(pdiag), _ptr(psub), _ptr(L), _ptr(G), _ptr(y), _ptr(logdet), None, _ptr(pl.ws),
                _ptr(pl.info), _stream())
        fn = pl.lib.mfgm_sparse_factor_q if packed else pl.lib.mfgm_sparse_factor_phase      # (nat2: the quadrant-packed site tensor)
        check(fn(pl.h, 0, *args), "sparse factor, phase 0")
        self._allreduce(self.exchange)
        check(fn(pl.h, 1, *args), "sparse factor, phase 1")
        return dict(L=L, G=G, y=y, logdet=logdet, quad=None, form=1)

    def left_marginal(self, Sig, x=None):
        """Marginal of the separator on the left of the owned range (node_lo - 1, owned by the left neighbour) from the replicated
        exchange level, written into the global-index arrays Sig / x (after `selinv`)."""
        from ._lib import check
        from .packed import _ptr, _stream
        check(self.plan.lib.mfgm_plan_shard_left_marginal(self.plan.h, _ptr(Sig), _ptr(x), _ptr(self.plan.ws), _stream()),
              "mfgm_plan_shard_left_marginal")
