import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vidp_amd as amd
from vidp_amd.distributed import ChainShard
from tests.helpers import random_dominant_btd
from tests.test_gpu_wide import _ThreadAllReduce
d, T, R0, world = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (16, 400, 10, 4)
rng = np.random.default_rng(71892305)
B = 2
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
host = lambda x: x.detach().cpu().numpy()
diag, sub = random_dominant_btd(rng, (B,), T, d)
r = rng.normal(size=(B, T, d))
whole = amd.Plan(B, T, d, R0=R0)
f0 = whole.factor(whole.pack(amd.SYM, dev(diag)), whole.pack(amd.FULL, dev(sub)), whole.pack(amd.VEC, dev(r)), want_logdet=True, want_quad=True)
Lref = host(whole.unpack(amd.TRI, f0["L"]))
yref = host(whole.unpack(amd.VEC, f0["y"]))
ld_node = np.log(np.diagonal(Lref, axis1=-2, axis2=-1)).sum(-1)      # [B, T]
print("ref logdet", host(f0["logdet"]), ld_node.sum(-1), "levels", whole.nlevels)
class NoReduce:
    def __call__(self, t): return t
group = _ThreadAllReduce(world)
res = [None] * world
def run(rank):
    plan = amd.Plan(B, T, d, R0=R0)
    # capture partial logdet before the all-reduce: wrap the allreduce
    calls = []
    inner = group.for_rank(rank)
    def ar(t):
        calls.append(t.clone())
        return inner(t)
    sh = ChainShard(plan, rank, world, allreduce=ar)
    f = sh.factor(dev(diag).reshape(-1), dev(sub).reshape(-1) if False else torch.cat([dev(sub), torch.zeros((B,1,d,d), dtype=torch.float64, device="cuda")], 1).reshape(-1), dev(r).reshape(-1), want_logdet=True, want_quad=True)
    Lr, yr = host(plan.unpack(amd.TRI, f["L"])), host(plan.unpack(amd.VEC, f["y"]))
    lo, hi = sh.node_lo, sh.node_hi
    errL = np.abs(Lr[:, lo:hi] - Lref[:, lo:hi]).max(axis=(0, 2, 3))
    erry = np.abs(yr[:, lo:hi] - yref[:, lo:hi]).max(axis=(0, 2))
    print(rank, "max L err per node (first 25):", np.array2string(errL[:25], precision=1), "y:", np.array2string(erry[:25], precision=1), flush=True)
    res[rank] = (sh.level, sh.node_lo, sh.node_hi, host(calls[1]), host(calls[2]))
ths = [threading.Thread(target=run, args=(k,)) for k in range(world)]
[t.start() for t in ths]; [t.join() for t in ths]
for rank, (lev, lo, hi, pl, pq) in enumerate(res):
    print(rank, "level", lev, "nodes", lo, hi, "partial logdet", pl, "expected", ld_node[:, lo:hi].sum(-1), "quad", pq, "exp", (yref[:, lo:hi] ** 2).sum((-1, -2)))
