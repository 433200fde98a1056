"""
Helper launched by tests/test_gpu_wide.py::test_one_chain_two_processes under torch.distributed.run (2 ranks, gloo, one GPU):
each process owns half of one chain (vidp_amd.distributed.ChainShard with the real process-group all-reduce) and checks its
slice of the posterior against a whole-chain solve done locally.  Exit code 0 = parity.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd  # noqa: E402
from tests.helpers import assert_close, random_dominant_btd  # noqa: E402
from vidp_amd.distributed import ChainShard, init_from_env  # noqa: E402


def main():
    rank, world = init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    B, T, d, R0 = 1, 240, 16, 8
    rng = np.random.default_rng(71892305)          # same inputs on every rank
    diag, sub = random_dominant_btd(rng, (B,), T, d)
    r = rng.normal(size=(B, T, d))
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    whole = vidp_amd.Plan(B, T, d, R0=R0)
    f0 = whole.factor(whole.pack(vidp_amd.SYM, dev(diag)), whole.pack(vidp_amd.FULL, dev(sub)), whole.pack(vidp_amd.VEC, dev(r)))
    s0 = whole.selinv(f0["L"], f0["G"], f0["y"])
    plan = vidp_amd.Plan(B, T, d, R0=R0)
    sh = ChainShard(plan, rank, world)
    lo, hi = sh.node_lo, sh.node_hi
    Dn, Sn, rn = np.full_like(diag, np.nan), np.full((B, T, d, d), np.nan), np.full_like(r, np.nan)
    Dn[:, lo:hi], rn[:, lo:hi] = diag[:, lo:hi], r[:, lo:hi]
    Sn[:, max(lo - 1, 0):min(hi, T - 1)] = sub[:, max(lo - 1, 0):min(hi, T - 1)]
    f = sh.factor(dev(Dn).reshape(-1), dev(Sn).reshape(-1), dev(rn).reshape(-1))
    s = sh.selinv(f["L"], f["G"], f["y"])
    plan.check_info()
    host = lambda x: x.cpu().numpy()
    for kind, a, b in ((vidp_amd.SYM, s["Sig"], s0["Sig"]), (vidp_amd.VEC, s["x"], s0["x"]), (vidp_amd.TRI, f["L"], f0["L"])):
        assert_close(host(plan.unpack(kind, a))[:, lo:hi], host(whole.unpack(kind, b))[:, lo:hi], rtol=1e-9)
    assert_close(host(plan.unpack(vidp_amd.FULL, s["Sub"], T - 1))[:, max(lo - 1, 0):hi - 1],
                 host(whole.unpack(vidp_amd.FULL, s0["Sub"], T - 1))[:, max(lo - 1, 0):hi - 1], rtol=1e-9)
    np.testing.assert_allclose(host(f["logdet"]), host(f0["logdet"]), rtol=1e-12)
    dist.barrier()
    if rank == 0:
        print("chain shard parity ok", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
