"""
World-size-2 gloo test (CPU) of the trajectory sharding and the ELBO all-reduce: each rank runs the oracle CVI model
on its shard of trajectories, the all-reduced sum must equal the single-process total.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _elbos(B, seed=3):
    """Per-trajectory ELBO of the oracle CVISitesSSM after one damped update (deterministic in the trajectory index)."""
    sys.path.insert(0, ROOT)
    from oracle import np_models, np_ssm
    from tests.helpers import random_ssm_params
    out = []
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        T, d = 25, 2
        ssm = np_ssm.StateSpaceModel(*random_ssm_params(rng, (), T, d))
        idx = np.arange(2, T, 4)
        y = rng.normal(size=(len(idx), d))
        m = np_models.CVISitesSSM(ssm, np.arange(T) * 0.1, idx, y, np_models.MultivariateGaussianLik(0.5 * np.eye(d)))
        m.update_data_sites(0.7)
        m.update_girsanov_sites(0.5)
        out.append(m.classic_elbo())
    return np.array(out)


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("vidp_dist", os.path.join(ROOT, "vi-diffusion-processes_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    r, w = D.init_from_env(backend="gloo")
    lo, hi = D.shard_bounds(B, r, w)
    local = _elbos(B)[lo:hi]
    t = torch.tensor([local.sum(), float(hi - lo)], dtype=torch.float64)
    D.allreduce_sum_(t)
    q.put((rank, lo, hi, t.tolist()))
    torch.distributed.destroy_process_group()


def test_shard_bounds():
    import importlib.util
    spec = importlib.util.spec_from_file_location("vidp_dist", os.path.join(ROOT, "vi-diffusion-processes_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for total in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            b = [D.shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_bounds(4, 2, 2)


def test_gloo_world2_elbo_allreduce():
    B, world = 5, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = _elbos(B).sum()
    for rank, lo, hi, (s, n) in res:
        assert n == B
        np.testing.assert_allclose(s, total, rtol=1e-12)
