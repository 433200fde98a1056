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


class _StubSDE:
    """Two trainable drift parameters behind the prior-SDE interface the trainers use (sde.py: trainable_variables / get / assign)."""
    trainable_variables = ["a", "b"]

    def __init__(self):
        self.v = {"a": 0.8, "b": -0.4}

    def get(self, n):
        return self.v[n]

    def assign(self, n, v):
        self.v[n] = float(v)


class _StubModel:
    """The model interface both trainers' prior-learning loops touch, on trajectories whose objective is analytic:
    ELBO_b(theta) = -sum_i w_bi (theta_i - c_bi)^2.  Per-trajectory gradients differ, so ranks holding different shards take different
    Adam steps unless the gradients are summed first."""
    time_grid = grid = None
    state_dim = 1

    def __init__(self, c, w):
        self.c, self.w, self.prior_sde = np.asarray(c), np.asarray(w), _StubSDE()

    def _theta(self):
        return np.array([self.prior_sde.get(n) for n in self.prior_sde.trainable_variables])

    def classic_elbo(self):
        return float(-(self.w * (self._theta() - self.c) ** 2).sum())

    def elbo(self, mS=None):
        return self.classic_elbo()

    def grad_KL_wrt_prior_params(self):          # d(-ELBO) / d theta, the loss the reference's Adam minimises
        return list((2.0 * self.w * (self._theta() - self.c)).sum(0))

    def grad_VE_wrt_prior_params(self):
        return [0.0, 0.0]

    def grad_prior_sde_params(self):
        return self.grad_KL_wrt_prior_params()

    def _refresh_sde_params(self):
        pass

    _refresh_drift_params = _refresh_sde_params

    def _forward_packed(self):
        return None


def _stub_batch(B):
    rng = np.random.default_rng(11)
    return rng.normal(size=(B, 2)), rng.uniform(0.5, 2.0, size=(B, 2))


def _prior_learning(lo, hi, B):
    """Both trainers' optimize_prior_sde on trajectories [lo, hi) of the stub batch: (elbo history, parameter history) each."""
    sys.path.insert(0, ROOT)
    import vidp_amd  # noqa: F401
    from vidp_amd.trainers import CVISitesTrainer, VIMarkovGPTrainer
    c, w = _stub_batch(B)
    out = []
    for cls in (CVISitesTrainer, VIMarkovGPTrainer):
        tr = cls(_StubModel(c[lo:hi], w[lo:hi]), learn_prior_sde=True, prior_sde_lr=0.05, learning_max_itr=12, learning_tol=0.0)
        e, _, _ = tr.optimize_prior_sde()
        out.append((e, {k: list(v) for k, v in tr.prior_params.items()}))
    return out


def _worker_grads(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import vidp_amd  # noqa: F401
    from vidp_amd import distributed as D
    r, w = D.init_from_env(backend="gloo")
    lo, hi = D.shard_bounds(B, r, w)
    q.put((rank, _prior_learning(lo, hi, B)))
    torch.distributed.destroy_process_group()


def test_gloo_world2_trainers_allreduce_gradients():
    """cvi_dp_trainer.py:207-235 / vi_markov_gp_trainer.py:163-201 with the batch spread over 2 ranks (3 + 2 trajectories): the ELBO and the
    hyper-parameter gradients are summed over the ranks before Adam sees them, so both ranks take the SAME parameter steps, equal to a
    single process holding all 5 trajectories (and different from what either shard alone would do)."""
    B, world = 5, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_grads, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = _prior_learning(0, B, B)              # no process group here: the single-process run
    alone = _prior_learning(0, 3, B)              # rank 0's shard without the reduction
    for rank in range(world):
        for (e, prm), (e0, prm0), (_, prm1) in zip(res[rank], whole, alone):
            np.testing.assert_allclose(e, e0, rtol=1e-12)
            for k in prm0:
                assert len(prm[k]) == len(prm0[k]) == 13
                np.testing.assert_allclose(prm[k], prm0[k], rtol=1e-12)
            assert max(abs(a - b) for a, b in zip(prm0[0], prm1[0])) > 1e-3


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
