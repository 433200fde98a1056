"""
Helper launched by tests/test_gpu_api.py::test_trainers_two_processes_share_the_batch under torch.distributed.run (2 ranks, gloo, one
GPU): 4 OU trajectories spread 2 + 2 over the processes, CVISitesTrainer.optimize() and VIMarkovGPTrainer prior learning with
learn_prior_sde (cvi_dp_trainer.py:138-250, vi_markov_gp_trainer.py:163-201).  Every process first runs the whole batch alone (no
process group yet: the reductions are identities), then its shard inside the group: ELBO histories and drift-parameter histories
must coincide -- the gradients and the ELBO are summed over the ranks before Adam and the convergence rules see them.
Exit code 0 = parity.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd  # noqa: E402,F401
from vidp_amd import sde as gsde  # noqa: E402
from vidp_amd.distributed import init_from_env, shard_bounds  # noqa: E402
from vidp_amd.likelihoods import MultivariateGaussian  # noqa: E402
from vidp_amd.trainers import CVISitesTrainer, VIMarkovGPTrainer  # noqa: E402
from vidp_amd.variational_cvi_sde import CVISitesSDE  # noqa: E402
from vidp_amd.vi_sde import VariationalMarkovGP  # noqa: E402


def data(B, T, dt, seed):
    rng = np.random.default_rng(seed)
    x = np.zeros((B, T))
    for k in range(1, T):
        x[:, k] = x[:, k - 1] - dt * 2.0 * x[:, k - 1] + np.sqrt(dt) * rng.normal(size=B)
    idx = np.arange(2, T - 1, 3)
    y = (x[:, idx] + 0.05 * rng.normal(size=(B, len(idx))))[..., None]
    test_idx = np.arange(3, T - 1, 9)
    return idx, y, test_idx, x[:, test_idx][..., None]


def run(lo, hi):
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = []
    T, dt = 240, 0.01
    grid = np.arange(T) * dt
    idx, y, tidx, ytest = data(4, T, dt, 5)
    lik = MultivariateGaussian(dev(0.05 * np.eye(1)))
    ou = gsde.OrnsteinUhlenbeckSDE(0.3, torch.eye(1, dtype=torch.float64), trainable=True)
    m = CVISitesSDE(ou, grid, (grid[idx], dev(y[lo:hi])), lik, prior_initial_state=(np.zeros(1), np.eye(1) / 0.6), stabilize_ssm=False)
    tr = CVISitesTrainer(m, test_data=(grid[tidx], dev(ytest[lo:hi])), girsanov_sites_lr=1.0, data_sites_lr=1.0, max_itr=2,
                         max_itr_sites_optim=3, learn_prior_sde=True, prior_sde_lr=0.2, learning_max_itr=6, learning_tol=1e-3)
    e, n, r, prm = tr.optimize()
    out.append((e, n, r, prm[0]))
    T, dt = 120, 0.01
    grid = np.arange(T) * dt
    idx, y, tidx, ytest = data(4, T, dt, 6)
    ou = gsde.OrnsteinUhlenbeckSDE(0.3, torch.eye(1, dtype=torch.float64), trainable=True)
    vm = VariationalMarkovGP((grid[idx], dev(y[lo:hi])), ou, grid, MultivariateGaussian(dev(0.3 * np.eye(1))),
                             prior_initial_state=(np.zeros(1), np.eye(1) / 0.6))
    tv = VIMarkovGPTrainer(vm, test_data=(grid[tidx], dev(ytest[lo:hi])), q_lr=0.05, x0_lr=0.05, max_itr=6, warmup_itr=2,
                           learn_prior_sde=True, prior_sde_lr=0.05, learning_max_itr=5, learning_tol=1e-7)
    e1, n1, r1 = tv.perform_inference()
    e2, n2, r2 = tv.optimize_prior_sde()
    out.append((e1 + e2, n1 + n2, r1 + r2, tv.prior_params[0]))
    return out


def main():
    torch.cuda.set_device(0)
    whole = run(0, 4)                                  # before the group exists
    rank, world = init_from_env(backend="gloo")
    part = run(*shard_bounds(4, rank, world))
    for (e, n, r, p), (e0, n0, r0, p0) in zip(part, whole):
        assert len(e) == len(e0) and len(p) == len(p0) and len(p0) > 3, (len(e), len(e0), len(p), len(p0))
        np.testing.assert_allclose(e, e0, rtol=1e-9)
        np.testing.assert_allclose(n, n0, rtol=1e-8)
        np.testing.assert_allclose(r, r0, rtol=1e-8)
        np.testing.assert_allclose(p, p0, rtol=1e-9)
        assert abs(p0[-1] - p0[0]) > 1e-2
    dist.barrier()
    if rank == 0:
        print("trainer shard parity ok", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
