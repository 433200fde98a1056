"""
Helper launched by tests/test_gpu_wide.py::test_sparse_cvi_one_chain_two_processes under torch.distributed.run (2 ranks, gloo, one GPU):
config 5 in miniature as ONE sparse-CVI chain shared between the processes at the model level
(SparseCVIGaussianProcess(shard=(rank, world)), real process-group collectives); each process checks its ELBO sequence and its owned
sites over three damped steps against a whole-chain model run locally.  Exit code 0 = parity.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd  # noqa: E402,F401
from tests.helpers import assert_close  # noqa: E402
from vidp_amd import kernels as K  # noqa: E402
from vidp_amd.distributed import init_from_env  # noqa: E402
from vidp_amd.likelihoods import Gaussian  # noqa: E402
from vidp_amd.sparse_variational_cvi import SparseCVIGaussianProcess  # noqa: E402


def kernel():
    ls = np.exp(np.linspace(np.log(0.05), np.log(2.0), 6))
    return K.Sum([K.Matern52(float(l), 1.0) for l in ls[:4]] + [K.Matern32(float(l), 1.0) for l in ls[4:]])


def main():
    rank, world = init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    rng = np.random.default_rng(71892305)          # same data on every rank
    M, dz = 240, 0.1
    z = np.linspace(0, dz * M, M)
    t = np.sort(rng.uniform(-0.2, dz * M + 0.2, size=2 * M))
    y = (np.sin(3 * t) + 0.1 * rng.normal(size=t.size)).reshape(-1, 1)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    data = (dev(t), dev(y))
    whole = SparseCVIGaussianProcess(kernel(), dev(z), Gaussian(0.01), learning_rate=0.5)
    part = SparseCVIGaussianProcess(kernel(), dev(z), Gaussian(0.01), learning_rate=0.5, shard=(rank, world))
    for _ in range(3):
        whole.update_sites(data)
        part.update_sites(data)
        np.testing.assert_allclose(float(part.classic_elbo(data)), float(whole.classic_elbo(data)), rtol=1e-9)
    part._shard.plan.check_info()
    lo, hi = part._m_lo, part._m_hi
    host = lambda x: x.cpu().numpy()
    assert_close(host(part.nat1)[lo:hi], host(whole.nat1)[lo:hi], rtol=1e-9)
    assert_close(host(part.nat2)[lo:hi], host(whole.nat2)[lo:hi], rtol=1e-9)
    mu, cov = part.dist_q.marginals                # gathers the sites
    mu0, cov0 = whole.dist_q.marginals
    assert_close(host(mu), host(mu0), rtol=1e-8)
    assert_close(host(cov), host(cov0), rtol=1e-8)
    dist.barrier()
    if rank == 0:
        print("sparse chain shard parity ok", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
