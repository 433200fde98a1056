"""Timing of one sparse-CVI step at config-5 shape on ONE GPU (development aid): Sum-of-Matern kernel with state dimension 16,
M inducing states on a regular grid, N noisy observations of a smooth function."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidp_amd import kernels as K  # noqa: E402
from vidp_amd.likelihoods import Gaussian  # noqa: E402
from vidp_amd.sparse_variational_cvi import SparseCVIGaussianProcess  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    span = M * 0.01
    kern = K.Sum([K.Matern52(lengthscale=0.02 + 0.01 * i, variance=1.0 / (i + 1)) for i in range(5)] + [K.Matern12(0.05, 0.5)])
    assert kern.state_dim == 16
    rng = np.random.default_rng(71892305 + 5)
    z = torch.linspace(0, span, M, dtype=torch.float64, device="cuda")
    t = torch.from_numpy(np.sort(rng.uniform(0, span, size=N))).cuda()
    y = (torch.sin(3 * t) + 0.3 * torch.from_numpy(rng.normal(size=N)).cuda())[:, None]
    model = SparseCVIGaussianProcess(kern, z, Gaussian(0.09), learning_rate=0.5)
    torch.cuda.synchronize()
    elbos = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        model.update_sites((t, y))
        e = float(model.classic_elbo((t, y)))
        torch.cuda.synchronize()
        elbos.append(e)
        print(f"step {it}: {1e3 * (time.perf_counter() - t0):.1f} ms  elbo {e:.6e}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
