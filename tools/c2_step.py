"""Config-2 step on ONE GPU (development aid): CVIGaussianProcess.update_sites() + elbo() with a Matern-5/2 kernel, T points, one chain."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidp_amd import kernels as K  # noqa: E402
from vidp_amd.likelihoods import Gaussian  # noqa: E402
from vidp_amd.variational_cvi import CVIGaussianProcess  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    rng = np.random.default_rng(71892305 + 2)
    t = torch.linspace(0, 0.01 * T, T, dtype=torch.float64, device="cuda")
    y = (torch.sin(12 * t) + 0.1 * torch.from_numpy(rng.normal(size=T)).cuda())[:, None]
    m = CVIGaussianProcess((t, y), K.Matern52(lengthscale=0.2, variance=1.0), Gaussian(0.01), learning_rate=0.5)
    for it in range(3):
        m.update_sites()
        e = float(m.elbo())
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        m.update_sites()
        e = float(m.elbo())
    torch.cuda.synchronize()
    pl = m.dist_p.plan
    print(f"T={T}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per update_sites + elbo; elbo {e:.6f}; plan levels {pl.nlevels} R {pl.R} lanes {pl.Lpad}")
    t0 = time.perf_counter()
    for _ in range(n):
        e = float(m.elbo())
    torch.cuda.synchronize()
    print(f"elbo alone (Kalman log-likelihood with sites): {1e3 * (time.perf_counter() - t0) / n:.3f} ms")
    os.environ["VIDP_FUSED_KF"] = "0"
    m2 = CVIGaussianProcess((t, y), K.Matern52(lengthscale=0.2, variance=1.0), Gaussian(0.01), learning_rate=0.5)
    for it in range(53):
        m2.update_sites()
        e2 = float(m2.elbo())
    print(f"unfused route after the same 53 steps: elbo {e2:.6f} (rel diff {abs(e - e2) / abs(e2):.2e})")


if __name__ == "__main__":
    main()
