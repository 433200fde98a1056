"""Config-2 step eager vs replayed from a HIP graph (development aid)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidp_amd import kernels as K  # noqa: E402
from vidp_amd.likelihoods import Gaussian  # noqa: E402
from vidp_amd.variational_cvi import CVIGaussianProcess  # noqa: E402


def model(T):
    rng = np.random.default_rng(71892305 + 2)
    t = torch.linspace(0, 0.01 * T, T, dtype=torch.float64, device="cuda")
    y = (torch.sin(12 * t) + 0.1 * torch.from_numpy(rng.normal(size=T)).cuda())[:, None]
    return CVIGaussianProcess((t, y), K.Matern52(lengthscale=0.2, variance=1.0), Gaussian(0.01), learning_rate=0.5)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n = 200
    a, b = model(T), model(T)
    for _ in range(3):
        a.update_sites(); ea = a.elbo()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        a.update_sites(); ea = a.elbo()
    torch.cuda.synchronize()
    print(f"eager: {1e3 * (time.perf_counter() - t0) / n:.4f} ms/step, elbo {float(ea):.9f}")
    step = b.step_graph()
    for _ in range(3):
        eb = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eb = step()
    torch.cuda.synchronize()
    print(f"graph: {1e3 * (time.perf_counter() - t0) / n:.4f} ms/step, elbo {float(eb):.9f}")
    b.update_sites()       # eager after replays
    print("mixed:", float(b.elbo()), "plan info", int(b.dist_p.plan.info.item()))


if __name__ == "__main__":
    main()
