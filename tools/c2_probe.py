"""Timing at config-2 shape on ONE GPU (development aid): Matern-5/2 GP regression, T = 100 000 points, one trajectory:
kernel -> SSM, Kalman log-likelihood (one block Cholesky) and posterior marginals (factor + selected inverse)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidp_amd import kernels as K  # noqa: E402
from vidp_amd.variational_cvi import GaussianProcessRegression  # noqa: E402


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, out


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    rng = np.random.default_rng(71892305 + 2)
    t = torch.linspace(0, 0.01 * T, T, dtype=torch.float64, device="cuda")
    y = (torch.sin(t) + 0.1 * torch.from_numpy(rng.normal(size=T)).cuda())[:, None]
    kern = K.Matern52(lengthscale=0.5, variance=1.0)
    gpr = GaussianProcessRegression((t, y), kern, torch.tensor([[0.1]], dtype=torch.float64, device="cuda"))
    ms, ssm = timeit(lambda: kern.state_space_model(t))
    print(f"kernel -> SSM: {ms:.3f} ms")
    ms, ll = timeit(lambda: gpr.log_likelihood())
    print(f"log_likelihood (kernel + precision + block Cholesky): {ms:.3f} ms   value {float(ll):.6f}")
    kf = gpr._kalman
    ms, post = timeit(lambda: kf.posterior_state_space_model())
    print(f"posterior SSM (factor + selected inverse + local maps): {ms:.3f} ms")
    ms, _ = timeit(lambda: post.marginals)
    pl = ssm.plan
    print(f"plan: levels {pl.nlevels}, R {pl.R}, lanes {pl.Lpad}")


if __name__ == "__main__":
    main()
