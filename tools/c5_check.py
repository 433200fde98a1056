"""ELBO trajectory of the config-5 model under the different sweep paths (development aid)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd  # noqa: E402
from vidp_amd import kernels as K  # noqa: E402
from vidp_amd.likelihoods import Gaussian  # noqa: E402
from vidp_amd.sparse_variational_cvi import SparseCVIGaussianProcess  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    d = 16
    N, span = 2 * M, 0.1 * M
    rng = np.random.default_rng(71892305 + 5)
    z = torch.linspace(0, span, M, dtype=torch.float64, device="cuda")
    t = torch.from_numpy(np.sort(rng.uniform(0, span, size=N))).cuda()
    y = (torch.sin(3 * t) + 0.1 * torch.from_numpy(rng.normal(size=N)).cuda())[:, None]
    ls = np.exp(np.linspace(np.log(0.05), np.log(2.0), 6))
    kern = K.Sum([K.Matern52(float(l), 1.0) for l in ls[:4]] + [K.Matern32(float(l), 1.0) for l in ls[4:]])
    m = SparseCVIGaussianProcess(kern, z, Gaussian(0.01), learning_rate=0.5)
    out = []
    for _ in range(steps):
        m.update_sites((t, y))
        out.append(float(m.classic_elbo((t, y))))
    m.dist_p.plan.check_info()
    print("inverse_form", os.environ.get("VIDP_SPARSE_INVERSE_FORM", "1"), "fused_theta", os.environ.get("VIDP_FUSED_THETA", "1"), " ".join(f"{e:.6f}" for e in out))


if __name__ == "__main__":
    main()
