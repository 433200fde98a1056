"""ELBO of the config-5 model vs the NumPy oracle as a function of the grid spacing (conditioning), both sweep forms (development aid)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd  # noqa: E402
from oracle import np_conditionals as npc, np_kernels, np_models  # noqa: E402
from vidp_amd import kernels as K  # noqa: E402
from vidp_amd.likelihoods import Gaussian  # noqa: E402
from vidp_amd.sparse_variational_cvi import SparseCVIGaussianProcess  # noqa: E402

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def main():
    ls = np.exp(np.linspace(np.log(0.05), np.log(2.0), 6))
    mk = lambda mod: mod.Sum([mod.Matern52(float(l), 1.0) for l in ls[:4]] + [mod.Matern32(float(l), 1.0) for l in ls[4:]])
    M = 100
    for dz in (0.01, 0.03, 0.1):
        rng = np.random.default_rng(1)
        z = np.linspace(0, dz * M, M)
        t = np.sort(rng.uniform(0, dz * M, size=2 * M))
        y = (np.sin(3 * t) + 0.1 * rng.normal(size=t.size)).reshape(-1, 1)
        o = npc.SparseCVIGaussianProcess(mk(np_kernels), z, np_models.GaussianLik(0.01), learning_rate=0.5)
        ref = []
        for _ in range(4):
            o.update_sites(t, y)
            ref.append(o.classic_elbo(t, y))
        pd, _ = o.dist_p.precision()
        cond = max(np.linalg.cond(pd[k]) for k in range(1, M - 1))
        for inv in ("0", "1"):
            os.environ["VIDP_SPARSE_INVERSE_FORM"] = inv
            g = SparseCVIGaussianProcess(mk(K), dev(z), Gaussian(0.01), learning_rate=0.5)
            got = []
            for _ in range(4):
                g.update_sites((dev(t), dev(y)))
                got.append(float(g.classic_elbo((dev(t), dev(y)))))
            err = [abs(a - b) / abs(b) for a, b in zip(got, ref)]
            print(f"dz={dz} cond(prior precision block)={cond:.1e} inverse_form={inv}: rel err per step " + " ".join(f"{e:.1e}" for e in err), " elbo", f"{ref[-1]:.4f}")


if __name__ == "__main__":
    main()
