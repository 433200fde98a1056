"""Diagnostic: per-output error of the three transformation round trips on the reference's KA5 setup (GPU vs truth)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels
from vidp_amd import kernels, ssm_gaussian_transformations as tr

def run(ncomp):
    kern = kernels.Sum([kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(ncomp)]) if ncomp > 1 else kernels.Matern52(lengthscale=0.01, variance=0.01)
    ssm = kern.state_space_model(torch.linspace(0, 1, 1001, dtype=torch.float64, device="cuda"))
    okern = np_kernels.Sum([np_kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(ncomp)]) if ncomp > 1 else np_kernels.Matern52(lengthscale=0.01, variance=0.01)
    ossm = okern.state_space_model(np.linspace(0, 1, 1001))
    ref = (ossm.A, ossm.b, ossm.cholP0, ossm.cholQ, ossm.mu0)
    names = ("A", "b", "cholP0", "cholQ", "mu0")
    for fwd, bwd in ((tr.ssm_to_expectations, tr.expectations_to_ssm_params), (tr.ssm_to_naturals, tr.naturals_to_ssm_params),
                     (tr.ssm_to_naturals_no_smoothing, tr.naturals_to_ssm_params_no_smoothing)):
        back = bwd(*fwd(ssm))
        out = []
        for n, a, b in zip(names, back, ref):
            a = a.cpu().numpy()
            err = np.abs(a - b)
            viol = err - (1e-6 + 1e-7 * np.abs(b))
            out.append(f"{n}: maxabs {err.max():.2e} maxviol {viol.max():.2e} scale {np.abs(b).max():.1e}")
        print(ncomp, fwd.__name__, " | ".join(out), flush=True)

for nc in (1, 2, 10):
    run(nc)
