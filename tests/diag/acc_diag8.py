"""Diagnostic: the diag6 sequence with and without running any libmfgm kernel first."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels, np_transforms
use_lib = len(sys.argv) > 1 and sys.argv[1] == "lib"
nc = 2
okern = np_kernels.Sum([np_kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(nc)])
ossm = okern.state_space_model(np.linspace(0, 1, 1001))
from vidp_amd import kernels, ssm_gaussian_transformations as tr
from vidp_amd.state_space_model import StateSpaceModel
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
if use_lib:
    kern = kernels.Sum([kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(nc)])
    ssm = kern.state_space_model(torch.linspace(0, 1, 1001, dtype=torch.float64, device="cuda"))
    _ = ssm.state_transitions
else:
    ssm = StateSpaceModel(dev(ossm.mu0), dev(ossm.cholP0), dev(ossm.A), dev(ossm.b), dev(ossm.cholQ))
rel = lambda a, b: float(np.nanmax(np.abs(a.cpu().numpy() - b)) / np.max(np.abs(b))) if np.max(np.abs(b)) > 0 else float(a.abs().max())
th = tr.ssm_to_naturals_no_smoothing(ssm)
tho = np_transforms.ssm_to_naturals_no_smoothing(ossm)
ref = (ossm.A, ossm.b, ossm.cholP0, ossm.cholQ, ossm.mu0)
print("forward", [rel(a, b) for a, b in zip(th, tho)])
for name, args in (("th", th), ("oracle", [dev(x) for x in tho]), ("clones", [x.contiguous().clone() for x in th]), ("oracle again", [dev(x) for x in tho])):
    try:
        back = tr.naturals_to_ssm_params_no_smoothing(*args)
        print(name, [rel(a, b) for a, b in zip(back, ref)], flush=True)
    except Exception as e:
        print(name, "raised", str(e)[:150], flush=True)
