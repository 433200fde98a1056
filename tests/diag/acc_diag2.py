"""Diagnostic: where does the cholQ round-trip error come from (sweeps vs torch's device linalg)?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels, np_transforms
from vidp_amd import kernels, ssm_gaussian_transformations as tr

def viol(a, b):
    return float(np.max(np.abs(a - b) - (1e-6 + 1e-7 * np.abs(b))))

kern = kernels.Matern52(lengthscale=0.01, variance=0.01)
ssm = kern.state_space_model(torch.linspace(0, 1, 1001, dtype=torch.float64, device="cuda"))
okern = np_kernels.Matern52(lengthscale=0.01, variance=0.01)
ossm = okern.state_space_model(np.linspace(0, 1, 1001))
ref = (ossm.A, ossm.b, ossm.cholP0, ossm.cholQ, ossm.mu0)
eg = [x.cpu().numpy() for x in tr.ssm_to_expectations(ssm)]
eo = np_transforms.ssm_to_expectations(ossm)
print("expectations gpu vs oracle viol", [viol(a, b) for a, b in zip(eg, eo)])
# oracle formulas on gpu-produced expectations
back = np_transforms.expectations_to_ssm_params(*eg)
print("numpy formulas on GPU expectations: cholQ viol", viol(back[3], ref[3]), "A", viol(back[0], ref[0]))
back = np_transforms.expectations_to_ssm_params(*eo)
print("numpy formulas on oracle expectations: cholQ viol", viol(back[3], ref[3]))
# torch formulas, cpu vs gpu, on oracle expectations
for devname in ("cpu", "cuda"):
    te = [torch.from_numpy(np.ascontiguousarray(x)).to(devname) for x in eo]
    back = tr.expectations_to_ssm_params(*te)
    print("torch formulas on", devname, "oracle expectations: cholQ viol", viol(back[3].cpu().numpy(), ref[3]), "A", viol(back[0].cpu().numpy(), ref[0]))
# naturals
ng = [x.cpu().numpy() for x in tr.ssm_to_naturals(ssm)]
no = np_transforms.ssm_to_naturals(ossm)
print("naturals gpu vs oracle viol", [viol(a, b) for a, b in zip(ng, no)])
back = np_transforms.naturals_to_ssm_params(*ng)
print("numpy naturals_to_ssm on GPU naturals: cholQ viol", viol(back[3], ref[3]))
back = tr.naturals_to_ssm_params(*[torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in no])
print("gpu naturals_to_ssm on oracle naturals: cholQ viol", viol(back[3].cpu().numpy(), ref[3]))
