"""Diagnostic: locate the failing step of the no-smoothing round trip for a two-component Sum kernel on the device."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels, np_transforms
from vidp_amd import kernels, ssm_gaussian_transformations as tr
nc = 2
kern = kernels.Sum([kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(nc)])
ssm = kern.state_space_model(torch.linspace(0, 1, 1001, dtype=torch.float64, device="cuda"))
okern = np_kernels.Sum([np_kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(nc)])
ossm = okern.state_space_model(np.linspace(0, 1, 1001))
rel = lambda a, b: float(np.nanmax(np.abs(a.cpu().numpy() - b)) / np.max(np.abs(b)))
print("A", rel(ssm.state_transitions, ossm.A), "chols", rel(ssm.concatenated_cholesky_process_covariance, ossm.concatenated_cholesky_process_covariance),
      "off", float(ssm.concatenated_state_offsets.abs().max()))
th = tr.ssm_to_naturals_no_smoothing(ssm)
tho = np_transforms.ssm_to_naturals_no_smoothing(ossm)
print("forward: ", [rel(a, b) if np.max(np.abs(b)) > 0 else float(a.abs().max()) for a, b in zip(th, tho)], [tuple(a.shape) for a in th], [a.is_contiguous() for a in th])
back = tr.naturals_to_ssm_params_no_smoothing(*th)
ref = (ossm.A, ossm.b, ossm.cholP0, ossm.cholQ, ossm.mu0)
print("backward:", [rel(a, b) if np.max(np.abs(b)) > 0 else float(a.abs().max()) for a, b in zip(back, ref)])
back = tr.naturals_to_ssm_params_no_smoothing(*[torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in tho])
print("backward from oracle naturals:", [rel(a, b) if np.max(np.abs(b)) > 0 else float(a.abs().max()) for a, b in zip(back, ref)])
back = tr.naturals_to_ssm_params_no_smoothing(*[x.contiguous().clone() for x in th])
print("backward from contiguous clones:", [rel(a, b) if np.max(np.abs(b)) > 0 else float(a.abs().max()) for a, b in zip(back, ref)])
