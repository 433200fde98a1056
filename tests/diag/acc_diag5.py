"""Diagnostic: torch.cholesky_solve on the device for the forward no-smoothing map (d = 6 and d = 30)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels, np_transforms
for nc in (1, 2, 10):
    okern = np_kernels.Sum([np_kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(nc)]) if nc > 1 else np_kernels.Matern52(lengthscale=0.01, variance=0.01)
    ossm = okern.state_space_model(np.linspace(0, 1, 1001))
    th = np_transforms.ssm_to_naturals_no_smoothing(ossm)
    chols_np = ossm.concatenated_cholesky_process_covariance
    for devname in ("cpu", "cuda"):
        chols = torch.from_numpy(np.ascontiguousarray(chols_np)).to(devname)
        A = torch.from_numpy(np.ascontiguousarray(ossm.A)).to(devname)
        sub = torch.cholesky_solve(A, chols[1:])
        eye = torch.eye(chols.shape[-1], dtype=chols.dtype, device=chols.device).expand(chols.shape)
        dg = -0.5 * torch.cholesky_solve(eye, chols)
        dg2 = -0.5 * torch.cholesky_solve(eye.contiguous(), chols)
        rel = lambda a, b: float(np.max(np.abs(a.cpu().numpy() - b)) / np.max(np.abs(b)))
        print(nc, devname, "sub relerr %.2e diag relerr %.2e diag(contig eye) relerr %.2e" % (rel(sub, th[2]), rel(dg, th[1]), rel(dg2, th[1])), flush=True)
