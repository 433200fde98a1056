"""Diagnostic: torch device linalg vs CPU on the no-smoothing maps (two Matern-5/2 components, d = 6)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels, np_transforms
okern = np_kernels.Sum([np_kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(2)])
ossm = okern.state_space_model(np.linspace(0, 1, 1001))
th = np_transforms.ssm_to_naturals_no_smoothing(ossm)
for devname in ("cpu", "cuda"):
    td = torch.from_numpy(np.ascontiguousarray(th[1])).to(devname)
    c = torch.linalg.cholesky(-2.0 * td)
    print(devname, "chol nan:", bool(torch.isnan(c).any()), "recon err", float(((c @ c.transpose(-1, -2)) + 2.0 * td).abs().max() / td.abs().max()))
    ts = torch.from_numpy(np.ascontiguousarray(th[2])).to(devname)
    As = torch.cholesky_solve(ts, c[1:])
    print(devname, "As err", float((As.cpu() - torch.from_numpy(ossm.A)).abs().max()))
    eye = torch.eye(6, dtype=c.dtype, device=c.device).expand(c.shape)
    inv = torch.cholesky_solve(eye, c)
    print(devname, "inv nan", bool(torch.isnan(inv).any()))
    try:
        ch = torch.linalg.cholesky(inv)
        print(devname, "chol(inv) err", float((ch[1:].cpu() - torch.from_numpy(ossm.cholQ)).abs().max()))
    except Exception as e:
        print(devname, "chol(inv) raised", str(e)[:200])
    ch2, info = torch.linalg.cholesky_ex(inv)
    print(devname, "cholesky_ex info max", int(info.max()))
