"""Diagnostic: repeatability of torch's device cholesky / cholesky_solve on batches of small ill-conditioned blocks."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels, np_transforms
for nc in (1, 2, 3):
    okern = np_kernels.Sum([np_kernels.Matern52(lengthscale=0.01, variance=0.01) for _ in range(nc)]) if nc > 1 else np_kernels.Matern52(lengthscale=0.01, variance=0.01)
    ossm = okern.state_space_model(np.linspace(0, 1, 1001))
    tho = np_transforms.ssm_to_naturals_no_smoothing(ossm)
    d = 3 * nc
    res = {}
    for devname in ("cpu", "cuda"):
        tl, td, ts = [torch.from_numpy(np.ascontiguousarray(x)).to(devname) for x in tho]
        for rep in range(3):
            c = torch.linalg.cholesky(-2.0 * td)
            As1 = torch.cholesky_solve(ts, c[1:])
            As2 = torch.cholesky_solve(ts, c[..., 1:, :, :])
            off = torch.cholesky_solve(tl[..., None], c)[..., 0]
            eye = torch.eye(d, dtype=c.dtype, device=c.device).expand(c.shape)
            inv = torch.cholesky_solve(eye, c)
            ch, info = torch.linalg.cholesky_ex(inv)
            r = lambda a, b: float(np.nanmax(np.abs(a.cpu().numpy() - b)) / np.max(np.abs(b)))
            print(nc, devname, rep, "As1 %.1e As2 %.1e chol %.1e info %d c-recon %.1e" % (r(As1, ossm.A), r(As2, ossm.A), r(ch[1:], ossm.cholQ), int(info.max()),
                  float((c @ c.transpose(-1, -2) + 2 * td).abs().max() / td.abs().max())), flush=True)
