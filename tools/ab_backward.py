"""A/B of the two level-0 backward kernels on the same data and box: with the stored L_{t+1,t} and rebuilt from S (USE_S)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd  # noqa: E402
from vidp_amd.packed import _ptr, _stream  # noqa: E402

B, T, d = 64, 100000, 6
plan = vidp_amd.Plan(B, T, d)
g = torch.Generator(device="cuda").manual_seed(0)
ET = d * (d + 1) // 2
D = plan.zeros(vidp_amd.SYM).view(plan.Lpad // 64, plan.R, ET, 64)
D.copy_(0.1 * torch.randn(D.shape, generator=g, device="cuda", dtype=torch.float64))
for r in range(d):
    D[:, :, r * (r + 1) // 2 + r, :] = 4.0 + torch.rand((plan.Lpad // 64, plan.R, 64), generator=g, device="cuda", dtype=torch.float64)
D = D.view(-1)
S = 0.3 * torch.randn(plan.R * d * d * plan.Lpad, generator=g, device="cuda", dtype=torch.float64)
r = torch.randn(plan.R * d * plan.Lpad, generator=g, device="cuda", dtype=torch.float64)
f = plan.factor(D, S, r)
s = plan.selinv_mom(f["L"], f["G"], f["y"])
s2 = plan.selinv_mom(f["L"], None, f["y"], S=S, aS=1.0)
torch.cuda.synchronize()
print("max diff Sig", float((s["Sig"] - s2["Sig"]).abs().max()), "mom", float((s["mom"] - s2["mom"]).abs().max()))
lib = plan.lib


def t(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with_g = lambda: lib.mfgm_packed_selinv_mom(plan.h, 0, _ptr(f["L"]), _ptr(f["G"]), _ptr(f["y"]), _ptr(s["Sig"]), None, _ptr(s["x"]),
                                            _ptr(s["mom"]), _ptr(plan.ws), _stream())
from_s = lambda: lib.mfgm_packed_selinv_mom_s(plan.h, 0, _ptr(f["L"]), _ptr(S), 1.0, _ptr(f["y"]), _ptr(s2["Sig"]), _ptr(s2["x"]),
                                              _ptr(s2["mom"]), _ptr(plan.ws), _stream())
for rep in range(4):
    print(rep, "stored G %.4f ms   from S %.4f ms" % (t(with_g), t(from_s)), flush=True)
