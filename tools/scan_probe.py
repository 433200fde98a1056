"""Timing of the exact Fisher-vector product's congruence scans: HIP (mfgm_congruence_scan) against the torch Hillis-Steele scan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vidp_amd
from vidp_amd import tape

for B, T, d in ((1, 100000, 3), (8, 100000, 6), (64, 100000, 6)):
    g = torch.Generator(device="cuda").manual_seed(1)
    Phi = torch.randn(B, T, d, d, dtype=torch.float64, device="cuda", generator=g) * (0.9 / d ** 0.5)
    Q = torch.randn(B, T, d, d, dtype=torch.float64, device="cuda", generator=g)
    Q = Q @ Q.transpose(-1, -2)
    plan = vidp_amd.Plan(B, T, d)
    res = {}
    for name, pl in (("hip", plan), ("torch", None)):
        if name == "torch" and B * T * d * d > 5e7:
            continue
        X = tape._congruence_scan(Phi, Q, plan=pl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            X = tape._congruence_scan(Phi, Q, plan=pl)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 3
        res[name + "_X"] = X
    line = f"B={B} T={T} d={d}: " + ", ".join(f"{k} {1e3 * v:.2f} ms" for k, v in res.items() if not k.endswith("_X"))
    if "torch_X" in res:
        line += f", max rel diff {float(((res['hip_X'] - res['torch_X']).abs().max() / res['torch_X'].abs().max())):.1e}"
    print(line)
    del Phi, Q, X, res
    torch.cuda.empty_cache()
