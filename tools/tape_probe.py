"""Timing of the tape's backward pass (exact Fisher-vector product) against a forward refresh (factor + selected inverse), natural-layout
tensors in and out, at config 2's size and at a batch of chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vidp_amd
from vidp_amd import tape


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for B, T, d in ((1, 100000, 3), (8, 100000, 6)):
    g = torch.Generator(device="cuda").manual_seed(1)
    sub = 0.3 * torch.randn(B, T - 1, d, d, dtype=torch.float64, device="cuda", generator=g)
    diag = torch.randn(B, T, d, d, dtype=torch.float64, device="cuda", generator=g)
    diag = diag @ diag.transpose(-1, -2) + 4.0 * d * torch.eye(d, dtype=torch.float64, device="cuda")
    lin = torch.randn(B, T, d, dtype=torch.float64, device="cuda", generator=g)
    th_d, th_s = -0.5 * diag, -sub                        # naturals of the precision (diag, sub)
    plan = vidp_amd.Plan(B, T, d)
    mu, cov, csub = tape._marginals(plan, lin, th_d, th_s)
    gl, gd, gs = torch.randn_like(lin), torch.randn_like(diag), torch.randn_like(sub)
    t_fwd = timed(lambda: tape._marginals(plan, lin, th_d, th_s))
    t_bwd = timed(lambda: tape.fisher_vector_product(plan, th_d, th_s, mu, cov, csub, gl, gd, gs))
    t_band = timed(lambda: tape.band_of_sigma_dP_sigma(cov, csub, -2.0 * gd, -1.0 * gs, plan=plan))
    print(f"B={B} T={T} d={d}: forward refresh (pack, factor, selected inverse, unpack) {1e3 * t_fwd:.2f} ms, Fisher-vector product "
          f"{1e3 * t_bwd:.2f} ms ({t_bwd / t_fwd:.1f} x), of which the band of Sigma dP Sigma {1e3 * t_band:.2f} ms")
