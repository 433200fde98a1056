"""Time of tape.band_of_sigma_dP_sigma for wide blocks: the HIP route (mfgm_wband_sigma_dP_sigma) against the torch scans
(VIDP_TAPE_TORCH_SCAN=1) on the same marginals.   usage (through gpurun): python tools/wband_time.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd as amd
from vidp_amd import tape
from tests.helpers import random_dominant_btd


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = []
rng = np.random.default_rng(0)
for B, T, d in [(1, 1001, 30), (1, 20000, 16), (1, 20000, 30), (8, 5000, 12)]:
    diag, sub = random_dominant_btd(rng, (B,), T, d)
    dev = lambda x: torch.as_tensor(x, dtype=torch.float64, device="cuda")
    plan = amd.Plan(B, T, d)
    Dp, Sp = plan.pack(amd.SYM, dev(diag)), plan.pack(amd.FULL, dev(sub))

    def refresh():
        f = plan.factor(Dp, Sp, None, want_logdet=False, moments_only=True)
        return plan.selinv(f["L"], f["G"], None, want_sub=True, form=f["form"])

    s = refresh()
    cov, csub = plan.unpack(amd.SYM, s["Sig"]), plan.unpack(amd.FULL, s["Sub"], T - 1)
    dPd = dev(rng.normal(size=(B, T, d, d)))
    dPd = dPd + dPd.transpose(-1, -2)
    dPs = dev(rng.normal(size=(B, T - 1, d, d)))
    hip = timed(lambda: tape.band_of_sigma_dP_sigma(cov, csub, dPd, dPs))
    gd, gs = tape.band_of_sigma_dP_sigma(cov, csub, dPd, dPs)
    os.environ["VIDP_TAPE_TORCH_SCAN"] = "1"
    tor = timed(lambda: tape.band_of_sigma_dP_sigma(cov, csub, dPd, dPs))
    td, ts = tape.band_of_sigma_dP_sigma(cov, csub, dPd, dPs)
    del os.environ["VIDP_TAPE_TORCH_SCAN"]
    rel = float(((gd - td).abs().max() / td.abs().max()).item())
    out.append(dict(B=B, T=T, d=d, hip_ms=round(hip, 3), torch_ms=round(tor, 3), refresh_ms=round(timed(refresh), 3),
                    max_rel_diff=rel))
    print(json.dumps(out[-1]), flush=True)
