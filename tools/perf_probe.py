"""Quick timing of the packed sweeps at a given size (development aid, not the bench)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import vidp_amd  # noqa: E402


def main():
    B, T, d = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else (64, 100000, 6)
    R0 = int(sys.argv[4]) if len(sys.argv) >= 5 else 0
    Rup = int(sys.argv[5]) if len(sys.argv) >= 6 else 0
    plan = vidp_amd.Plan(B, T, d, R0=R0, Rup=Rup)
    print(f"B={B} T={T} d={d} nlevels={plan.nlevels} R={plan.R} P={plan.P} Lpad={plan.Lpad}")
    g = torch.Generator(device="cuda").manual_seed(0)
    ET = d * (d + 1) // 2
    # diagonally dominant SPD system directly in packed form
    D = plan.zeros(vidp_amd.SYM).view(plan.Lpad // 64, plan.R, ET, 64)
    D.copy_(0.1 * torch.randn(D.shape, generator=g, device="cuda", dtype=torch.float64))
    for r in range(d):
        D[:, :, r * (r + 1) // 2 + r, :] = 4.0 + torch.rand((plan.Lpad // 64, plan.R, 64), generator=g, device="cuda", dtype=torch.float64)
    S = 0.3 * torch.randn(plan.R * d * d * plan.Lpad, generator=g, device="cuda", dtype=torch.float64)
    r = torch.randn(plan.R * d * plan.Lpad, generator=g, device="cuda", dtype=torch.float64)
    D = D.view(-1)
    f = plan.factor(D, S, r)
    s = plan.selinv(f["L"], f["G"], f["y"])
    torch.cuda.synchronize()
    plan.check_info()
    for name, fn in (("factor", lambda: plan.factor(D, S, r, out=f)), ("selinv", lambda: plan.selinv(f["L"], f["G"], f["y"], out=s))):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        ev0.record()
        for _ in range(n):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / n
        nodes = B * T
        if name == "factor":
            byt = nodes * 8 * ((ET + d * d + d) * 2 + (ET + d * d + d))   # reduce reads + forward reads/writes
        else:
            byt = nodes * 8 * (ET + d * d + d) * 2
        print(f"{name}: {ms:.3f} ms  moved {byt/1e9:.2f} GB  -> {byt/ms/1e9:.2f} TB/s")


if __name__ == "__main__":
    main()
