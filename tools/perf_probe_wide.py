"""Quick timing of the wide (8 < d <= 32) sweeps at a given size (development aid, not the bench)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import vidp_amd  # noqa: E402


def main():
    B, T, d = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else (1, 200000, 16)
    R0 = int(sys.argv[4]) if len(sys.argv) >= 5 else 0
    Rup = int(sys.argv[5]) if len(sys.argv) >= 6 else 0
    plan = vidp_amd.Plan(B, T, d, R0=R0, Rup=Rup)
    print(f"B={B} T={T} d={d} nlevels={plan.nlevels} R={plan.R} P={plan.P}")
    g = torch.Generator(device="cuda").manual_seed(0)
    D = 0.1 * torch.randn((B, T, d, d), generator=g, device="cuda", dtype=torch.float64)
    D = D + D.transpose(-1, -2) + (4.0 + d ** 0.5) * torch.eye(d, device="cuda", dtype=torch.float64)
    S = (0.3 / d ** 0.5) * torch.randn((B, T, d, d), generator=g, device="cuda", dtype=torch.float64)
    r = torch.randn((B, T, d), generator=g, device="cuda", dtype=torch.float64)
    D, S, r = D.reshape(-1), S.reshape(-1), r.reshape(-1)
    mo = __import__("os").environ.get("PROBE_MOMENTS_ONLY", "0") != "0"
    f = plan.factor(D, S, r, moments_only=mo)
    form = f["form"]
    print("form", form)
    s = plan.selinv(f["L"], f["G"], f["y"], form=form)
    torch.cuda.synchronize()
    plan.check_info()
    for name, fn in (("factor", lambda: plan.factor(D, S, r, out=f, moments_only=mo)), ("selinv", lambda: plan.selinv(f["L"], f["G"], f["y"], out=s, form=form))):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        ev0.record()
        for _ in range(n):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / n
        nodes = B * T
        E = 2 * d * d + d
        byt = nodes * 8 * E * (3 if name == "factor" else 2)
        fl = nodes * d ** 3 * (2 * (1 / 3 + 1 + 1 + 1) + 2 * (1 / 3 + 1 + 1) if name == "factor" else 2 * (1 / 3 + 5))
        print(f"{name}: {ms:.3f} ms  {byt/ms/1e9:.3f} TB/s algorithmic, {fl/ms/1e9:.2f} TFLOP/s")


if __name__ == "__main__":
    main()
