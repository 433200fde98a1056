"""
VDP (VariationalMarkovGP) inference step at the headline size, one GPU: the loop body of VIMarkovGPTrainer.perform_inference
(Lagrange sweep, parameter update, forward pass, ELBO) on B double-well trajectories of T steps, d dimensions,
with the reference's stabilize_system clipping (a chain this long overflows the multiplier recursion without it, vi_sde.py:59-60).

    python tools/vdp_probe.py [B T d steps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (synthetic data of the bench workload)
import vidp_amd  # noqa: E402
from vidp_amd.likelihoods import MultivariateGaussian  # noqa: E402
from vidp_amd.sde import DoubleWellSDE  # noqa: E402
from vidp_amd.vi_sde import VariationalMarkovGP  # noqa: E402


LR = float(os.environ.get("VDP_LR", "0.01"))      # q_lr; the reference's experiments use 0.01 ... 0.1 with dt-dependent stability


def main():
    B, T, d, steps = (int(a) for a in (sys.argv[1:5] + ["64", "100000", "6", "20"][len(sys.argv) - 1:]))
    dev = torch.device("cuda", 0)
    dt, noise = 0.01, 0.1
    idx, ys = bench.synth_double_well(B, T, d, dt, 50, noise, seed=7)
    grid = np.arange(T) * dt
    lik = MultivariateGaussian(torch.from_numpy(bench.obs_chol(d, noise)).to(dev))
    m = VariationalMarkovGP((grid[idx], torch.from_numpy(ys).to(dev)), DoubleWellSDE(q=torch.eye(d, dtype=torch.float64)), grid, lik,
                            prior_initial_state=(np.zeros(d), np.eye(d)), stabilize_system=True, plan=vidp_amd.Plan(B, T, d, device=dev))
    # q starts at the OU drift -4 x (the role a CVI-DP warm start plays, exp_io.warm_start_vdp_from_cvi): from A = 0 the marginal
    # variance of a chain this long reaches T dt = 1000 and the double-well moments of order six overflow the first update
    eye = (4.0 * torch.eye(d, dtype=torch.float64, device=dev)).expand(B, T, d, d).contiguous()
    m.plan.pack(vidp_amd.FULL, eye, out=m.A)
    ev = {}

    def timed(name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        ev.setdefault(name, []).append((a, b))
        return out

    state = {"mS": m._forward_packed()}

    def step():
        # as the trainer: the marginals that close one iteration open the next
        mS = state["mS"]
        timed("lagrange+param", lambda: m.update_lagrange_and_param(mS, lr=LR))
        state["mS"] = mS = timed("forward_pass", m._forward_packed)
        return timed("elbo", lambda: m.elbo(mS))

    for _ in range(3):
        e = step()
    torch.cuda.synchronize()
    ev.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        e = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m.plan.check_info()
    print(f"VDP B={B} T={T} d={d}: {steps / el:.1f} ELBO steps/s ({1e3 * el / steps:.2f} ms/step), elbo {float(e):.6g}")
    for k, v in ev.items():
        print(f"  {k:16s} {sum(a.elapsed_time(b) for a, b in v) / steps:7.3f} ms/step ({len(v) // steps} calls)")


if __name__ == "__main__":
    main()
