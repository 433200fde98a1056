"""Development probe: the headline CVI-DP step with the 64 trajectories split into G groups, each with its own model, plan and HIP
stream, so that one group's latency-bound coarse levels run under the other groups' level-0 sweeps.
    python tools/groups_probe.py G R0 [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd  # noqa: E402
from bench import obs_chol, synth_double_well  # noqa: E402
from vidp_amd.likelihoods import MultivariateGaussian  # noqa: E402
from vidp_amd.sde import DoubleWellSDE  # noqa: E402
from vidp_amd.variational_cvi_sde import CVISitesSDE  # noqa: E402


def main():
    G, R0 = int(sys.argv[1]), int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    B, T, d, dt, noise = 64, 100000, 6, 0.01, 0.1
    device = torch.device("cuda", 0)
    idx, ys = synth_double_well(B, T, d, dt, 50, noise, seed=71892305 + 3)
    grid = np.arange(T) * dt
    bounds = [round(B * g / G) for g in range(G + 1)]
    models, streams = [], []
    for g in range(G):
        lo, hi = bounds[g], bounds[g + 1]
        st = torch.cuda.Stream() if G > 1 else torch.cuda.current_stream()
        with torch.cuda.stream(st):
            lik = MultivariateGaussian(torch.from_numpy(obs_chol(d, noise)).to(device))
            plan = vidp_amd.Plan(hi - lo, T, d, R0=R0, device=device)
            m = CVISitesSDE(DoubleWellSDE(q=torch.eye(d, dtype=torch.float64)), grid, (grid[idx], torch.from_numpy(ys[lo:hi]).to(device)), lik,
                            prior_initial_state=(np.zeros(d), np.eye(d)), plan=plan)
        models.append(m)
        streams.append(st)
    torch.cuda.synchronize()
    print(f"G={G} R0={R0} plans: " + ", ".join(f"B={m.plan.B} R={m.plan.R} lanes={m.plan.Lpad} levels={m.plan.nlevels}" for m in models))
    e = [None] * G

    def step():
        for g, m in enumerate(models):
            with torch.cuda.stream(streams[g]):
                m.update_data_sites(0.5)
                m.update_girsanov_sites(0.1)
                e[g] = m.classic_elbo_per_trajectory()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tot = float(sum(x.sum() for x in e))
    print(f"G={G} R0={R0}: {1e3 * el / steps:.3f} ms/step ({steps / el:.1f} steps/s), host issue {1e3 * t_issue / steps:.3f} ms/step, elbo {tot:.6f}")


if __name__ == "__main__":
    main()
