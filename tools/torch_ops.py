"""Which torch operators run inside a bench step (torch.profiler, leaf aten ops with device time): headline (CVI-DP) or c3 (VDP)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import vidp_amd
from vidp_amd import sde as gsde
from vidp_amd.likelihoods import MultivariateGaussian
from torch.profiler import profile, ProfilerActivity

which = sys.argv[1] if len(sys.argv) > 1 else "headline"
B, d, dt, noise = 64, 6, 0.01, 0.1
dev = torch.device("cuda", 0)
lik = MultivariateGaussian(torch.from_numpy(bench.obs_chol(d, noise)).to(dev))
q = torch.eye(d, dtype=torch.float64)
if which == "headline":
    from vidp_amd.variational_cvi_sde import CVISitesSDE
    T = 100000
    idx, ys = bench.synth_double_well(B, T, d, dt, 50, noise, seed=5)
    grid = np.arange(T) * dt
    m = CVISitesSDE(gsde.DoubleWellSDE(q=q), grid, (grid[idx], torch.from_numpy(ys).to(dev)), lik, prior_initial_state=(np.zeros(d), np.eye(d)),
                    plan=vidp_amd.Plan(B, T, d))

    def step():
        m.update_data_sites(0.5); m.update_girsanov_sites(0.1); return m.classic_elbo_per_trajectory()
else:
    from vidp_amd.vi_sde import VariationalMarkovGP
    T = 50000
    idx, ys = bench.synth_double_well(B, T, d, dt, 50, noise, seed=5)
    grid = np.arange(T) * dt
    m = VariationalMarkovGP((grid[idx], torch.from_numpy(ys).to(dev)), gsde.DoubleWellSDE(q=q), grid, lik, prior_initial_state=(np.zeros(d), np.eye(d)),
                            stabilize_system=True, plan=vidp_amd.Plan(B, T, d))
    m.plan.pack(vidp_amd.FULL, (4.0 * torch.eye(d, dtype=torch.float64, device=dev)).expand(B, T, d, d).contiguous(), out=m.A)
    st = {"mS": m._forward_packed()}

    def step():
        m.update_lagrange_and_param(st["mS"], lr=0.01); st["mS"] = m._forward_packed(); return m.elbo(st["mS"])
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
tot = 0.0
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.device_time_total > 0 and not any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        tot += ev.device_time_total
        print(ev.name, str(ev.input_shapes)[:70], round(ev.device_time_total, 1))
print("torch device time in one step:", round(tot, 1), "us")
