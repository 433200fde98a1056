import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench, vidp_amd
from vidp_amd.likelihoods import MultivariateGaussian
from vidp_amd.sde import DoubleWellSDE
from vidp_amd.vi_sde import VariationalMarkovGP
B, T, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stab = len(sys.argv) > 4 and sys.argv[4] == "1"
dev = torch.device("cuda", 0)
dt, noise = 0.01, 0.1
idx, ys = bench.synth_double_well(B, T, d, dt, 50, noise, seed=7)
grid = np.arange(T) * dt
lik = MultivariateGaussian(torch.from_numpy(bench.obs_chol(d, noise)).to(dev))
m = VariationalMarkovGP((grid[idx], torch.from_numpy(ys).to(dev)), DoubleWellSDE(q=torch.eye(d, dtype=torch.float64)), grid, lik,
                        prior_initial_state=(np.zeros(d), np.eye(d)), stabilize_system=stab, plan=vidp_amd.Plan(B, T, d, device=dev))
def chk(tag, *ts):
    info = int(m.plan.info.item()); m.plan.info.zero_()
    print(tag, "info", info, [(bool(torch.isfinite(t).all()), float(t[torch.isfinite(t)].abs().max())) for t in ts], flush=True)
for it in range(4):
    mS = m._forward_packed(); chk(f"it{it} fwd", mS[0], mS[1], m.A, m.b)
    m.update_lagrange(mS); chk("  lagrange", m.psi_lagrange, m.lambda_lagrange)
    m.update_param(mS, lr=0.01); chk("  param", m.A, m.b)
    mS = m._forward_packed(); chk("  fwd2", mS[0], mS[1])
    print("  elbo", float(m.elbo(mS)))
