import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench, vidp_amd
from vidp_amd import sde as gsde
from vidp_amd.likelihoods import MultivariateGaussian
from vidp_amd.variational_cvi_sde import CVISitesSDE
B, T, d, dt, noise = 64, 100000, 6, 0.01, 0.1
idx, ys = bench.synth_double_well(B, T, d, dt, 50, noise, seed=5)
grid = np.arange(T) * dt
dev = torch.device("cuda", 0)
lik = MultivariateGaussian(torch.from_numpy(bench.obs_chol(d, noise)).to(dev))
for rows in ("16", "0"):
    os.environ["MFGM_COARSE_ROWS"] = rows
    plan = vidp_amd.Plan(B, T, d)
    m = CVISitesSDE(gsde.DoubleWellSDE(q=torch.eye(d, dtype=torch.float64)), grid, (grid[idx], torch.from_numpy(ys).to(dev)), lik,
                    prior_initial_state=(np.zeros(d), np.eye(d)), plan=plan)
    for _ in range(3):
        m.update_data_sites(0.5); m.update_girsanov_sites(0.1)
    torch.cuda.synchronize()
    pass

    print("done rows", rows)
    del m, plan
