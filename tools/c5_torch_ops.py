"""Which torch operators (and from where) run inside the config-5 step: torch.profiler with stacks around update_sites + classic_elbo."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import vidp_amd
from vidp_amd import kernels as K
from vidp_amd.likelihoods import Gaussian
from vidp_amd.sparse_variational_cvi import SparseCVIGaussianProcess

M = 200000
N, span = 2 * M, 0.1 * M
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
z = torch.linspace(0, span, M, dtype=torch.float64, device=dev)
t = torch.from_numpy(np.sort(rng.uniform(0, span, size=N))).to(dev)
y = (torch.sin(3 * t) + 0.1 * torch.from_numpy(rng.normal(size=N)).to(dev))[:, None]
m = SparseCVIGaussianProcess(bench.sum16_kernel(K), z, Gaussian(0.01), learning_rate=0.5)
for _ in range(3):
    m.update_sites((t, y)); e = m.classic_elbo((t, y))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        m.update_sites((t, y)); e = m.classic_elbo((t, y))
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.device_time_total > 0 and not any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        st = [s for s in ev.stack if "vi-diffusion" in s or "bench" in s][:2]
        rows.append((ev.name, str(ev.input_shapes)[:60], round(ev.device_time_total, 1), " <- ".join(x.split("/")[-1] for x in st)))
for r in rows[: len(rows) // 2]:
    print(r)
