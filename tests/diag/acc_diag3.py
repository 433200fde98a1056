"""Diagnostic: is the round-trip sensitivity caused by the GPU factor or by the GPU selected inverse?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels, np_btd, np_transforms
import vidp_amd
from vidp_amd import SYM, FULL, VEC, TRI

def viol(a, b):
    return float(np.max(np.abs(a - b) - (1e-6 + 1e-7 * np.abs(b))))
okern = np_kernels.Matern52(lengthscale=0.01, variance=0.01)
ossm = okern.state_space_model(np.linspace(0, 1, 1001))
od, os_ = ossm.precision()
Ld, Ls = np_btd.cholesky(od, os_)
mu = np.zeros((1001, 3))
def round_trip(Sd, Ss):
    eta_d = Sd + mu[:, :, None] * mu[:, None, :]
    eta_s = Ss + mu[1:, :, None] * mu[:-1, None, :]
    back = np_transforms.expectations_to_ssm_params(mu, eta_d, eta_s)
    return viol(back[3], ossm.cholQ), viol(back[0], ossm.A)
print("oracle factor + oracle selinv:", round_trip(*np_btd.inverse_blocks(Ld, Ls)))
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
for R0 in (1001, 0):
    plan = vidp_amd.Plan(1, 1001, 3, R0=R0)
    f = plan.factor(plan.pack(SYM, dev(od[None])), plan.pack(FULL, dev(os_[None])))
    Lg = plan.unpack(TRI, f["L"]).cpu().numpy()[0]
    Gg = plan.unpack(FULL, f["G"], 1000).cpu().numpy()[0]
    print("R0", R0, "gpu factor + oracle selinv:", round_trip(*np_btd.inverse_blocks(Lg, Gg)))
    s = plan.selinv(f["L"], f["G"])
    print("R0", R0, "gpu factor + gpu selinv:", round_trip(plan.unpack(SYM, s["Sig"]).cpu().numpy()[0], plan.unpack(FULL, s["Sub"], 1000).cpu().numpy()[0]))
    s = plan.selinv(plan.pack(TRI, dev(Ld[None])), plan.pack(FULL, dev(Ls[None])))
    print("R0", R0, "oracle factor + gpu selinv:", round_trip(plan.unpack(SYM, s["Sig"]).cpu().numpy()[0], plan.unpack(FULL, s["Sub"], 1000).cpu().numpy()[0]))
