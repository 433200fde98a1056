"""Diagnostic: accuracy of the selected inverse on the ill-conditioned Matern-5/2 precision (KA5 setup, one component)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import np_kernels, np_btd
import vidp_amd
from vidp_amd import SYM, FULL, VEC, TRI

okern = np_kernels.Matern52(lengthscale=0.01, variance=0.01)
ossm = okern.state_space_model(np.linspace(0, 1, 1001))
A = ossm.A.astype(np.longdouble); cQ = ossm.cholQ.astype(np.longdouble); cP0 = ossm.cholP0.astype(np.longdouble)
P = cP0 @ cP0.T
truth = [P]
for k in range(A.shape[0]):
    P = A[k] @ P @ A[k].T + cQ[k] @ cQ[k].T
    truth.append(P)
truth = np.array(truth).astype(np.float64)
od, os_ = ossm.precision()
Ld, Ls = np_btd.cholesky(od, os_)
Sd, Ss = np_btd.inverse_blocks(Ld, Ls)
sc = np.sqrt(np.einsum("tii->ti", truth))
def scaled(a, b):
    return float(np.max(np.abs(a - b) / (sc[:, :, None] * sc[:, None, :])))
def viol(a, b):
    return float(np.max(np.abs(a - b) - (1e-6 + 1e-7 * np.abs(b))))
print("oracle selinv vs truth: scaled err %.2e viol %.2e" % (scaled(Sd, truth), viol(Sd, truth)))
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
for R0 in (0, 1001, 8, 64):
    plan = vidp_amd.Plan(1, 1001, 3, R0=R0)
    f = plan.factor(plan.pack(SYM, dev(od[None])), plan.pack(FULL, dev(os_[None])))
    s = plan.selinv(f["L"], f["G"])
    Sg = plan.unpack(SYM, s["Sig"]).cpu().numpy()[0]
    Lg = plan.unpack(TRI, f["L"]).cpu().numpy()[0]
    # oracle selected inverse from the GPU factor
    Gg = plan.unpack(FULL, f["G"], 1000).cpu().numpy()[0]
    Sd2, _ = np_btd.inverse_blocks(Lg, Gg)
    print("R0 %d levels %d: gpu selinv vs truth scaled %.2e viol %.2e | vs oracle scaled %.2e | oracle-selinv(gpu factor) vs truth scaled %.2e | L relerr %.2e"
          % (R0, plan.nlevels, scaled(Sg, truth), viol(Sg, truth), scaled(Sg, Sd), scaled(Sd2, truth), np.max(np.abs(Lg - Ld)) / np.max(np.abs(Ld))))
