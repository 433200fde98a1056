"""Accuracy of the two factor forms of the wide sweeps against the NumPy oracle (development aid)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidp_amd as amd  # noqa: E402
from oracle import np_btd, np_ssm  # noqa: E402
from tests.helpers import random_ssm_params  # noqa: E402

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
host = lambda t: t.detach().cpu().numpy()


def main():
    rng = np.random.default_rng(3)
    for d, T in ((14, 40), (16, 200), (30, 25), (30, 200)):
        prm = random_ssm_params(rng, (2,), T, d)
        o = np_ssm.StateSpaceModel(*prm)
        diag, sub = o.precision()
        Ld, Ls = np_btd.cholesky(diag, sub)
        Sd, Ss = np_btd.inverse_blocks(Ld, Ls)
        cond = max(np.linalg.cond(np_btd.to_dense(diag[b], sub[b])) for b in range(2))
        r = rng.normal(size=(2, T, d))
        x = np_btd.solve(Ld, Ls, np_btd.solve(Ld, Ls, r), transpose_left=True)
        for R0 in (0, 8):
            plan = amd.Plan(2, T, d, R0=R0)
            Dp, Sp, rp = plan.pack(amd.SYM, dev(diag)), plan.pack(amd.FULL, dev(sub)), plan.pack(amd.VEC, dev(r))
            for mo in (False, True):
                f = plan.factor(Dp, Sp, rp, moments_only=mo)
                s = plan.selinv(f["L"], f["G"], f["y"], form=f["form"])
                plan.check_info()
                eS = np.abs(host(plan.unpack(amd.SYM, s["Sig"])) - Sd).max() / np.abs(Sd).max()
                eU = np.abs(host(plan.unpack(amd.FULL, s["Sub"], T - 1)) - Ss).max() / np.abs(Ss).max()
                ex = np.abs(host(plan.unpack(amd.VEC, s["x"])) - x).max() / np.abs(x).max()
                el = np.abs(host(f["logdet"]) - np_btd.abs_log_det(Ld)).max()
                print(f"d={d} T={T} R0={R0} cond={cond:.1e} form={f['form']}: Sig {eS:.1e} Sub {eU:.1e} x {ex:.1e} logdet {el:.1e}")


if __name__ == "__main__":
    main()
