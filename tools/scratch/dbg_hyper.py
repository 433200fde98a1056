import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import vidp_amd
from vidp_amd import kernels as K, tape
from vidp_amd.likelihoods import Gaussian
from vidp_amd.variational_cvi import CVIGaussianProcess
rng = np.random.default_rng(71892305)
t = np.sort(rng.uniform(0, 5, size=10)); y = np.sin(2 * t)[:, None] + 0.1 * rng.normal(size=(10, 1)); noise = 0.4
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
g = CVIGaussianProcess((dev(t), dev(y)), K.Matern32(1.1, 0.7), Gaussian(noise), learning_rate=1.0)
g.update_sites()
def val(l, v):
    lv = {"lengthscale": torch.tensor(l, dtype=torch.float64, device="cuda", requires_grad=True),
          "variance": torch.tensor(v, dtype=torch.float64, device="cuda", requires_grad=True)}
    e, lv = g.classic_elbo_tape_hyper(lv)
    return e, lv
e0, lv = val(1.1, 0.7)
gr = torch.autograd.grad(e0, [lv["lengthscale"], lv["variance"]])
print("autograd", [float(x) for x in gr])
h = 1e-3
fd = lambda f: (8 * (f(h) - f(-h)) - (f(2 * h) - f(-2 * h))) / (12 * h)
print("fd of tape forward", fd(lambda e: float(val(1.1 + e, 0.7)[0])), fd(lambda e: float(val(1.1, 0.7 + e)[0])))
# pieces: gradient of each term
e0, lv = val(1.1, 0.7)
p, _ = g._kernel.differentiable_ssm(g._time_points, lv, plan=g.dist_p.plan)
for name, fn in (("logdet_p", lambda p: p.log_det_precision().sum()), ("nat_sum", lambda p: sum((x * x).sum() for x in p.naturals())),
                 ("pmean", lambda p: p.marginal_means.sum())):
    v = fn(p)
    ga = torch.autograd.grad(v, [lv["lengthscale"]], retain_graph=True, allow_unused=True)[0]
    def f(e):
        lv2 = {"lengthscale": torch.tensor(1.1 + e, dtype=torch.float64, device="cuda"), "variance": torch.tensor(0.7, dtype=torch.float64, device="cuda")}
        p2, _ = g._kernel.differentiable_ssm(g._time_points, lv2, plan=g.dist_p.plan)
        return float(fn(p2))
    print(name, None if ga is None else float(ga), fd(f))
