import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import vidp_amd
from vidp_amd import kernels as K, tape
from vidp_amd.likelihoods import Gaussian
from vidp_amd.ssm_natgrad import GaussMarkovELBO
from vidp_amd.state_space_model import StateSpaceModel
from tests.helpers import random_ssm_params
rng = np.random.default_rng(71892305)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
T, noise = 10, 0.3
gk = K.Sum([K.Matern32(1.1, 0.7), K.Matern12(0.5, 1.2)])
d = gk.state_dim
t = np.sort(rng.uniform(0, 4, size=T)); y = np.cos(3 * t)[:, None] + 0.1 * rng.normal(size=(T, 1))
p = gk.state_space_model(dev(t)); em = gk.generate_emission_model(dev(t)); lik = Gaussian(noise)
q = StateSpaceModel(*[dev(a) for a in random_ssm_params(rng, (), T, d)], plan=p.plan)
H, yy = em.emission_matrix, dev(y)
def neg_elbo(qt):
    mu, cov = qt.marginals
    fm = torch.einsum("toi,bti->bto", H, mu); fv = torch.einsum("toi,btij,toj->bto", H, cov, H)
    return -(lik.variational_expectations(fm, fv, yy).sum() - qt.kl_divergence(p).sum())
closed = GaussMarkovELBO(p, em, lik, yy)
(cl, cd, cs), _ = closed.grad_wrt_expectations(q)
pl = q.plan
ref = [pl.unpack(vidp_amd.VEC, cl), pl.unpack(vidp_amd.SYM, cd), pl.unpack(vidp_amd.FULL, cs, T - 1)]
for rs in (3e-3, 1e-3, 3e-4, 1e-4, 3e-5):
    tape.NaturalsToExpectations.rel_step = rs
    _, g = tape.natgrad_wrt_expectations(neg_elbo, q)
    print(rs, [float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, ref)])
