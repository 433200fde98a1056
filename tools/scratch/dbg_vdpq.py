import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import vidp_amd
from oracle import np_sde, np_models
from vidp_amd import sde as gsde
from vidp_amd.likelihoods import MultivariateGaussian
from vidp_amd.vi_sde import VariationalMarkovGPQuadrature
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
T, dt, d = 30, 0.02, 2
grid = np.arange(T) * dt
q = 0.5 * np.eye(2)
o_sde, g_sde = np_sde.VanderPolSDE(1.3, 0.9, q), gsde.VanderPolOscillatorSDE(1.3, 0.9, torch.from_numpy(q))
idx = np.arange(4, T - 1, 6)
y = rng.normal(size=(1, len(idx), d))
cholR = 0.5 * np.eye(d)
init = (np.zeros(d), 0.6 * np.eye(d))
g = VariationalMarkovGPQuadrature((grid[idx], dev(y)), g_sde, grid, MultivariateGaussian(dev(cholR)), prior_initial_state=init)
o = np_models.VariationalMarkovGP(idx, y[0], o_sde, grid, np_models.MultivariateGaussianLik(cholR), *init)
mS = g._forward_packed(); m, S = o.forward_pass()
g.update_lagrange(mS); o.update_lagrange(m, S)
g.update_param(mS, lr=0.1); o.update_param(m, S, 0.1)
mS = g._forward_packed(); m, S = o.forward_pass()
gm, gS = g._natural(mS)
print("marg", np.abs(gm.cpu().numpy()[0] - m).max(), np.abs(gS.cpu().numpy()[0] - S).max())
print("esde", float(g.E_sde(mS)[0]), o.E_sde(), o.E_sde(m[:-1], S[:-1]))
print("kl0", float(g.KL_initial_state()[0]), o.KL_initial_state())
mu = g.plan.gather_nodes(vidp_amd.VEC, mS[0], g.obs_node_ids); cov = g.plan.gather_nodes(vidp_amd.SYM, mS[1], g.obs_node_ids)
print("eobs", float(g.likelihood.variational_expectations(mu, cov, g.observations.reshape(-1, d)).sum()),
      np.sum(o.lik.variational_expectations(m[o.obs_index], S[o.obs_index], o.y)))
print("elbo", float(g.elbo()), o.elbo())
print("---- loop")
for it in range(1, 4):
    mS = g._forward_packed(); m, S = o.forward_pass()
    g.update_lagrange(mS); o.update_lagrange(m, S)
    g.update_param(mS, lr=0.1); o.update_param(m, S, 0.1)
    if it > 1:
        g.update_initial_statistics(0.1); o.update_initial_statistics(0.1)
    mS2 = g._forward_packed(); m2, S2 = o.forward_pass()
    gm, gS = g._natural(mS2)
    print(it, "marg", np.abs(gm.cpu().numpy()[0] - m2).max(), "esde", float(g.E_sde(mS2)[0]) - o.E_sde(), "kl0", float(g.KL_initial_state()[0]), o.KL_initial_state(),
          "q0", np.abs(g.q0_mu.cpu().numpy()[0] - o.q0_mu).max(), np.abs(g.q0_chol.cpu().numpy()[0] - o.q0_chol).max(), "elbo", float(g.elbo()) - o.elbo())
