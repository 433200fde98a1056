"""
On-disk formats of the reference's diffusion-process experiments (docs/diffusion_processes/), so that its shipped data and
checkpoints run against this backend unchanged:

* experiment data `.npz` -- `load_exp_data` (exp_dp_utils.py:108-125): keys Q, x0, sigma, latent_process, observation_grid,
  observations, test_grid, test_observations, time_grid;
* CVI-DP checkpoint `cvi_model.npz` (cvi_dp.py:144-149): data_sites_nat1, data_sites_nat2, girsanov_sites_nat1,
  girsanov_sites_nat2_diag, girsanov_sites_nat2_subdiag; `posteriors.npz` (cvi_dp.py:140): cvi_m, cvi_S, time_grid;
* VDP checkpoint `vi_gp_model.npz` (vi_markov_gp.py:175-178): A, b, lambda_lagrange, psi_lagrange, x0_m, x0_S;
  `posteriors.npz` with vi_m, vi_S, time_grid;
* the CVI-DP -> VDP warm start (vi_markov_gp.py:89-115): A = -(A_ssm - I)/dt, b = b_ssm/dt from the CVI posterior (LinearDrift.set_from_ssm,
  sde/drift.py:39-62).

Arrays of a single trajectory have the reference's shapes; models holding B > 1 trajectories add a leading axis.
"""
import os

import numpy as np
import torch

from ._lib import FULL, SYM, VEC


def load_exp_data(data_path, device="cuda"):
    """(Q, x0, noise_stddev [1, 1], latent_process, (observation_grid, observations), time_grid, (test_grid, test_observations))."""
    data = np.load(data_path)
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=device)
    return (data["Q"], data["x0"], np.asarray(data["sigma"]).reshape((1, 1)), data["latent_process"],
            (t(data["observation_grid"]), t(data["observations"])), t(data["time_grid"]),
            (t(data["test_grid"]), t(data["test_observations"])))


def save_exp_data(data_path, Q, x0, sigma, latent_process, observation_grid, observations, test_grid, test_observations, time_grid):
    """Write an experiment file with the reference's key names (used to synthesise data sets in its format)."""
    np.savez(data_path, Q=Q, x0=x0, sigma=sigma, latent_process=latent_process, observation_grid=observation_grid,
             observations=observations, test_grid=test_grid, test_observations=test_observations, time_grid=time_grid)


def _squeeze_b(x, B):
    x = x.detach().cpu().numpy()
    return x[0] if B == 1 else x


def _batched(a, B, ndim_single):
    a = np.asarray(a, dtype=np.float64)
    return a[None] if a.ndim == ndim_single else a


# ---- CVI-DP (CVISitesSSM / CVISitesSDE) ---------------------------------------------------------------------------------------------
def cvi_model_arrays(model):
    pl, B, n, d = model.plan, model.B, model.n_obs, model.state_dim
    g = model.girsanov_sites
    return dict(data_sites_nat1=_squeeze_b(model.data_nat1.reshape(B, n, d), B),
                data_sites_nat2=_squeeze_b(model.data_nat2.reshape(B, n, d, d), B),
                girsanov_sites_nat1=_squeeze_b(pl.unpack(VEC, g.lin), B),
                girsanov_sites_nat2_diag=_squeeze_b(pl.unpack(SYM, g.diag), B),
                girsanov_sites_nat2_subdiag=_squeeze_b(pl.unpack(FULL, g.sub, model.T - 1), B))


def save_cvi_model(output_dir, model, time_grid=None):
    """cvi_model.npz (+ posteriors.npz with the current posterior path) in `output_dir`."""
    os.makedirs(output_dir, exist_ok=True)
    np.savez(os.path.join(output_dir, "cvi_model.npz"), **cvi_model_arrays(model))
    m, S = model.dist_q.marginals
    tg = model.time_grid if time_grid is None else time_grid
    np.savez(os.path.join(output_dir, "posteriors.npz"), cvi_m=_squeeze_b(m.reshape(model.B, model.T, -1), model.B),
             cvi_S=_squeeze_b(S.reshape(model.B, model.T, model.state_dim, model.state_dim), model.B),
             time_grid=np.asarray(tg.detach().cpu() if torch.is_tensor(tg) else tg))


def load_cvi_model(path, model):
    """Assign the sites of a `cvi_model.npz` (file or directory) to `model` (same grid, observations and prior)."""
    from .variational_cvi_sde import PackedBTDNat
    if os.path.isdir(path):
        path = os.path.join(path, "cvi_model.npz")
    z = np.load(path)
    pl, B, n, d, T = model.plan, model.B, model.n_obs, model.state_dim, model.T
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(model.device)
    n1, n2 = _batched(z["data_sites_nat1"], B, 2), _batched(z["data_sites_nat2"], B, 3)
    if n1.shape != (B, n, d) or n2.shape != (B, n, d, d):
        raise ValueError(f"data sites of shape {n1.shape}, {n2.shape} do not fit a model with B={B}, n_obs={n}, d={d}")
    g1 = _batched(z["girsanov_sites_nat1"], B, 2)
    gd, gs = _batched(z["girsanov_sites_nat2_diag"], B, 3), _batched(z["girsanov_sites_nat2_subdiag"], B, 3)
    if g1.shape != (B, T, d) or gd.shape != (B, T, d, d) or gs.shape != (B, T - 1, d, d):
        raise ValueError("Girsanov sites do not fit the model's time grid")
    model.data_nat1, model.data_nat2 = dev(n1.reshape(B * n, d)), dev(n2.reshape(B * n, d, d))
    g = PackedBTDNat(pl.pack(VEC, dev(g1)), pl.pack(SYM, dev(gd)), pl.pack(FULL, dev(gs)))
    model._rebuild_theta_q(g)
    model._started, model._q = True, None
    return model


# ---- VDP (VariationalMarkovGP) ------------------------------------------------------------------------------------------------------
def vi_gp_model_arrays(model):
    """Arrays with the reference's shapes: one row per transition ([T-1, ...]), vi_sde.py:89-106."""
    pl, B, T = model.plan, model.B, model.plan.T
    q0_S = model.q0_chol @ model.q0_chol.transpose(-1, -2)
    return dict(A=_squeeze_b(pl.unpack(FULL, model.A, T - 1), B), b=_squeeze_b(pl.unpack(VEC, model.b)[:, :T - 1], B),
                lambda_lagrange=_squeeze_b(pl.unpack(VEC, model.lambda_lagrange)[:, :T - 1], B),
                psi_lagrange=_squeeze_b(pl.unpack(FULL, model.psi_lagrange, T - 1), B),
                x0_m=_squeeze_b(model.q0_mu, B), x0_S=_squeeze_b(q0_S, B))


def save_vi_gp_model(output_dir, model):
    os.makedirs(output_dir, exist_ok=True)
    np.savez(os.path.join(output_dir, "vi_gp_model.npz"), **vi_gp_model_arrays(model))


def load_vi_gp_model(path, model, restore_lagrange=True, restore_initial_state=True):
    """vi_markov_gp.py:111-115 assigns A, b and lambda_lagrange; psi_lagrange and q(x0) are restored too when present."""
    from . import linalg
    if os.path.isdir(path):
        path = os.path.join(path, "vi_gp_model.npz")
    z = np.load(path)
    pl, B, T, d = model.plan, model.B, model.plan.T, model.state_dim
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(model.device)
    A, b = _batched(z["A"], B, 3), _batched(z["b"], B, 2)
    if A.shape != (B, T - 1, d, d) or b.shape != (B, T - 1, d):
        raise ValueError(f"A {A.shape} / b {b.shape} do not fit a model with B={B}, {T - 1} transitions, d={d}")

    def put(kind, arr, out):       # one row per transition; the last node of the packed array is unused
        pad = np.zeros((B, 1) + arr.shape[2:])
        pl.pack(kind, dev(np.concatenate([arr, pad], axis=1)), out=out)
    put(FULL, A, model.A)
    put(VEC, b, model.b)
    model._param_version = getattr(model, "_param_version", 0) + 1      # cached by-products of the old (A, b) are stale
    if restore_lagrange:
        put(VEC, _batched(z["lambda_lagrange"], B, 2), model.lambda_lagrange)
        if "psi_lagrange" in z.files:
            put(FULL, _batched(z["psi_lagrange"], B, 3), model.psi_lagrange)
    if restore_initial_state and "x0_m" in z.files and "x0_S" in z.files:
        model.q0_mu = dev(_batched(z["x0_m"], B, 1)).contiguous()
        model.q0_chol = linalg.cholesky(dev(_batched(z["x0_S"], B, 2)).contiguous())
    return model


def warm_start_vdp_from_cvi(vdp_model, cvi_model):
    """A = -(A_ssm - I)/dt, b = b_ssm/dt from the CVI-DP posterior SSM (vi_markov_gp.py:104-109, sde/drift.py:55-62)."""
    ssm = cvi_model.dist_q
    pl, B, T, d = vdp_model.plan, vdp_model.B, vdp_model.plan.T, vdp_model.state_dim
    dt = float(cvi_model.time_grid[1] - cvi_model.time_grid[0])
    At = ssm.state_transitions.reshape(B, T - 1, d, d)
    bt = ssm.state_offsets.reshape(B, T - 1, d)
    eye = torch.eye(d, dtype=At.dtype, device=At.device)
    pad_m = torch.zeros((B, 1, d, d), dtype=At.dtype, device=At.device)
    pad_v = torch.zeros((B, 1, d), dtype=At.dtype, device=At.device)
    pl.pack(FULL, torch.cat([-(At - eye) / dt, pad_m], dim=1).contiguous(), out=vdp_model.A)
    pl.pack(VEC, torch.cat([bt / dt, pad_v], dim=1).contiguous(), out=vdp_model.b)
    vdp_model._param_version = getattr(vdp_model, "_param_version", 0) + 1
    return vdp_model
