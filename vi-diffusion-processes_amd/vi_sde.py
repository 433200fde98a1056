"""
Host-side mirror of markovflow/models/vi_sde.py: `VariationalMarkovGP` (VDP, Archambeau et al. 2007) --
`forward_pass`, `update_lagrange`, `update_param`, `update_initial_statistics`, `E_sde`, `KL_initial_state`, `elbo`.

A leading batch of B independent trajectories is supported (observations [B, n_obs, d]); all per-time-step state
(A, b, psi, lambda, marginals) lives in the packed device layout.  The reference's Python loop over T in
update_lagrange (with an O(T) tensor copy per step) is a partitioned affine recurrence here.
Prior / initial distributions are (mean [d], covariance [d, d]) pairs.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib, linalg
from ._lib import FULL, SYM, TRI, VEC
from .packed import Plan, _ptr, _stream, aligned_segment_length, observation_period
from .variational_cvi_sde import grid_indices


class VariationalMarkovGP:
    """vi_sde.py:63-482."""

    def __init__(self, input_data, prior_sde, grid, likelihood, prior_initial_state=None, stabilize_system=False, plan=None):
        self.stabilize_system = bool(stabilize_system)
        obs_times, observations = input_data
        if observations.dim() == 2:
            observations = observations[None]
        self.observations = observations.contiguous()
        self.B, self.n_obs, self.state_dim = observations.shape
        self.prior_sde, self.likelihood = prior_sde, likelihood
        self.grid = torch.as_tensor(grid, dtype=torch.float64)
        self.num_states = int(self.grid.numel())
        self.num_transitions = self.num_states - 1
        self.dt = float(self.grid[1] - self.grid[0])
        self.device = observations.device
        d = self.state_dim
        if plan is None:
            # segments aligned with an equally spaced observation grid (packed.aligned_segment_length)
            r0 = aligned_segment_length(self.B, self.num_states, d, observation_period(grid_indices(self.grid, obs_times)))
            plan = Plan(self.B, self.num_states, d, R0=r0, device=self.device)
        self.plan = plan
        pl = self.plan
        self.lib = pl.lib
        if prior_initial_state is None:
            # vi_sde.py:93-96: N(0, q * I)
            prior_initial_state = (np.zeros(d), prior_sde.q.cpu().numpy() * np.eye(d))
        self.p0_mu = np.asarray(prior_initial_state[0], dtype=np.float64).reshape(d)
        self.p0_cov = np.asarray(prior_initial_state[1], dtype=np.float64).reshape(d, d)
        # q(x0) starts at the prior (vi_sde.py:98-100); per trajectory
        self.q0_mu = torch.from_numpy(self.p0_mu).to(self.device).expand(self.B, d).contiguous()
        self.q0_chol = torch.from_numpy(np.linalg.cholesky(self.p0_cov)).to(self.device).expand(self.B, d, d).contiguous()
        self.A, self.b = pl.zeros(FULL), pl.zeros(VEC)
        self.lambda_lagrange = pl.zeros(VEC)
        self.psi_lagrange = pl.zeros(FULL)
        self._reset_lagrange()
        self.obs_index = grid_indices(self.grid, obs_times).to(self.device)
        self.obs_node_ids = pl.node_ids(self.obs_index)
        # jump-condition constants of the Gaussian likelihood: yR = R^{-1} y, dobsS = -1/2 R^{-1} at the observation nodes
        Rinv = likelihood.inv_covariance
        n = self.B * self.n_obs
        self._yR, self._dobsS = pl.zeros(VEC), pl.zeros(SYM)
        pl.scatter_nodes(VEC, self._yR, self.obs_node_ids, (self.observations.reshape(n, d) @ Rinv), accumulate=True)
        pl.scatter_nodes(SYM, self._dobsS, self.obs_node_ids, (-0.5 * Rinv).expand(n, d, d).contiguous(), accumulate=True)
        # every observation contributes the same block -1/2 R^{-1}: the sweeps read one count per node (packed order [tile][step][64])
        # and that block instead of the dense array, which is zero at all but the observation nodes
        ids = self.obs_node_ids
        lane = (ids // self.num_states) * pl.P + (ids % self.num_states) // pl.R
        slot = ((lane // 64) * pl.R + (ids % self.num_states) % pl.R) * 64 + lane % 64
        self._obs_count = torch.zeros(pl.Lpad * pl.R, dtype=torch.int32, device=self.device)
        self._obs_count.index_add_(0, slot, torch.ones_like(slot, dtype=torch.int32))
        il = torch.tril_indices(d, d, device=self.device)
        self._dobs_const = (-0.5 * Rinv)[il[0], il[1]].contiguous()
        self._seg = torch.empty(self.lib.mfgm_vdp_workspace_doubles(pl.h), dtype=torch.float64, device=self.device)
        af, bf = prior_sde.drift_cubic()
        self._prm = _lib.VdpParams()
        for i in range(d):
            self._prm.af[i], self._prm.bf[i], self._prm.q[i] = af, bf, prior_sde.q_diag[i]
        self._prm.dt = self.dt
        self._prm.clip = 5000.0 if self.stabilize_system else 0.0        # CLIP_MAX, vi_sde.py:59-60
        self._ssm_bufs = None
        self.dist_q_ssm = None
        self._param_version = 0        # bumped by every method that changes (A, b) or the drift parameters (see E_sde)

    def _reset_lagrange(self):
        """lambda = 0, psi = 1e-10 I on every transition (vi_sde.py:102-103, 328-329)."""
        self.lambda_lagrange.zero_()
        d = self.state_dim
        eye = (1e-10 * torch.eye(d, dtype=torch.float64, device=self.device)).expand(self.B, self.num_states, d, d).contiguous()
        self.plan.pack(FULL, eye, out=self.psi_lagrange)

    def _params(self, b_index=None, lr=0.0):
        """Parameter block; q(x0) differs per trajectory only through update_initial_statistics, which is applied per chain
        on the host side by writing node 0 of the SSM arrays."""
        self._prm.lr = float(lr)
        return self._prm

    # -- forward_pass ------------------------------------------------------------------------------------------
    @property
    def forward_pass(self):
        """Marginal means / covariances of the SSM of the current linear drift (vi_sde.py:171-204), natural tensors [B, T, ...]."""
        m, S = self._forward_packed()
        return self.plan.unpack(VEC, m), self.plan.unpack(SYM, S)

    def _forward_packed(self):
        pl = self.plan
        prm = self._params()
        if pl.d <= 8 and pl.T > 1:
            if self.forward_mode == "moments":
                return self._forward_packed_moments(prm)
            if self.forward_mode == "precision":
                return self._forward_packed_direct(prm)
        if self._ssm_bufs is None:
            self._ssm_bufs = (pl.empty(FULL), pl.empty(VEC), pl.empty(TRI))
        A, off, chol = self._ssm_bufs
        # stabilize_system (vi_sde.py:186-200): NaN -> 1e-8 and clipping of the state transitions and offsets to [-1, 1] happen
        # inside the kernel (prm.clip > 0)
        _lib.check(self.lib.mfgm_packed_vdp_to_ssm(pl.h, ctypes.byref(prm), _ptr(self.A), _ptr(self.b), _ptr(A), _ptr(off), _ptr(chol),
                                                   _stream()), "mfgm_packed_vdp_to_ssm")
        # node 0 carries q(x0), which is per trajectory
        node0 = pl.node_ids(torch.zeros(1, dtype=torch.int64))
        pl.scatter_nodes(VEC, off, node0, self.q0_mu)
        pl.scatter_nodes(TRI, chol, node0, self.q0_chol)
        pr = pl.ssm_to_naturals(A, off, chol, precision=True)
        f = pl.factor(pr["diag"], pr["sub"], pr["lin"], want_logdet=False)
        s = pl.selinv(f["L"], f["G"], f["y"], want_sub=False)
        self._mS = (s["x"], s["Sig"])
        return self._mS

    # how forward_pass obtains the marginals (d <= 8): "moments" = the partitioned moment recursion (mfgm_packed_vdp_marginals, what
    # the reference's forward_pass computes); "precision" = precision blocks + factorisation + selected inverse; "ssm" = the same over
    # explicit SSM arrays.  VIDP_VDP_FORWARD selects; the three agree to rounding (tests/test_gpu_api.py).
    forward_mode = os.environ.get("VIDP_VDP_FORWARD", "moments")
    # the moment recursion also makes the first passes of the Lagrange sweep that follows (VIDP_VDP_FUSE_LAGRANGE: 0 none, 1 the A-only
    # products pass, 2 the offsets pass too -- inside the final sweep for d <= 5, as a launch of that call for d >= 6)
    fuse_lagrange = int(os.environ.get("VIDP_VDP_FUSE_LAGRANGE", "2"))

    def _forward_packed_moments(self, prm):
        pl, d = self.plan, self.state_dim
        if getattr(self, "_q0_key", None) is None or self._q0_key[0] is not self.q0_mu or self._q0_key[1] is not self.q0_chol:
            cov = self.q0_chol @ self.q0_chol.transpose(-1, -2)
            il = torch.tril_indices(d, d, device=self.device)
            self._q0_dev = (self.q0_mu.contiguous(), cov[:, il[0], il[1]].contiguous())
            self._q0_key = (self.q0_mu, self.q0_chol)
        mu, Sig = pl.empty(VEC), pl.empty(SYM)        # fresh outputs: callers keep the marginals of earlier passes
        e = torch.empty(self.B, dtype=torch.float64, device=self.device)
        if self.fuse_lagrange:
            # the sweeps also make the first passes of the Lagrange call that follows (valid while (A, b) and these marginals stand):
            # level 1 the A-only products of its segment maps, level 2 the affine offsets as well
            if getattr(self, "_lseg", None) is None:
                self._lseg = torch.empty_like(self._seg)
            jump = (_ptr(self._yR), _ptr(self._dobsS), *self._jump_args()) if self.fuse_lagrange >= 2 else (None, None, None, None)
            _lib.check(self.lib.mfgm_packed_vdp_marginals_products(pl.h, ctypes.byref(prm), _ptr(self.A), _ptr(self.b),
                                                                   _ptr(self._q0_dev[0]), _ptr(self._q0_dev[1]), _ptr(mu), _ptr(Sig),
                                                                   _ptr(e), _ptr(self._seg), _ptr(self._lseg), *jump, _ptr(pl.ws),
                                                                   _stream()), "mfgm_packed_vdp_marginals_products")
            self._products_of = (self.A, self._param_version, mu if self.fuse_lagrange >= 2 else None, bool(prm.clip > 0.0))
        else:
            _lib.check(self.lib.mfgm_packed_vdp_marginals(pl.h, ctypes.byref(prm), _ptr(self.A), _ptr(self.b), _ptr(self._q0_dev[0]),
                                                          _ptr(self._q0_dev[1]), _ptr(mu), _ptr(Sig), _ptr(e), _ptr(self._seg),
                                                          _ptr(pl.ws), _stream()), "mfgm_packed_vdp_marginals")
        self._mS = (mu, Sig)
        # E_sde of exactly these marginals under the current (A, b), a by-product of the final sweep; valid until the variational or
        # drift parameters change (self._esde_of is compared by identity in E_sde)
        self._esde_of = (mu, Sig, e * self.dt, self._param_version)
        return self._mS

    def _forward_packed_direct(self, prm):
        """
        The same marginals without materialising the SSM: (A, b) -> precision blocks in one kernel (mfgm_packed_vdp_to_naturals),
        a factorisation that does not store L_{t+1,t}, and the selected inverse rebuilt from the sub-diagonal precision blocks.
        """
        pl, d = self.plan, self.state_dim
        key = (self.q0_mu, self.q0_chol)
        if getattr(self, "_p0_key", None) is None or self._p0_key[0] is not key[0] or self._p0_key[1] is not key[1]:
            p0inv = linalg.spd_inverse(chol=self.q0_chol)                                   # [B, d, d]
            il = torch.tril_indices(d, d, device=self.device)
            self._p0 = (p0inv[:, il[0], il[1]].contiguous(), (p0inv @ self.q0_mu[..., None])[..., 0].contiguous())
            self._p0_key = key
        if getattr(self, "_fw", None) is None:
            self._fw = dict(nat=(pl.empty(VEC), pl.empty(SYM), pl.empty(FULL)), f={})
        lin, diag, sub = self._fw["nat"]
        _lib.check(self.lib.mfgm_packed_vdp_to_naturals(pl.h, ctypes.byref(prm), _ptr(self.A), _ptr(self.b), _ptr(self._p0[0]),
                                                        _ptr(self._p0[1]), _ptr(lin), _ptr(diag), _ptr(sub), _stream()),
                   "mfgm_packed_vdp_to_naturals")
        f = pl.factor(diag, sub, lin, want_logdet=False, out=self._fw["f"], store_G=False)
        self._fw["f"].update(L=f["L"], y=f["y"])
        s = pl.selinv_s(f["L"], sub, 1.0, f["y"])        # fresh outputs: callers keep the marginals of earlier passes
        self._mS = (s["x"], s["Sig"])
        return self._mS

    # -- energies ----------------------------------------------------------------------------------------------
    def E_sde(self, mS=None):
        """E_sde per trajectory [B] (vi_sde.py:422-434)."""
        pl = self.plan
        m, S = mS if mS is not None else self._forward_packed()
        c = getattr(self, "_esde_of", None)
        if c is not None and c[0] is m and c[1] is S and c[3] == self._param_version:
            return c[2]
        out = torch.empty(self.B, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mfgm_packed_vdp_esde(pl.h, ctypes.byref(self._params()), _ptr(m), _ptr(S), _ptr(self.A), _ptr(self.b),
                                                 _ptr(out), None, None, _ptr(pl.ws), _stream()), "mfgm_packed_vdp_esde")
        return out * self.dt

    def _grad_E_sde(self, mS=None):
        """(dE/dm / dt, dE/dS / dt) as natural tensors [B, T-1, ...] (vi_sde.py:206-239)."""
        pl = self.plan
        m, S = mS if mS is not None else self._forward_packed()
        out = torch.empty(self.B, dtype=torch.float64, device=self.device)
        gm, gS = pl.zeros(VEC), pl.zeros(SYM)
        _lib.check(self.lib.mfgm_packed_vdp_esde(pl.h, ctypes.byref(self._params()), _ptr(m), _ptr(S), _ptr(self.A), _ptr(self.b),
                                                 _ptr(out), _ptr(gm), _ptr(gS), _ptr(pl.ws), _stream()), "mfgm_packed_vdp_esde")
        return pl.unpack(VEC, gm, self.num_transitions), pl.unpack(SYM, gS, self.num_transitions)

    # -- prior-parameter learning (vi_sde.py:457-470) ------------------------------------------------------------------------
    def _refresh_drift_params(self):
        af, bf = self.prior_sde.drift_cubic()
        for i in range(self.state_dim):
            self._prm.af[i], self._prm.bf[i] = af, bf
        self._param_version += 1

    def set_prior_initial_state(self, mean, cov):
        self.p0_mu = np.asarray(mean, dtype=np.float64).reshape(self.state_dim)
        self.p0_cov = np.asarray(cov, dtype=np.float64).reshape(self.state_dim, self.state_dim)

    def grad_prior_sde_params(self):
        """
        d E_sde / d (trainable drift parameters) at fixed (m, S), in `prior_sde.trainable_variables` order.  As in the reference,
        the path handed to E_sde here is states 1..N (`m[1:]`, vi_sde.py:462-466) against the N transitions' (A, b).
        With r = f_p - f_q = af x - bf x^3 + A x - b per dimension:  dE/daf = dt sum w E[r x],  dE/dbf = -dt sum w E[r x^3],
        Gaussian moments to order six (E[x_j x_i^3] = m_j E[x_i^3] + 3 S_ji E[x_i^2]).
        """
        pl, N, d = self.plan, self.num_transitions, self.state_dim
        mp, Sp = self._forward_packed()
        m, S = pl.unpack(VEC, mp)[:, 1:], pl.unpack(SYM, Sp)[:, 1:]
        A, b = pl.unpack(FULL, self.A, N), pl.unpack(VEC, self.b)[:, :N]
        af, bf = self.prior_sde.drift_cubic()
        w = torch.tensor([1.0 / v for v in self.prior_sde.q_diag], dtype=torch.float64, device=self.device)
        s = torch.diagonal(S, dim1=-2, dim2=-1)
        m2, s2 = m * m, s * s
        Ex2 = m2 + s
        Ex3 = m * (m2 + 3.0 * s)
        Ex4 = m2 * m2 + 6.0 * m2 * s + 3.0 * s2
        Ex6 = m2 * m2 * m2 + 15.0 * m2 * m2 * s + 45.0 * m2 * s2 + 15.0 * s2 * s
        Am = (A @ m[..., None])[..., 0]
        ASii = (A * S.transpose(-1, -2)).sum(-1)                       # sum_j A_ij S_ji
        Erx = af * Ex2 - bf * Ex4 + ASii + Am * m - b * m
        Erx3 = af * Ex4 - bf * Ex6 + Am * Ex3 + 3.0 * ASii * Ex2 - b * Ex3
        daf = float(self.dt * (w * Erx).sum())
        dbf = float(-self.dt * (w * Erx3).sum())
        jac = self.prior_sde.drift_cubic_jacobian()
        return [daf * jac[n][0] + dbf * jac[n][1] for n in self.prior_sde.trainable_variables]

    def grad_initial_state(self):
        """
        (d KL[q(x0) || p(x0)] / d loc, d / d scale) of the prior initial state N(loc, scale scale^T), summed over trajectories
        (vi_sde.py:472-482): P^{-1}(loc - m_q) and tril(2 G scale) with G = 1/2 (P^{-1} - P^{-1}(S_q + dd^T) P^{-1}).
        """
        d = self.state_dim
        P0 = torch.from_numpy(self.p0_cov).to(self.device)
        mu0 = torch.from_numpy(self.p0_mu).to(self.device)
        Lp = linalg.cholesky(P0)
        Pinv = linalg.spd_inverse(chol=Lp)
        delta = mu0 - self.q0_mu                                            # [B, d]
        S0 = self.q0_chol @ self.q0_chol.transpose(-1, -2)
        M = S0 + delta[:, :, None] * delta[:, None, :]
        G = 0.5 * (Pinv[None] - Pinv[None] @ M @ Pinv[None]).sum(0)
        g_loc = (delta @ Pinv).sum(0)
        g_scale = torch.tril(2.0 * G @ Lp)
        return g_loc, g_scale

    # -- updates -----------------------------------------------------------------------------------------------
    # VIDP_VDP_DENSE_JUMPS=1: the sweeps read the dense d_obs_S array instead of (count per node) x (constant block)
    dense_jumps = os.environ.get("VIDP_VDP_DENSE_JUMPS", "0") == "1"

    def _jump_args(self):
        return (None, None) if self.dense_jumps else (_ptr(self._obs_count), _ptr(self._dobs_const))

    def update_lagrange(self, mS=None):
        """Backward sweep with jump conditions for (psi, lambda) (vi_sde.py:289-347)."""
        pl = self.plan
        m, S = mS if mS is not None else self._mS
        _lib.check(self.lib.mfgm_packed_vdp_lagrange(pl.h, ctypes.byref(self._params()), _ptr(m), _ptr(S), _ptr(self.A), _ptr(self.b),
                                                     _ptr(self._yR), _ptr(self._dobsS), _ptr(self.psi_lagrange),
                                                     _ptr(self.lambda_lagrange), _ptr(self._seg), *self._jump_args(), _stream()),
                   "mfgm_packed_vdp_lagrange")
        self._mult0 = None

    def update_param(self, mS=None, lr=0.1):
        """A <- (1-lr) A + lr A~, b <- (1-lr) b + lr b~ (vi_sde.py:377-414)."""
        self._param_version += 1
        pl = self.plan
        m, S = mS if mS is not None else self._mS
        # stabilize_system (vi_sde.py:393-397): psi / lambda are scrubbed and clipped in place by the kernel (prm.clip > 0)
        _lib.check(self.lib.mfgm_packed_vdp_update_param(pl.h, ctypes.byref(self._params(lr=lr)), _ptr(m), _ptr(S),
                                                         _ptr(self.psi_lagrange), _ptr(self.lambda_lagrange), _ptr(self.A),
                                                         _ptr(self.b), _stream()), "mfgm_packed_vdp_update_param")

    # update_lagrange_and_param keeps the multipliers of node 0 only (VIDP_VDP_STORE_MULTIPLIERS=1: the [B, T] arrays every time)
    store_multipliers = os.environ.get("VIDP_VDP_STORE_MULTIPLIERS", "0") == "1"

    def update_lagrange_and_param(self, mS=None, lr=0.1, store_multipliers=None):
        """
        update_lagrange(mS) followed by update_param(mS, lr), as the trainer calls them (vi_markov_gp_trainer.py:56-57), in one
        set of sweeps: the final Lagrange sweep makes the parameter update node by node (mfgm_packed_vdp_lagrange_update).
        By default the sweep keeps the multipliers of node 0 only (mfgm_packed_vdp_lagrange_update0): the fused update has consumed every
        other psi_t / lambda_t when it leaves node t, and update_initial_statistics -- the one later reader in the loop -- needs node 0;
        `psi_lagrange` / `lambda_lagrange` are then NOT current until update_lagrange (or a call with store_multipliers=True) runs.
        """
        po = getattr(self, "_products_of", None)
        have_products = po is not None and po[0] is self.A and po[1] == self._param_version
        self._products_of = None
        self._param_version += 1
        pl = self.plan
        m, S = mS if mS is not None else self._mS
        # 3: the offsets of the segment maps are there too (made on these very marginals, same stabilisation)
        mode = 0 if not have_products else (3 if (po[2] is m and po[3] == bool(self._params(lr=lr).clip > 0.0)) else 2)
        if self.store_multipliers if store_multipliers is None else store_multipliers:
            _lib.check(self.lib.mfgm_packed_vdp_lagrange_update(pl.h, ctypes.byref(self._params(lr=lr)), _ptr(m), _ptr(S), _ptr(self.A),
                                                                _ptr(self.b), _ptr(self._yR), _ptr(self._dobsS), _ptr(self.psi_lagrange),
                                                                _ptr(self.lambda_lagrange), _ptr(self._seg), *self._jump_args(), _stream()),
                       "mfgm_packed_vdp_lagrange_update")
            self._mult0 = None
            return
        if getattr(self, "_mult0_bufs", None) is None:
            d = self.state_dim
            self._mult0_bufs = (torch.empty((self.B, d, d), dtype=torch.float64, device=self.device),
                                torch.empty((self.B, d), dtype=torch.float64, device=self.device))
        psi0, lam0 = self._mult0_bufs
        _lib.check(self.lib.mfgm_packed_vdp_lagrange_update0(pl.h, ctypes.byref(self._params(lr=lr)), _ptr(m), _ptr(S), _ptr(self.A),
                                                             _ptr(self.b), _ptr(self._yR), _ptr(self._dobsS), _ptr(psi0), _ptr(lam0),
                                                             _ptr(self._lseg if have_products else self._seg), *self._jump_args(),
                                                             mode, _stream()),
                   "mfgm_packed_vdp_lagrange_update0")
        self._mult0 = (psi0, lam0)          # what update_initial_statistics reads until the arrays are current again

    def update_initial_statistics(self, lr):
        """q(x0) from the multipliers at t = 0 (vi_sde.py:241-260); tiny per-trajectory d x d algebra."""
        pl = self.plan
        if getattr(self, "_mult0", None) is not None:
            psi0, lam0 = self._mult0
        else:
            node0 = pl.node_ids(torch.zeros(1, dtype=torch.int64))
            lam0 = pl.gather_nodes(VEC, self.lambda_lagrange, node0)
            psi0 = pl.gather_nodes(FULL, self.psi_lagrange, node0)
        P0 = torch.from_numpy(self.p0_cov).to(self.device)
        mu0 = torch.from_numpy(self.p0_mu).to(self.device)
        mean = mu0 - (P0 @ lam0[..., None])[..., 0]
        # psi is not symmetric (the sweep adds psi A + psi A): a general inverse, as the reference's tf.linalg.inv (the SPD inverse used
        # until round 3 read the lower triangle only)
        cov = linalg.general_inverse(linalg.spd_inverse(P0) + 2.0 * psi0)
        q0_cov = self.q0_chol @ self.q0_chol.transpose(-1, -2)
        self.q0_mu = (1 - lr) * self.q0_mu + lr * mean
        self.q0_chol = linalg.cholesky(((1 - lr) * q0_cov + lr * cov).contiguous())

    def KL_initial_state(self):
        """KL[q(x0) || p(x0)] per trajectory (vi_sde.py:416-420)."""
        d = self.state_dim
        mu0, P0inv, ldP0 = self._prior_x0_device()
        # a function of q(x0) and p(x0) alone (~15 d x d launches): kept until either moves (update_initial_statistics assigns new tensors,
        # in-place edits move the version counters, the prior's initial state is compared by value)
        key = (self.q0_mu, self.q0_mu._version, self.q0_chol, self.q0_chol._version, mu0, P0inv)
        c = getattr(self, "_kl0", None)
        if c is not None and len(c[0]) == len(key) and all(a is b if isinstance(a, torch.Tensor) else a == b for a, b in zip(c[0], key)):
            return c[1]
        S0 = self.q0_chol @ self.q0_chol.transpose(-1, -2)
        dm = mu0 - self.q0_mu
        tr = (P0inv * S0).sum(dim=(-1, -2))
        mh = ((dm @ P0inv) * dm).sum(-1)
        ld0 = 2.0 * torch.log(torch.diagonal(self.q0_chol, dim1=-2, dim2=-1)).sum(-1)
        out = 0.5 * (tr + mh - d + ldP0 - ld0)
        self._kl0 = (key, out)
        return out

    def _prior_x0_device(self):
        """(mu0, P0^{-1}, log det P0) of the prior initial state on the device: computed once per (p0_mu, p0_cov), not on every
        ELBO evaluation (two host-to-device copies and a dozen d x d launches each time otherwise)."""
        key = (self.p0_mu.tobytes(), self.p0_cov.tobytes())
        if getattr(self, "_p0_dev", (None,))[0] != key:
            P0 = torch.from_numpy(self.p0_cov).to(self.device)
            self._p0_dev = (key, torch.from_numpy(self.p0_mu).to(self.device), linalg.spd_inverse(P0), linalg.logdet_spd(P0))
        return self._p0_dev[1:]

    def elbo_per_trajectory(self, mS=None):
        pl = self.plan
        m, S = mS if mS is not None else self._forward_packed()
        n, d = self.B * self.n_obs, self.state_dim
        lik = self.likelihood
        if pl.d <= 8 and hasattr(lik, "inv_covariance") and hasattr(lik, "log_det_chol"):
            # multivariate Gaussian likelihood: gather, arithmetic and per-trajectory sums in one launch
            cst = getattr(lik, "ve_constant", None)             # host scalar kept by the likelihood: float(device tensor) here is a sync per ELBO
            if cst is None:
                cst = -float(lik.log_det_chol) - 0.5 * lik.obs_dim * math.log(2.0 * math.pi)
            e_obs = pl.mvn_obs_ve(m, S, self.obs_node_ids, self.n_obs, self.observations.reshape(n, d).contiguous(), lik.inv_covariance, cst)
        else:
            mu = pl.gather_nodes(VEC, m, self.obs_node_ids)
            cov = pl.gather_nodes(SYM, S, self.obs_node_ids)
            e_obs = lik.variational_expectations(mu, cov, self.observations.reshape(n, d)).reshape(self.B, self.n_obs).sum(-1)
        # the reference re-runs forward_pass inside E_sde() (vi_sde.py:443): identical parameters, identical value
        return e_obs - self.E_sde((m, S)) - self.KL_initial_state()

    def elbo(self, mS=None):
        """Variational lower bound summed over trajectories (vi_sde.py:436-455)."""
        return self.elbo_per_trajectory(mS).sum()


class VariationalMarkovGPQuadrature(VariationalMarkovGP):
    """
    VDP (vi_sde.py:63-482) for the drifts the closed-form kernels do not cover -- drifts that couple the state dimensions
    (`sde.VanderPolOscillatorSDE`), the network drift (`sde.MLPDrift`), the per-dimension non-polynomial ones (through their own
    classes: not here) -- and for a full diffusion matrix: the reference takes any SDE (vi_sde.py:377-414 calls
    `expected_drift` / `expected_gradient_drift`, :422-434 the 20-point quadrature of the squared drift difference).  The local
    quantities come from the HIP quadrature kernels (csrc/mfgm_quad.h: E_sde and its gradients with respect to (m, S) and the drift
    parameters, E f / E df/dx), the Lagrange sweep from `mfgm_quad_vdp_lagrange`, the marginals of the linear drift's SSM from the
    HIP sweeps (StateSpaceModel.marginals).  State (A, b, psi, lambda) is held in the reference's natural layout: these are the small
    models of that formulation (20^d quadrature nodes per time step, d <= 3).  Same method names as VariationalMarkovGP, so that
    VIMarkovGPTrainer drives either.
    """

    def __init__(self, input_data, prior_sde, grid, likelihood, prior_initial_state=None, stabilize_system=False, plan=None):
        self.stabilize_system = bool(stabilize_system)
        obs_times, observations = input_data
        if observations.dim() == 2:
            observations = observations[None]
        self.observations = observations.contiguous()
        self.B, self.n_obs, self.state_dim = observations.shape
        self.prior_sde, self.likelihood = prior_sde, likelihood
        self.grid = torch.as_tensor(grid, dtype=torch.float64)
        self.num_states = int(self.grid.numel())
        self.num_transitions = self.num_states - 1
        self.dt = float(self.grid[1] - self.grid[0])
        self.device = observations.device
        d, B, N = self.state_dim, self.B, self.num_transitions
        if getattr(prior_sde, "quad_kind", None) is None or d > 3:
            raise ValueError("VariationalMarkovGPQuadrature needs a drift of the quadrature kernels and a state dimension <= 3")
        self.plan = plan if plan is not None else Plan(B, self.num_states, d, device=self.device)
        self.lib = self.plan.lib
        if prior_initial_state is None:
            prior_initial_state = (np.zeros(d), prior_sde.q.cpu().numpy() * np.eye(d))
        self.p0_mu = np.asarray(prior_initial_state[0], dtype=np.float64).reshape(d)
        self.p0_cov = np.asarray(prior_initial_state[1], dtype=np.float64).reshape(d, d)
        self.q0_mu = torch.from_numpy(self.p0_mu).to(self.device).expand(B, d).contiguous()
        self.q0_chol = torch.from_numpy(np.linalg.cholesky(self.p0_cov)).to(self.device).expand(B, d, d).contiguous()
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.A, self.b = z(B, N, d, d), z(B, N, d)
        self.lambda_lagrange = z(B, N, d)
        self.psi_lagrange = (1e-10 * torch.eye(d, dtype=torch.float64, device=self.device)).expand(B, N, d, d).contiguous()
        self.obs_index = grid_indices(self.grid, obs_times).to(self.device)
        self.obs_node_ids = self.plan.node_ids(self.obs_index)
        self._q = prior_sde.q.to(self.device)
        self._cholQ = linalg.cholesky((self.dt * self._q).contiguous())
        self._param_version = 0
        self._mS = None

    def _quad_prm(self):
        return self.prior_sde.quad_params(self.dt, self.p0_mu, self.p0_cov)

    def _refresh_drift_params(self):
        self._param_version += 1

    def _reset_lagrange(self):
        self.lambda_lagrange.zero_()
        d = self.state_dim
        self.psi_lagrange = (1e-10 * torch.eye(d, dtype=torch.float64, device=self.device)).expand(self.B, self.num_transitions, d, d).contiguous()

    def _natural(self, mS):
        pl = self.plan
        return pl.unpack(VEC, mS[0]), pl.unpack(SYM, mS[1])

    def _forward_packed(self):
        """Marginals of the SSM of the linear drift -A x + b (vi_sde.py:171-204): transitions I - dt A, offsets dt b, process noise
        dt q; stabilised: NaN -> 1e-8 and clipping to [-1, 1] of both."""
        from .state_space_model import StateSpaceModel
        d = self.state_dim
        T_ = torch.eye(d, dtype=torch.float64, device=self.device) - self.dt * self.A
        off = self.dt * self.b
        if self.stabilize_system:
            T_ = torch.nan_to_num(T_, nan=1e-8).clamp(-1.0, 1.0)
            off = torch.nan_to_num(off, nan=1e-8).clamp(-1.0, 1.0)
        cholQ = self._cholQ.expand(self.B, self.num_transitions, d, d).contiguous()
        self.dist_q_ssm = StateSpaceModel(self.q0_mu, self.q0_chol, T_.contiguous(), off.contiguous(), cholQ, plan=self.plan)
        post = self.dist_q_ssm._posterior_packed()["s"]
        self._mS = (post["x"], post["Sig"])
        return self._mS

    def _esde_terms(self, mS, param_grad=False, shift=0):
        from . import quad
        m, S = self._natural(mS if mS is not None else self._forward_packed())
        N = self.num_transitions
        return quad.esde(self._quad_prm(), m[:, shift:shift + N], S[:, shift:shift + N], self.A, self.b, grads=not param_grad,
                         param_grad=param_grad)

    def E_sde(self, mS=None):
        return self._esde_terms(mS)[0].sum(-1) * self.dt

    def _grad_E_sde(self, mS=None):
        _, (dm, dS, _, _), _ = self._esde_terms(mS)
        return dm, dS

    def grad_prior_sde_params(self):
        """d E_sde / d (trainable drift parameters) on the path m[1:], S[1:] against the N transitions' (A, b), as the reference
        (vi_sde.py:457-470); a vector-valued parameter (the network's weights) gets a vector."""
        _, _, gth = self._esde_terms(None, param_grad=True, shift=1)
        g = (self.dt * gth.sum(dim=(0, 1))).cpu().numpy()
        out = []
        for n, jac in self.prior_sde.quad_param_jacobian().items():
            out.append(g.copy() if jac is None else float(sum(gk * jk for gk, jk in zip(g, jac))))
        return out

    def _jump_arrays(self, m, S):
        """The likelihood's gradients with respect to (m, S) at the observation nodes, scattered on the grid (vi_sde.py:262-287)."""
        B, T, d = m.shape
        n = B * self.n_obs
        mo = m.reshape(B * T, d)[self.obs_node_ids]
        Rinv = self.likelihood.inv_covariance
        dmu = (self.observations.reshape(n, d) - mo) @ Rinv
        dobsm = torch.zeros((B * T, d), dtype=torch.float64, device=self.device).index_add_(0, self.obs_node_ids, dmu)
        dobsS = torch.zeros((B * T, d, d), dtype=torch.float64, device=self.device).index_add_(
            0, self.obs_node_ids, (-0.5 * Rinv).expand(n, d, d))
        return dobsm.view(B, T, d), dobsS.view(B, T, d, d)

    def update_lagrange(self, mS=None):
        from . import quad
        mS = mS if mS is not None else self._mS
        m, S = self._natural(mS)
        dEdm, dEdS = self._grad_E_sde(mS)
        dobsm, dobsS = self._jump_arrays(m, S)
        self.psi_lagrange, self.lambda_lagrange = quad.vdp_lagrange(self.A, dEdm, dEdS, dobsm, dobsS, self.dt,
                                                                    clip=5000.0 if self.stabilize_system else 0.0)

    def update_param(self, mS=None, lr=0.1):
        """A~ = -E[df/dx] + 2 q psi, b~ = E f + A~ m - q lambda, damped (vi_sde.py:377-414); the expectations by the 10-point rule."""
        from . import quad
        self._param_version += 1
        m, S = self._natural(mS if mS is not None else self._mS)
        N, d = self.num_transitions, self.state_dim
        m, S = m[:, :N], S[:, :N]
        if self.stabilize_system:
            self.psi_lagrange = torch.nan_to_num(self.psi_lagrange, nan=1e-8).clamp(-5000.0, 5000.0)
            self.lambda_lagrange = torch.nan_to_num(self.lambda_lagrange, nan=1e-8).clamp(-5000.0, 5000.0)
        Al, bl = quad.linearize(self._quad_prm(), m, S)                 # I + dt E J,  dt (E f - E J m)
        EJ = (Al - torch.eye(d, dtype=torch.float64, device=self.device)) / self.dt
        Ef = bl / self.dt + (EJ @ m[..., None])[..., 0]
        A_t = -EJ + 2.0 * self._q @ self.psi_lagrange
        b_t = Ef + (A_t @ m[..., None])[..., 0] - (self._q @ self.lambda_lagrange[..., None])[..., 0]
        self.A = (1 - lr) * self.A + lr * A_t
        self.b = (1 - lr) * self.b + lr * b_t

    def update_lagrange_and_param(self, mS=None, lr=0.1):
        self.update_lagrange(mS)
        self.update_param(mS, lr)

    def update_initial_statistics(self, lr):
        lam0, psi0 = self.lambda_lagrange[:, 0], self.psi_lagrange[:, 0]
        P0 = torch.from_numpy(self.p0_cov).to(self.device)
        mu0 = torch.from_numpy(self.p0_mu).to(self.device)
        mean = mu0 - (P0 @ lam0[..., None])[..., 0]
        # psi is not symmetric (the sweep adds psi A + psi A): a general inverse, as the reference's tf.linalg.inv
        cov = linalg.general_inverse(linalg.spd_inverse(P0) + 2.0 * psi0)
        q0_cov = self.q0_chol @ self.q0_chol.transpose(-1, -2)
        self.q0_mu = (1 - lr) * self.q0_mu + lr * mean
        self.q0_chol = linalg.cholesky(((1 - lr) * q0_cov + lr * cov).contiguous())

    def elbo_per_trajectory(self, mS=None):
        pl = self.plan
        m, S = mS if mS is not None else self._forward_packed()
        n, d = self.B * self.n_obs, self.state_dim
        mu = pl.gather_nodes(VEC, m, self.obs_node_ids)
        cov = pl.gather_nodes(SYM, S, self.obs_node_ids)
        e_obs = self.likelihood.variational_expectations(mu, cov, self.observations.reshape(n, d)).reshape(self.B, self.n_obs).sum(-1)
        return e_obs - self.E_sde((m, S)) - self.KL_initial_state()
