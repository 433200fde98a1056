"""
bench.py -- ELBO steps/sec on the block-tri-diagonal Gauss-Markov path.

  python bench.py --gpus N --steps K --warmup W [--config headline|c1|c2|c3|c5]
      N > 1: either the caller starts the N ranks (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...), or,
      when no torch.distributed environment is present, bench.py starts them itself as a child job and relays rank 0's line.

--config headline (default; the size BASELINE.json's metric is quoted on): one "step" = one iteration of the reference's inner
training loop (docs/diffusion_processes/cvi_dp_trainer.py:72-75)
    model.update_data_sites(lr); model.update_girsanov_sites(lr); model.classic_elbo()
on B = 64 independent synthetic double-well trajectories per GPU (T = 100 000 states, state dim d = 6; weak scaling: every rank
owns its own B trajectories, the only collective is the RCCL all-reduce of the scalar ELBO sum).
The other configs are BASELINE.json's `configs` (SURVEY.md 8d recipes), same JSON contract:
    c1  CVI-DP on a 1-d Ornstein-Uhlenbeck SDE, T = 1001, 32 observations, one trajectory (the reference's own CPU-runnable case)
    c2  Matern-5/2 kernel, T = 100 000, d = 3, one chain: CVIGaussianProcess.update_sites(); elbo()
    c3  double-well SDE, VDP model (VariationalMarkovGP), T = 50 000, d = 6, 64 trajectories per GPU (c4 = c3 with --gpus 8)
    c5  Sum-of-Matern kernel d = 16, 200 000 inducing states, 400 000 observations: SparseCVIGaussianProcess.update_sites(); classic_elbo()

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : the dominant kernel timed alone with HIP events on its stream (other_kernels: the rest of the level-0 sweeps)
  cpu_baseline : the CPU port / oracle of the same step on the host cores, bounded sample, N = 1 only
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def launch_ranks(n):
    """
    `python bench.py --gpus N` with N > 1 and no torch.distributed environment: start the N ranks as a CHILD
    `python -m torch.distributed.run` job (one process per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  This
    process has not touched the GPU (nothing but the standard library is imported yet) and never replaces itself.
    """
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or not lines:
        print(f"bench.py: the {n}-rank job failed (exit code {proc.returncode}, {len(lines)} result lines)", file=sys.stderr)
        sys.exit(proc.returncode or 1)
    print(lines[-1])
    sys.exit(0)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)


def obs_chol(d, noise):
    """Cholesky factor of the observation-noise covariance: correlated across the state dimensions, so that the data sites
    are full d x d blocks and the posterior genuinely couples the dimensions (the drift and diffusion act per dimension)."""
    return noise * (np.eye(d) + 0.3 * np.eye(d, k=-1))


def synth_double_well(B, T, d, dt, obs_every, noise, seed):
    """Euler-Maruyama double-well trajectories f(x) = 4x(1-x^2), q = I and noisy observations."""
    Lc = obs_chol(d, noise)
    rng = np.random.default_rng(seed)
    x = np.where(rng.random((B, d)) < 0.5, -1.0, 1.0)
    idx = np.arange(obs_every, T, obs_every)
    ys = np.empty((B, len(idx), d))
    sq = np.sqrt(dt)
    k = 0
    for t in range(1, T):
        x = x + dt * 4.0 * x * (1.0 - x * x) + sq * rng.standard_normal((B, d))
        if k < len(idx) and t == idx[k]:
            ys[:, k] = x + rng.standard_normal((B, d)) @ Lc.T
            k += 1
    return idx, ys


def cpu_baseline(args, idx, ys, dt, noise, gpu_first_elbo=None, drift=None, Lc=None):
    """The plain-C port (oracle/csrc/btd_ref.c) of the same CVI-DP step on a bounded sample of trajectories.
    drift = (alpha, beta, init_var): the Euler map x + dt f(x) = alpha x - beta x^3 and the variance of p(x0) = of the initial posterior
    path the prior is linearised on (default: the headline's double well, N(0, I))."""
    from oracle import c_ref
    lib = c_ref.load()
    threads = int(lib.ref_num_threads())
    Bs = max(1, min(args.B, threads))
    T, d = args.T, args.d
    alpha, beta, v0 = drift if drift is not None else (1.0 + dt * 4.0, dt * 4.0, 1.0)
    Jlin = alpha - 3.0 * beta * v0                  # linearisation on N(0, v0 I): A = E u'(x) = alpha - 3 beta v0, b = 0
    A = np.broadcast_to(Jlin * np.eye(d), (T - 1, d, d))
    off = np.zeros((T, d))
    chol = np.concatenate([np.sqrt(v0) * np.eye(d)[None], np.broadcast_to(np.sqrt(dt) * np.eye(d), (T - 1, d, d))], axis=0)
    lin, diag, sub = c_ref.ssm_to_naturals(A, off, chol)
    rep = lambda a: np.broadcast_to(a, (Bs,) + a.shape).copy()
    Lc = obs_chol(d, noise) if Lc is None else Lc
    Rinv = np.linalg.inv(Lc @ Lc.T)
    st = c_ref.CviDpStepState(rep(lin), rep(diag), rep(sub), idx, ys[:Bs], Rinv, 2 * np.sum(np.log(np.diag(Lc))), alpha, beta,
                              np.ones(d), dt, np.zeros(d), v0 * np.eye(d))
    st.step(args.lr_data, args.lr_girsanov)         # first step doubles as warm-up and as a parity probe
    first = st.elbo.copy()
    for _ in range(2):                              # >= 20 steps after 3 warm-ups (bounded at 30 s of CPU work)
        st.step(args.lr_data, args.lr_girsanov)
    times = []
    t0 = time.perf_counter()
    while True:
        t1 = time.perf_counter()
        st.step(args.lr_data, args.lr_girsanov)
        times.append(time.perf_counter() - t1)
        if time.perf_counter() - t0 > 30.0 or len(times) >= 20:
            break
    n, el = len(times), float(sum(times))
    per_step_sample = float(np.median(times))
    value = 1.0 / (per_step_sample * args.B / Bs)
    # one thread, one trajectory (the OpenMP loop runs over trajectories: a single trajectory is the single-thread figure)
    lib.ref_set_num_threads(1)
    st1 = c_ref.CviDpStepState(lin[None].copy(), diag[None].copy(), sub[None].copy(), idx, ys[:1], Rinv, 2 * np.sum(np.log(np.diag(Lc))),
                               alpha, beta, np.ones(d), dt, np.zeros(d), v0 * np.eye(d))
    st1.step(args.lr_data, args.lr_girsanov)
    t1 = time.perf_counter()
    st1.step(args.lr_data, args.lr_girsanov)
    one = time.perf_counter() - t1
    lib.ref_set_num_threads(threads)
    out = {"value": value, "unit": "ELBO steps/s", "cores": threads, "threads_used": Bs, "kind": "port",
           "single_thread_value": 1.0 / (one * args.B),
           "sample": f"median of {n} steps after 3 warm-ups of {Bs} of the {args.B} trajectories (T={T}, d={d}), OpenMP over trajectories ({Bs} of the {threads} "
                     f"hardware threads busy: one per trajectory); {per_step_sample:.3f} s per sampled step, scaled linearly to "
                     f"{args.B} trajectories; single thread: one step of one trajectory {one:.3f} s, scaled to {args.B}; "
                     f"closed-form cubic-drift moments as on the GPU (the reference's 20^d-point quadrature is infeasible at d={d})"}
    if gpu_first_elbo is not None:
        ref = first[:Bs]
        out["first_step_elbo_max_rel_diff_vs_gpu"] = float(np.max(np.abs(gpu_first_elbo[:Bs] - ref) / np.abs(ref)))
    if T * d >= 100000:
        out["lapack_banded"] = lapack_banded_comparator(T, d, dt)
    return out


def cpu_baseline_vdp(B, T, d, dt, noise, idx, ys, gpu_first_elbo=None):
    """The plain-C port (oracle/csrc/btd_ref.c: ref_vdp_step, OpenMP over trajectories) of the same VDP inference step -- update_lagrange +
    update_param, forward_pass, elbo (vi_markov_gp_trainer.py:55-75) -- on a bounded sample, started at the OU drift -4 x like the GPU run."""
    from oracle import c_ref
    lib = c_ref.load()
    threads = int(lib.ref_num_threads())
    Bs = max(1, min(B, threads))
    Lc = obs_chol(d, noise)
    A0 = np.broadcast_to(4.0 * np.eye(d), (Bs, T - 1, d, d))
    st = c_ref.VdpStepState(A0, np.zeros((Bs, T - 1, d)), idx, ys[:Bs], np.linalg.inv(Lc @ Lc.T), 2 * np.sum(np.log(np.diag(Lc))), 4.0, 4.0,
                            np.ones(d), dt, np.zeros(d), np.eye(d), stabilize=True)
    st.step(0.01)
    first = st.elbo.copy()
    n, t0 = 0, time.perf_counter()
    while True:
        st.step(0.01)
        n += 1
        el = time.perf_counter() - t0
        if el > 8.0 or n >= 20:
            break
    per = el / n
    out = {"value": 1.0 / (per * B / Bs), "unit": "ELBO steps/s", "cores": threads, "threads_used": Bs, "kind": "port",
           "sample": f"{n} steps of {Bs} of the {B} trajectories (T={T}, d={d}), OpenMP over trajectories; {per:.3f} s per sampled step, scaled "
                     f"linearly to {B} trajectories; closed-form cubic-drift moments as on the GPU"}
    if gpu_first_elbo is not None:
        out["first_step_elbo_max_rel_diff_vs_gpu"] = float(np.max(np.abs(gpu_first_elbo[:Bs] - first) / np.abs(first)))
    return out


def lapack_banded_comparator(T, d, dt):
    """Independent CPU comparator: LAPACK's banded Cholesky (dpbtrf / dpbtrs through scipy.linalg.cholesky_banded /
    cho_solve_banded, one thread) on ONE trajectory's posterior precision of the same shape -- T*d rows, lower bandwidth 2d-1 -- i.e.
    what the reference's banded_matrices ops do per refresh (factor + one solve; the step has two refreshes and a selected inverse each
    on top).  Times only; not part of `value`."""
    try:
        from scipy.linalg import cho_solve_banded, cholesky_banded
    except ImportError:
        return None
    n, bw = T * d, 2 * d - 1
    rng = np.random.default_rng(0)
    ab = np.zeros((bw + 1, n))
    ab[0] = 2.0 / dt + rng.random(n)                      # diagonally dominant SPD band, magnitudes of the Euler-chain precision
    for k in range(1, bw + 1):
        ab[k, :n - k] = -0.3 / dt * rng.random(n - k) / bw
    r = rng.standard_normal(n)
    t0 = time.perf_counter()
    cb = cholesky_banded(ab, lower=True, check_finite=False)
    t1 = time.perf_counter()
    cho_solve_banded((cb, True), r, check_finite=False)
    t2 = time.perf_counter()
    return {"what": f"scipy.linalg.cholesky_banded + cho_solve_banded (LAPACK dpbtrf / dpbtrs), one trajectory, n = {n}, bandwidth {bw}, 1 thread",
            "factor_ms": 1e3 * (t1 - t0), "solve_ms": 1e3 * (t2 - t1)}


def vdp_step_rate(B, T, d, dt, noise, idx, ys, device, steps=10):
    """
    Secondary figure: the VDP (VariationalMarkovGP) inference step of VIMarkovGPTrainer.perform_inference on the same trajectories
    (Lagrange sweep + parameter update, forward pass, ELBO), stabilize_system on, q started at the OU drift -4 x (from A = 0 the
    marginal variance of a chain this long reaches T dt and the sixth-order moments overflow the first update).
    """
    import torch
    import vidp_amd
    from vidp_amd.likelihoods import MultivariateGaussian
    from vidp_amd.sde import DoubleWellSDE
    from vidp_amd.vi_sde import VariationalMarkovGP
    grid = np.arange(T) * dt
    lik = MultivariateGaussian(torch.from_numpy(obs_chol(d, noise)).to(device))
    m = VariationalMarkovGP((grid[idx], torch.from_numpy(ys).to(device)), DoubleWellSDE(q=torch.eye(d, dtype=torch.float64)), grid, lik,
                            prior_initial_state=(np.zeros(d), np.eye(d)), stabilize_system=True, plan=None)
    eye = (4.0 * torch.eye(d, dtype=torch.float64, device=device)).expand(B, T, d, d).contiguous()
    m.plan.pack(vidp_amd.FULL, eye, out=m.A)
    del eye
    mS = m._forward_packed()
    e = None
    for it in range(3 + steps):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        m.update_lagrange_and_param(mS, lr=0.01)
        mS = m._forward_packed()
        e = m.elbo(mS)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m.plan.check_info()
    e = float(e)
    assert np.isfinite(e), "non-finite VDP ELBO"
    return {"value": steps / el, "unit": "ELBO steps/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "elbo_last": e,
            "workload": f"VDP (VariationalMarkovGP) inference step on the same {B} trajectories, T={T}, d={d}, stabilize_system on"}


def pmc_traffic(kernel_name, B, T, d, build):
    """(HBM bytes per launch of `kernel_name`, source note) from the committed rocprofv3 PMC passes of this bench
    (profiles/rNN_pmc/pmc_traffic.json, newest round first: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs, read side doubled as the
    gfx950 guide prescribes).  The figure is archival, not measured by this run: it is used only when the file was collected on THIS
    build of the library (mfgm_version()) and this workload size; otherwise (None, why)."""
    import glob
    why = "no PMC profile committed"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc", "pmc_traffic.json")), reverse=True):
        rel = os.path.relpath(path, ROOT)
        try:
            with open(path) as fh:
                prof = json.load(fh)
        except (OSError, ValueError):
            continue
        if prof.get("workload") != {"B": B, "T": T, "d": d}:
            why = "the committed PMC profile is for another workload size"
            continue
        if prof.get("library_build") != build:
            why = f"the committed PMC profile ({rel}) was collected on library build {prof.get('library_build')!r}, this is {build!r}"
            continue
        try:
            return float(prof["kernels"][kernel_name]["hbm_bytes_per_launch"]), f"{rel} (build {build})"
        except KeyError:
            why = f"kernel not in the committed PMC profile ({rel})"
    return None, why


class Harness:
    """What every configuration shares: the warm-up / timed loop bracketed by barriers and synchronisation, the maximum over ranks,
    HIP-event timing of one call, and the contract fields of the JSON line."""

    def __init__(self, args, rank, world, device, dist, vdist):
        self.args, self.rank, self.world, self.device, self.dist, self.vdist = args, rank, world, device, dist, vdist

    def fence(self):
        import torch
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def run(self, step):
        """W untimed + K timed steps; returns seconds (maximum over ranks)."""
        import torch
        for _ in range(self.args.warmup):
            step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(self.args.steps):
            step()
        self.fence()
        el = time.perf_counter() - t0
        return float(self.vdist.allreduce_max_(torch.tensor([el], dtype=torch.float64, device=self.device)).item())

    @staticmethod
    def timed(fn, reps=20):
        """average milliseconds of fn() by HIP events on torch's current stream (the stream the library launches on)"""
        import torch
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        fn()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    def line(self, elapsed, workload, name, extra_config):
        a = self.args
        return {"metric": "ELBO steps/sec (T=100k, d=6) at 1/2/4/8 GPU; HBM GB/s vs roofline", "value": a.steps / elapsed,
                "unit": "ELBO steps/s", "n_gpus": self.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": dict({"workload": workload, "name": name}, **extra_config)}

    @staticmethod
    def roofline(kernel, what, k_ms, alg_bytes, per_step, ms_per_step, **more):
        ach = alg_bytes / (k_ms * 1e-3) / 1e9
        return dict({"bound": "hbm", "kernel": f"{kernel} ({what})", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_source": "no PMC profile committed for this configuration",
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes, "launches_per_step": per_step,
                     "share_of_step": per_step * k_ms / ms_per_step}, **more)


def bench_vdp(h, data_rank):
    """Config 3 (and 4 with --gpus 8): the VDP inference step of VIMarkovGPTrainer.perform_inference (vi_markov_gp_trainer.py:50-75:
    Lagrange sweep + parameter update, forward pass, ELBO) on 64 double-well trajectories per GPU, T = 50 000, d = 6."""
    import ctypes
    import torch
    import vidp_amd
    from vidp_amd.likelihoods import MultivariateGaussian
    from vidp_amd.packed import _ptr, _stream
    from vidp_amd.sde import DoubleWellSDE
    from vidp_amd.vi_sde import VariationalMarkovGP
    a, device = h.args, h.device
    B, T, d, dt, noise = a.B, 50000, 6, 0.01, 0.1
    idx, ys = synth_double_well(B, T, d, dt, a.obs_every, noise, seed=71892305 + 3 + data_rank)
    grid = np.arange(T) * dt
    lik = MultivariateGaussian(torch.from_numpy(obs_chol(d, noise)).to(device))
    m = VariationalMarkovGP((grid[idx], torch.from_numpy(ys).to(device)), DoubleWellSDE(q=torch.eye(d, dtype=torch.float64)), grid, lik,
                            prior_initial_state=(np.zeros(d), np.eye(d)), stabilize_system=True, plan=None)
    # q starts at the OU drift -4 x: from A = 0 the marginal variance of a chain this long reaches T dt and the sixth-order moments
    # overflow the first update
    eye = (4.0 * torch.eye(d, dtype=torch.float64, device=device)).expand(B, T, d, d).contiguous()
    m.plan.pack(vidp_amd.FULL, eye, out=m.A)
    del eye
    state = {"mS": m._forward_packed(), "e": None, "first": None}

    def step():
        m.update_lagrange_and_param(state["mS"], lr=0.01)
        state["mS"] = m._forward_packed()
        if state["first"] is None:
            state["first"] = m.elbo_per_trajectory(state["mS"]).clone()      # parity probe against the C port's first step
        state["e"] = h.vdist.allreduce_sum_(m.elbo(state["mS"]))

    elapsed = h.run(step)
    m.plan.check_info()
    e = float(state["e"])
    assert np.isfinite(e), "non-finite VDP ELBO"
    pl = m.plan
    out = h.line(elapsed, f"VDP (VariationalMarkovGP) inference step (update_lagrange + update_param, forward_pass, elbo) on double-well SDE "
                          f"trajectories, T={T}, d={d}, {B} trajectories per GPU, stabilize_system on", "c3",
                 {"trajectories_per_gpu": B, "T": T, "d": d, "total_trajectories": B * h.world,
                  "partition": {"levels": pl.nlevels, "segment_len": pl.R, "lanes": pl.Lpad}})
    out["elbo_last"] = e
    if h.rank == 0:
        # the final Lagrange sweep (psi, lambda by the partitioned affine recurrence, then the update of (A, b) node by node) alone,
        # on scratch copies of (A, b): the kernel with the largest share of the step
        lib = vidp_amd._lib.load()
        A2, b2 = m.A.clone(), m.b.clone()
        mS = state["mS"]
        prm = m._params(lr=0.01)

        lean = not m.store_multipliers
        psi0, lam0 = (torch.empty((B, d, d), dtype=torch.float64, device=device), torch.empty((B, d), dtype=torch.float64, device=device))

        # the seg array the loop's Lagrange calls ran on (their segment scans are what the final sweep starts from)
        lseg = m._lseg if getattr(m, "_lseg", None) is not None else m._seg

        def final():
            if lean:
                rc = lib.mfgm_packed_vdp_lagrange_update0(pl.h, ctypes.byref(prm), _ptr(mS[0]), _ptr(mS[1]), _ptr(A2), _ptr(b2), _ptr(m._yR),
                                                          _ptr(m._dobsS), _ptr(psi0), _ptr(lam0), _ptr(lseg), *m._jump_args(), 1, _stream())
            else:
                rc = lib.mfgm_packed_vdp_lagrange_update_final(pl.h, ctypes.byref(prm), _ptr(mS[0]), _ptr(mS[1]), _ptr(A2), _ptr(b2),
                                                               _ptr(m._yR), _ptr(m._dobsS), _ptr(m.psi_lagrange), _ptr(m.lambda_lagrange),
                                                               _ptr(m._seg), *m._jump_args(), _stream())
            assert rc == 0
        ET = d * (d + 1) // 2
        # reads m, S, A, b, yR, the observation count (half a double); writes the new A, b (and, unless the sweep keeps the multipliers
        # of node 0 only, psi and lambda)
        doubles = (d + ET + d * d + d + d + 0.5) + (0 if lean else d * d + d) + (d * d + d)
        out["roofline"] = h.roofline(f"void mfgm::k_vdp_lagrange<{d}, {5 if lean else 4}>(...)",
                                     "final Lagrange sweep with the parameter update: reads m, S, A, b, R^-1 y, writes A, b"
                                     + (" (the multipliers are kept at node 0 only: the fused update has consumed the others)" if lean
                                        else ", psi, lambda"),
                                     h.timed(final), 8 * doubles * B * T, 1, out["ms_per_step"])
        if h.world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline_vdp(B, T, d, dt, noise, idx, ys, state["first"].cpu().numpy())
            except OSError as e:  # library not built
                out["cpu_baseline"] = {"value": None, "unit": "ELBO steps/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return out


def bench_cvigp(h, data_rank):
    """Config 2: Matern-5/2 kernel, T = 100 000, d = 3, one chain; one step = CVIGaussianProcess.update_sites(); elbo()
    (variational_cvi.py:351-379; elbo = the Kalman log-likelihood with sites, kalman_filter.py:184-255)."""
    import ctypes
    import torch
    import vidp_amd
    from vidp_amd import kernels as K
    from vidp_amd.likelihoods import Gaussian
    from vidp_amd.packed import _ptr, _stream
    from vidp_amd.variational_cvi import CVIGaussianProcess
    a, device = h.args, h.device
    T, d = 100000, 3
    rng = np.random.default_rng(71892305 + 2 + data_rank)
    t = torch.linspace(0, 0.01 * T, T, dtype=torch.float64, device=device)          # rho = dt / lengthscale = 0.05
    y = (torch.sin(12 * t) + 0.1 * torch.from_numpy(rng.normal(size=T)).to(device))[:, None]
    m = CVIGaussianProcess((t, y), K.Matern52(lengthscale=0.2, variance=1.0), Gaussian(0.01), learning_rate=0.5)
    state = {"e": None}

    firsts = []

    def step():
        m.update_sites()
        e_loc = m.elbo()
        if len(firsts) < 2:
            firsts.append(e_loc.clone())                 # this rank's own chain, for the parity probe against the C port
        state["e"] = h.vdist.allreduce_sum_(e_loc)

    elapsed = h.run(step)
    e = float(state["e"])
    assert np.isfinite(e), "non-finite ELBO"
    pl = m.dist_p.plan
    out = h.line(elapsed, f"CVI for GP regression (CVIGaussianProcess.update_sites + elbo), Matern-5/2 kernel, T={T}, d={d}, one chain per GPU",
                 "c2", {"trajectories_per_gpu": 1, "T": T, "d": d, "total_trajectories": h.world,
                        "partition": {"levels": pl.nlevels, "segment_len": pl.R, "lanes": pl.Lpad}})
    out["elbo_last"] = e
    if h.rank == 0:
        lib = vidp_amd._lib.load()
        ssm = m.dist_p
        pr, b = ssm._precision_packed(), ssm._kf_cache["bufs"]
        null = ctypes.c_void_p(0)

        def fwd():
            assert lib.mfgm_packed_factor_stage(pl.h, 1, 0, _ptr(b["D"]), _ptr(pr["sub"]), _ptr(b["r"]), 1.0, 1.0, 1.0, _ptr(b["L"]), null,
                                                _ptr(b["y"]), _ptr(pl.ws), _ptr(pl.info), _stream()) == 0
        ET = d * (d + 1) // 2
        out["roofline"] = h.roofline(f"void mfgm::k_forward<{d}, true, false, true>(mfgm::SweepArgs)",
                                     "level 0 forward sweep; one chain: the step is bound by dependent launches and short sweeps, not by HBM",
                                     h.timed(fwd), 8 * ((ET + d * d + d) + (ET + d)) * T, 2, out["ms_per_step"])
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_cvigp(t.cpu().numpy(), y.cpu().numpy(), T, [float(x) for x in firsts])
    return out


def cpu_baseline_cvigp(t, y, T, gpu_elbos=None):
    """The plain-C port (oracle/csrc/btd_ref.c: ref_cvigp_step, one thread -- the step is one chain) of the same update_sites + elbo
    step at FULL size; the prior's precision blocks come from the NumPy oracle's kernel, once, as the GPU model builds its own once.
    gpu_elbos: the ELBOs of the GPU run's first steps, compared with the port's."""
    from oracle import c_ref, np_kernels
    k = np_kernels.Matern52(0.2, 1.0)
    st = c_ref.CviGpStepState(k.state_space_model(t), k.emission_matrix(t[:1])[0, 0], y, 0.01, 0.5)
    first = [st.step() for _ in range(2)]
    n, t0 = 0, time.perf_counter()
    while True:
        st.step()
        n += 1
        el = time.perf_counter() - t0
        if el > 10.0 or n >= 40:
            break
    out = {"value": n / el, "unit": "ELBO steps/s", "cores": 1, "threads_used": 1, "kind": "port",
           "sample": f"{n} full-size steps (T={T}) of the C port, one thread (one chain: nothing to spread over cores), {el / n:.4f} s each"}
    if gpu_elbos is not None and len(gpu_elbos) >= 2:
        out["first_steps_elbo_max_rel_diff_vs_gpu"] = float(max(abs(a - b) / abs(b) for a, b in zip(gpu_elbos[:2], first)))
    return out


def sum16_kernel(K):
    """Sum of 4 x Matern-5/2 + 2 x Matern-3/2 (state dimension 3*4 + 2*2 = 16), lengthscales log-spaced 0.05 .. 2, unit variances."""
    ls = np.exp(np.linspace(np.log(0.05), np.log(2.0), 6))
    return K.Sum([K.Matern52(float(l), 1.0) for l in ls[:4]] + [K.Matern32(float(l), 1.0) for l in ls[4:]])


def bench_sparse(h, data_rank):
    """Config 5: sparse / inducing-state CVI (SparseCVIGaussianProcess.update_sites + classic_elbo, sparse_variational_cvi.py:176-292),
    Sum-of-Matern kernel with state dimension 16, M = T = 200 000 inducing states on a uniform grid, N = 2 M observations."""
    import ctypes
    import torch
    import vidp_amd
    from vidp_amd import kernels as K
    from vidp_amd.likelihoods import Gaussian
    from vidp_amd.packed import _ptr, _stream
    from vidp_amd.sparse_variational_cvi import SparseCVIGaussianProcess
    a, device = h.args, h.device
    M, d = a.c5_M, 16
    # grid spacing 0.1: rho = spacing / lengthscale >= 0.05 as in config 2 ("value ranges keep every Q_k SPD with cond <~ 1e8";
    # at spacing 0.01 the Matern-5/2 component with lengthscale 2 makes the prior precision numerically singular in fp64, cond ~ 1e15)
    N, span = 2 * M, 0.1 * M
    shared = h.world > 1          # config 5 as BASELINE states it: ONE chain over the GPUs of the node, cut along time (strong scaling)
    rng = np.random.default_rng(71892305 + 5 + (0 if shared else data_rank))
    z = torch.linspace(0, span, M, dtype=torch.float64, device=device)
    t = torch.from_numpy(np.sort(rng.uniform(0, span, size=N))).to(device)
    y = (torch.sin(3 * t) + 0.1 * torch.from_numpy(rng.normal(size=N)).to(device))[:, None]
    kern = sum16_kernel(K)
    assert kern.state_dim == d
    # shared: every rank owns a contiguous range of inducing states, their sites and the data points between them; per step the
    # ranks exchange the edge sites (all-gather), the exchange level of the factorisation and the four ELBO scalars (distributed.ChainShard)
    m = SparseCVIGaussianProcess(kern, z, Gaussian(0.01), learning_rate=0.5, shard=(h.rank, h.world) if shared else None)
    state = {"e": None}

    def step():
        m.update_sites((t, y))
        state["e"] = m.classic_elbo((t, y))              # shared: already the sum over the ranks

    elapsed = h.run(step)
    pl = m._shard.plan if shared else m.dist_p.plan
    pl.check_info()
    e = float(state["e"])
    assert np.isfinite(e), "non-finite ELBO"
    out = h.line(elapsed, f"sparse / inducing-state CVI (SparseCVIGaussianProcess.update_sites + classic_elbo), Sum-of-Matern kernel d={d}, "
                          f"{M} inducing states, {N} observations, " + (f"ONE chain shared between {h.world} GPUs along time" if shared
                                                                         else "one chain on one GPU"), "c5",
                 {"trajectories_per_gpu": 1.0 / h.world, "T": M, "d": d, "observations": N, "total_trajectories": 1,
                  "partition": {"levels": pl.nlevels, "segment_len": pl.R, "segments": pl.P}})
    if shared:
        sh = m._shard
        out["scaling"] = "strong"
        out["config"]["parallelism"] = (f"time-sharded chain: exchange level {sh.level} ({sh.exchange.numel()} doubles all-reduced per "
                                        f"factorisation), edge sites all-gathered ({m.nat1.shape[-1] + m.nat2.shape[-1] ** 2} doubles per rank), "
                                        f"4 ELBO scalars all-reduced; rank 0 owns inducing states [{sh.node_lo}, {sh.node_hi})")
    out["elbo_last"] = e
    if h.rank == 0 and not shared:
        # the level-0 kernels of the MFMA sweeps alone (mfgm_wide_stage), on the model's own arrays, in the form the model uses
        lib = vidp_amd._lib.load()
        m._marginals()
        f, sv = m._sweep_bufs["f"], m._sweep_bufs["s"]
        form = f.get("form", 0)
        fused = m._fused_theta()       # the passes read the prior naturals + the sites instead of materialised posterior naturals
        null = None
        packed = fused and getattr(m, "_packed", False)      # the sites' resident form: quadrant-packed (include/mfgm.h)
        if fused:
            pn = m._prior_natural()
            lin, diag, sub, s1, s2 = pn["lin"], pn["diag"], pn["sub"], m.nat1, (m._nat2q if packed else m.nat2)
        else:
            (lin, diag, sub), s1, s2 = m._theta(), None, None

        def stage(which):
            if which < 2 and packed:
                rc = lib.mfgm_wide_stage_q(pl.h, which, _ptr(diag), _ptr(sub), _ptr(lin), -2.0, -1.0, 1.0, _ptr(f["L"]), _ptr(f["G"]),
                                           _ptr(f["y"]), _ptr(s1), _ptr(s2), _ptr(pl.ws), _ptr(pl.info), _stream())
            elif which < 2:
                rc = lib.mfgm_wide_stage(pl.h, form, which, _ptr(diag), _ptr(sub), _ptr(lin), -2.0, -1.0, 1.0, _ptr(f["L"]), _ptr(f["G"]),
                                         _ptr(f["y"]), null, null, null, _ptr(s1), _ptr(s2), _ptr(pl.ws), _ptr(pl.info), _stream())
            else:
                rc = lib.mfgm_wide_stage(pl.h, form, 2, null, null, null, 1.0, 1.0, 1.0, _ptr(f["L"]), _ptr(f["G"]), _ptr(f["y"]),
                                         _ptr(sv["Sig"]), _ptr(sv["Sub"]), _ptr(sv["x"]), null, null, _ptr(pl.ws), _ptr(pl.info), _stream())
            assert rc == 0

        pre = "kmi" if form == 1 else "km"
        EF = d * d
        ETq = d * (d + 1) // 2
        # fused: prior naturals + the site quadrants a node reads (packed: two triangles + one full block) + site vectors
        rd = ((2 * EF + d) + ((2 * ETq + EF) if packed else 3 * EF) + 2 * d) if fused else (2 * EF + d)
        rows = [h.roofline(f"mfgm::{pre}_{name}", what, h.timed(lambda w=w: stage(w)), 8 * dbl * M, 1, out["ms_per_step"])
                for name, w, dbl, what in (
                    ("forward<1, true, false, true>", 1, rd + (2 * EF + d),
                     "level 0 forward of the MFMA sweeps (one wavefront per segment, 16 x 16 fp64 MFMA tiles): pivot blocks inverted by 4 x 4 "
                     "block sweeps; reads D, S, r, writes F^-1, S F^-1, F^-1 h" if form == 1 else
                     "level 0 forward of the MFMA sweeps: block Cholesky + forward substitution; reads D, S, r, writes L, L_{t+1,t}, y"),
                    ("reduce<1, true, false>", 0, rd, "level 0 reduce of the MFMA sweeps: segment elimination; reads D, S, r"
                     + (" as prior naturals + overlap-added sites" if fused else "")),
                    ("backward<1, true, true, true>", 2, (2 * EF + d) + (2 * EF + d),
                     "level 0 backward of the MFMA sweeps: selected inverse + back-substitution; reads the factor arrays, writes Sigma_tt, "
                     "Sigma_{t+1,t}, mu"))]
        rows.sort(key=lambda r: -r["share_of_step"])
        # these kernels are fp64-arithmetic bound, not HBM bound: per node the forward pass issues 16 and the reduce 28
        # v_mfma_f64_16x16x4_f64 (64 cycles each on the one fp64 pipe of a SIMD) plus the pivot-block inverses
        out["roofline"] = dict(rows[0], other_kernels=rows[1:])
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_sparse(M, N, device=device)
    return out


def c5_sample_problem(sample):
    """Config 5's recipe at `sample` inducing states (same grid spacing, 2 x sample observations): the bounded problem the C port is
    timed on and on which the GPU model is compared with it -- SAME size on both sides, nothing scaled."""
    rng = np.random.default_rng(5)
    span = 0.1 * sample
    z = np.linspace(0, span, sample)
    t = np.sort(rng.uniform(0, span, size=2 * sample))
    y = np.sin(3 * t) + 0.1 * rng.normal(size=t.size)
    return z, t, y


def c5_gpu_elbos(sample, steps, device):
    """SparseCVIGaussianProcess on c5_sample_problem(sample): the ELBO after each of `steps` damped update_sites (learning rate 0.5)."""
    import torch
    from vidp_amd import kernels as K
    from vidp_amd.likelihoods import Gaussian
    from vidp_amd.sparse_variational_cvi import SparseCVIGaussianProcess
    z, t, y = c5_sample_problem(sample)
    td, yd = torch.from_numpy(t).to(device), torch.from_numpy(y).to(device)[:, None]
    m = SparseCVIGaussianProcess(sum16_kernel(K), torch.from_numpy(z).to(device), Gaussian(0.01), learning_rate=0.5)
    out = []
    for _ in range(steps):
        m.update_sites((td, yd))
        out.append(float(m.classic_elbo((td, yd))))
    m.dist_p.plan.check_info()
    return out


def cpu_baseline_sparse(M, N, sample=20000, device=None):
    """The plain-C port (oracle/csrc/btd_ref.c: ref_sparse_cvi_step, one thread -- the step is one chain) of the same update_sites +
    classic_elbo step on a BOUNDED sample: `sample` inducing states on the same grid spacing with 2 x sample observations.  Its RATE is
    scaled linearly to M (the algorithm is O(M + N)) and labelled as such; its ELBOs over the first three damped steps are compared
    with the GPU model run on the SAME sample (no scaling).  The data-dependent constants (interval, h^T P_n, h^T T_n h per
    observation) and the prior's precision blocks come from the NumPy oracle's kernel, once."""
    from oracle import c_ref, np_kernels
    z, t, y = c5_sample_problem(sample)
    st = c_ref.SparseCviStepState(sum16_kernel(np_kernels), z, t, y, 0.01, 0.5)
    first = [st.step() for _ in range(3)]
    n, t0 = 0, time.perf_counter()
    while True:
        st.step()
        n += 1
        el = time.perf_counter() - t0
        if el > 10.0 or n >= 10:
            break
    per = el / n * (M / sample)
    out = {"value": 1.0 / per, "unit": "ELBO steps/s", "cores": 1, "threads_used": 1, "kind": "port",
           "sample": f"{n} steps of the C port on {sample} inducing states / {2 * sample} observations ({el / n:.3f} s each, one thread: the "
                     f"step is one chain); the RATE is an extrapolation, scaled linearly to {M} / {N}"}
    if device is not None:
        got = c5_gpu_elbos(sample, 3, device)
        out["first_steps_elbo_max_rel_diff_vs_gpu"] = float(max(abs(a - b) / abs(b) for a, b in zip(got, first)))
        out["parity_sample"] = f"GPU model and C port on the same {sample} inducing states / {2 * sample} observations, three damped steps"
    return out


OTHER_CONFIGS = {"c3": bench_vdp, "c2": bench_cvigp, "c5": bench_sparse}


def bench_cvidp(h, data_rank):
    """The headline configuration (CVI-DP site-update loop on double-well trajectories, B = 64, T = 100 000, d = 6) and config 1 (the
    reference's own CPU-runnable case: 1-d Ornstein-Uhlenbeck, T = 1001, 32 observations): CVISitesSDE.update_data_sites +
    update_girsanov_sites + classic_elbo (docs/diffusion_processes/cvi_dp_trainer.py:72-75)."""
    import torch
    import vidp_amd
    from vidp_amd.likelihoods import MultivariateGaussian
    from vidp_amd.sde import DoubleWellSDE, OrnsteinUhlenbeckSDE
    from vidp_amd.variational_cvi_sde import CVISitesSDE
    args, rank, world, device, dist, vdist = h.args, h.rank, h.world, h.device, h.dist, h.vdist

    dt, noise = 0.01, 0.1
    if args.config == "c1":
        # OU process generated with decay 0.5, prior decay 1.2, q = 1, T = 1001, 32 observations at random grid
        # indices, sigma = 0.1, x0 = 1, both learning rates 1 (configs/cvi_linear_process.yaml)
        args.B, args.T, args.d, args.lr_data, args.lr_girsanov = 1, 1001, 1, 1.0, 1.0
        B, T, d = 1, 1001, 1
        rng = np.random.default_rng(71892305 + 1 + data_rank)
        x, xs = 1.0, np.empty(T)
        for t in range(T):
            xs[t] = x
            x = x + dt * (-0.5 * x) + np.sqrt(dt) * rng.standard_normal()
        idx = np.sort(rng.choice(np.arange(1, T), size=32, replace=False))
        ys = (xs[idx] + noise * rng.standard_normal(32))[None, :, None]
        sde = OrnsteinUhlenbeckSDE(1.2, torch.eye(1, dtype=torch.float64))
        lik = MultivariateGaussian(torch.tensor([[noise]], dtype=torch.float64, device=device))
        init = (np.zeros(1), np.eye(1) / 2.4)
        what = "CVI-DP on a 1-d Ornstein-Uhlenbeck SDE (config 1: T=1001, 32 observations, prior decay 1.2, lr 1)"
    else:
        B, T, d = args.B, args.T, args.d
        idx, ys = synth_double_well(B, T, d, dt, args.obs_every, noise, seed=71892305 + 3 + data_rank)
        sde = DoubleWellSDE(q=torch.eye(d, dtype=torch.float64))
        lik = MultivariateGaussian(torch.from_numpy(obs_chol(d, noise)).to(device))
        init = (np.zeros(d), np.eye(d))
        what = (f"CVI-DP site-update loop (CVISitesSDE: update_data_sites + update_girsanov_sites + classic_elbo) on double-well SDE "
                f"trajectories, T={T}, d={d}, {B} trajectories per GPU, observation every {args.obs_every} steps, correlated "
                f"observation noise (full d x d data sites)")
    grid = np.arange(T) * dt
    # plan=None: the model builds its own partition, segments aligned with the equally spaced observation grid (packed.aligned_segment_length)
    model = CVISitesSDE(sde, grid, (grid[idx], torch.from_numpy(ys).to(device)), lik, prior_initial_state=init, plan=None)
    if getattr(args, "dense", False):
        # the DENSE-naturals route (what VIDP_CQ=0 selects, and what every model without the structure of DESIGN 5b runs on: coupled
        # drifts, non-Gaussian likelihoods, loaded Girsanov sites): theta_q as full packed (lin, diag, sub) arrays
        model.cq_enabled = False
        what += "; DENSE posterior naturals (cq state off)"
    plan = model.plan

    elbos, first_elbo = [], []

    def step():
        model.update_data_sites(args.lr_data)
        model.update_girsanov_sites(args.lr_girsanov)
        if not first_elbo:
            e_traj = model.classic_elbo_per_trajectory()     # the parity probe against the C port's first step
            first_elbo.append(e_traj.clone())
            e = e_traj.sum()
        else:
            e = model.classic_elbo()                         # the reference's call (cvi_dp_trainer.py:75): the sum over trajectories
        elbos.append(vdist.allreduce_sum_(e))                # the only collective: scalar ELBO sum over ranks

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = float(vdist.allreduce_max_(torch.tensor([elapsed], dtype=torch.float64, device=device)).item())
    plan.check_info()
    elbo_vals = [float(e.item()) for e in elbos]
    assert all(np.isfinite(elbo_vals)), "non-finite ELBO"

    ms_per_step = 1e3 * elapsed / args.steps
    out = {
        "metric": "ELBO steps/sec (T=100k, d=6) at 1/2/4/8 GPU; HBM GB/s vs roofline",
        "value": args.steps / elapsed, "unit": "ELBO steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": what, "name": args.config, "trajectories_per_gpu": B, "T": T, "d": d, "total_trajectories": B * world,
                   "partition": {"levels": plan.nlevels, "segment_len": plan.R, "lanes": plan.Lpad}},
        "elbo_last": elbo_vals[-1],
    }

    if rank == 0:
        # ---- roofline: every level-0 sweep kernel of the step timed alone; the one with the largest share of the step is reported --
        lib = vidp_amd._lib.load()
        from vidp_amd.packed import _ptr, _stream
        import ctypes
        f, s = model._bufs["f"], model._bufs["s"]
        ET, EF = d * (d + 1) // 2, d * d
        null = ctypes.c_void_p(0)
        klbuf = torch.empty(B, dtype=torch.float64, device=device)
        cq = model._cq

        def timed(fn, reps=20):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            fn()
            torch.cuda.synchronize()
            ev[0].record()   # torch's current stream is the stream the kernels are launched on (_stream())
            for _ in range(reps):
                fn()
            ev[1].record()
            torch.cuda.synchronize()
            return ev[0].elapsed_time(ev[1]) / reps

        if cq is not None:
            # structured posterior naturals (csrc/mfgm_cq.h): 3 d doubles per node + an int observation slot; the dense blocks are
            # rebuilt in registers.  Level-0 kernels alone, on the model's own state (outputs overwritten with identical values).
            cst = cq.struct()
            out["config"]["state_layout"] = ("cq: (theta_lin, diag theta_diag, diag theta_sub) per node + uniform off-diagonals; the data "
                                             "sites are added inside the sweeps (csrc/mfgm_cq.h)")
            out["config"]["pipelined_across_steps"] = bool(model.pipelined)

            def stage(st):
                assert lib.mfgm_cq_factor_stage(plan.h, st, ctypes.byref(cst), _ptr(f["L"]), _ptr(f["y"]), _ptr(plan.ws), _ptr(plan.info),
                                                _stream()) == 0

            lazy = model._q is not None and model._q["mu"] is None     # the step does not write the [B, T] marginal arrays

            def backward():
                assert lib.mfgm_cq_selinv_kl(plan.h, 0, ctypes.byref(cst), _ptr(f["L"]), _ptr(f["y"]), ctypes.byref(model._sde_prm),
                                             None if lazy else _ptr(s["Sig"]), None if lazy else _ptr(s["x"]), _ptr(klbuf),
                                             _ptr(model.fx_mus_obs), _ptr(model.fx_covs_obs), _ptr(plan.ws), _stream()) == 0

            def girsanov():
                assert lib.mfgm_cq_selinv_girsanov(plan.h, 0, ctypes.byref(cst), _ptr(f["L"]), _ptr(f["y"]), ctypes.byref(model._sde_prm),
                                                   _ptr(cq.spare), _ptr(plan.ws), _stream()) == 0

            E3, slot = 3 * d, 0.5      # the int slot per node counts as half a double
            # the cache-policy variant of the sweeps this plan runs (mfgm_plan_create: streamed when one d x d array is >= 128 MB;
            # MFGM_NT overrides) -- part of the kernels' names in the profiler's tables
            nt = int(os.environ.get("MFGM_NT", 2 if 8.0 * B * T * d * d >= 128 * 2 ** 20 else 0))
            ntv = 2 if nt >= 2 else 0
            cand = [
                (f"void mfgm::k_forward_cq<{d}, {ntv}>(mfgm::SweepArgs, mfgm::CqArgs)", 2, (E3 + slot) + (ET + d), lambda: stage(1),
                 "level 0 forward: block Cholesky + forward substitution; reads the cq record and the slot, writes L and y"),
                (f"void mfgm::k_reduce_cq<{d}>(mfgm::SweepArgs, mfgm::CqArgs)", 2, (E3 + slot), lambda: stage(0),
                 "level 0 reduce: segment elimination; reads the cq record and the slot"
                 + (" (timed alone; in the pipelined loop one of the step's two reduces runs as k_reduce_cq_lean on a second stream, "
                    "next to the forward sweep of the previous factorisation: share_of_step counts it as if it ran alone)"
                    if model.pipelined else "")),
                (f"void mfgm::k_backward_kl_cq<{d}, {ntv}>(mfgm::SweepArgs, mfgm::SdeParams, mfgm::CqArgs)", 1,
                 (ET + d + d + slot) + (0 if lazy else ET + d), backward,
                 "level 0 backward + KL sum: selected inverse, back-substitution, E_q[log p]; reads L, y, diag theta_sub, slot; writes "
                 + ("the marginals at the observation nodes only" if lazy else "Sigma, mu")),
                (f"void mfgm::k_backward_girsanov_cq<{d}, {ntv}>(mfgm::SweepArgs, mfgm::SdeParams, mfgm::CqArgs, double*)", 1, (ET + d + E3) + E3, girsanov,
                 "level 0 backward fused with the Girsanov-site update; reads L, y, the cq record, writes the new record"),
            ]
        else:
            tq, sp = model.full_sites(), getattr(model, "_theta_spare", None)
            fused = sp is not None

            def stage(st):
                # level-0 reduce (0) / forward (1) of the model's own factorisation (L_{t+1,t} not stored: G = NULL)
                assert lib.mfgm_packed_factor_stage(plan.h, st, 0, _ptr(tq.diag), _ptr(tq.sub), _ptr(tq.lin), -2.0, -1.0, 1.0, _ptr(f["L"]),
                                                    null, _ptr(f["y"]), _ptr(plan.ws), _ptr(plan.info), _stream()) == 0

            def backward():
                # level-0 backward of the refresh before the ELBO: reads theta_sub in place of L_{t+1,t}; fused: the KL sum is taken in
                # the sweep and no moment array is written
                if fused:
                    assert lib.mfgm_packed_selinv_kl(plan.h, 0, _ptr(f["L"]), _ptr(tq.sub), -1.0, _ptr(f["y"]), ctypes.byref(model._sde_prm),
                                                     _ptr(s["Sig"]), _ptr(s["x"]), _ptr(klbuf), _ptr(plan.ws), _stream()) == 0
                else:
                    assert lib.mfgm_packed_selinv_mom_s(plan.h, 0, _ptr(f["L"]), _ptr(tq.sub), -1.0, _ptr(f["y"]), _ptr(s["Sig"]), _ptr(s["x"]),
                                                        _ptr(s["mom"]), _ptr(plan.ws), _stream()) == 0

            def girsanov():
                # level-0 backward fused with the Girsanov-site update; writes the spare theta_q buffers (the model's state is untouched)
                assert lib.mfgm_packed_selinv_girsanov(plan.h, 0, _ptr(f["L"]), _ptr(tq.sub), -1.0, _ptr(f["y"]), ctypes.byref(model._sde_prm),
                                                       _ptr(tq.lin), _ptr(tq.diag), _ptr(sp.lin), _ptr(sp.diag), _ptr(sp.sub),
                                                       _ptr(plan.ws), _stream()) == 0

            # (kernel, launches per step, algorithmic doubles per node read + written, launcher)
            cand = [
                (f"void mfgm::k_forward<{d}, true, false, true>(mfgm::SweepArgs)", 2, (ET + EF + d) + (ET + d), lambda: stage(1),
                 "level 0 forward: block Cholesky + forward substitution; reads theta_q, writes L and y"),
                (f"void mfgm::k_reduce<{d}, true, false>(mfgm::SweepArgs)", 2, (ET + EF + d), lambda: stage(0),
                 "level 0 reduce: segment elimination; reads theta_q"),
                ((f"void mfgm::k_backward_kl<{d}>(mfgm::SweepArgs, mfgm::SdeParams)", 1, (ET + EF + d) + (ET + d), backward,
                  "level 0 backward + KL sum: selected inverse, back-substitution, E_q[log p]; reads L, theta_sub, y, writes Sigma, mu") if fused else
                 (f"void mfgm::k_backward<{d}, true, true, false, true, true>(mfgm::SweepArgs)", 2, (ET + EF + d) + (ET + d + 3 * d), backward,
                  "level 0 backward: selected inverse + back-substitution; reads L, theta_sub, y, writes Sigma, mu, moments")),
            ]
            if fused:
                cand.append((f"void mfgm::k_backward_girsanov<{d}>(mfgm::SweepArgs, mfgm::SdeParams, mfgm::GirsanovArgs)", 1,
                             (ET + EF + d) + (d + ET) + (d + ET + EF), girsanov,
                             "level 0 backward fused with the Girsanov-site update; reads L, y, theta_q, writes the new theta_q"))
        rows = []
        for kname, per_step, doubles, fn, what in cand:
            k_ms = timed(fn)
            alg_bytes = 8 * doubles * B * T
            ach = alg_bytes / (k_ms * 1e-3) / 1e9
            traffic, tsrc = pmc_traffic(kname, B, T, d, lib.mfgm_version().decode())
            rows.append({"bound": "hbm", "kernel": f"{kname} ({what})", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc, "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes, "launches_per_step": per_step,
                         "share_of_step": per_step * k_ms / ms_per_step})
        rows.sort(key=lambda r: -r["share_of_step"])
        out["roofline"] = dict(rows[0], other_kernels=rows[1:])
        # the whole step against SURVEY 8d's own byte count for it (full CVI step: 8 (11 d^2 + 5 d) B per trajectory-time-step, dense
        # naturals): 21.8 GB at the headline size, i.e. 2.73 ms at the HBM peak
        step_bytes = 8 * (11 * d * d + 5 * d) * B * T
        # what the step actually moves: the algorithmic bytes of its level-0 launches (PMC traffic agrees to 3 %, profiles/); the coarse
        # levels and the small kernels add < 1 GB
        actual = sum(r["algorithmic_bytes_per_launch"] * r["launches_per_step"] for r in rows)
        level0_share = sum(r["share_of_step"] for r in rows)
        out["roofline"]["step"] = {
            "actual_bytes": actual, "actual_GBps": actual / (ms_per_step * 1e-3) / 1e9, "actual_frac": actual / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "survey_dense_bytes_per_step": step_bytes,
            # a normalised speed, NOT an HBM utilisation: SURVEY 8d's byte count assumes dense naturals, which the cq state does not move
            "effective_vs_survey_dense_bytes": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "level0_share_of_step": level0_share,
            "coarse_levels_and_small_kernels_share_of_step": 1.0 - level0_share}
        if cq is not None:
            # the coarse levels (one launch per pass: separator systems of 1021 -> 4 nodes per chain, latency-bound), timed directly:
            # the library's coarse-only profiling stages on the workspace the last level-0 reduce / the last factorisation left
            stage(0)
            coarse_f = timed(lambda: stage(2))
            plan.cq_factor(cq, want_logdet=True, out=f)

            def coarse_selinv():
                assert lib.mfgm_cq_selinv_kl(plan.h, 1, ctypes.byref(cst), _ptr(f["L"]), _ptr(f["y"]), ctypes.byref(model._sde_prm),
                                             None if lazy else _ptr(s["Sig"]), None if lazy else _ptr(s["x"]), _ptr(klbuf),
                                             _ptr(model.fx_mus_obs), _ptr(model.fx_covs_obs), _ptr(plan.ws), _stream()) == 0
            coarse_b = timed(coarse_selinv)
            out["roofline"]["step"].update(coarse_factor_ms_per_refresh=coarse_f, coarse_backward_ms_per_refresh=coarse_b,
                                           coarse_levels_share_of_step=2 * (coarse_f + coarse_b) / ms_per_step)
        if getattr(args, "through_trainer", False) and args.config in ("headline", "c1"):
            out["trainer"] = trainer_rate(model, args)
        if world == 1 and not args.no_vdp and args.config == "headline":
            # free the CVI-DP state first: the VDP model keeps its own ~15 GB resident
            del model, f, s, cand, cq
            tq = sp = cst = None
            torch.cuda.empty_cache()
            out["vdp"] = vdp_step_rate(B, T, d, dt, noise, idx, ys, device)
        if world == 1 and not args.no_cpu_baseline:
            try:
                if args.config == "c1":
                    out["cpu_baseline"] = cpu_baseline(args, idx, ys, dt, noise, first_elbo[0].cpu().numpy(),
                                                       drift=(1.0 - 1.2 * dt, 0.0, 1.0 / 2.4), Lc=np.array([[noise]]))
                else:
                    out["cpu_baseline"] = cpu_baseline(args, idx, ys, dt, noise, first_elbo[0].cpu().numpy())
            except OSError as e:  # library not built
                out["cpu_baseline"] = {"value": None, "unit": "ELBO steps/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return out



def trainer_rate(model, args):
    """The same loop driven by the TRAINER (vidp_amd.trainers.CVISitesTrainer._optimize_sites_under_stable_prior =
    cvi_dp_trainer.py:63-95: update_data_sites, update_girsanov_sites, classic_elbo, learning-rate decay and convergence rules),
    continuing from the state the bare loop left: steps/s with the trainer's own control flow around the kernels."""
    import torch
    from vidp_amd.trainers import CVISitesTrainer
    k = max(4, args.steps)
    tr = CVISitesTrainer(model, max_itr_sites_optim=k, optim_tol=0.0, data_sites_lr=args.lr_data, girsanov_sites_lr=args.lr_girsanov,
                         sync_every=args.trainer_sync_every)
    tr._optimize_sites_under_stable_prior()                 # warm-up pass of k iterations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    elbos, _, _ = tr._optimize_sites_under_stable_prior()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": len(elbos) / el, "unit": "ELBO steps/s", "ms_per_step": 1e3 * el / len(elbos), "iterations": len(elbos),
            "sync_every": tr.sync_every, "elbo_last": float(elbos[-1]), "learning_rates": [tr.data_sites_lr, tr.girsanov_sites_lr],
            "workload": "CVISitesTrainer._optimize_sites_under_stable_prior (cvi_dp_trainer.py:63-95) on the same model: the bare "
                        "loop plus the trainer's ELBO bookkeeping, learning-rate decay and convergence checks"}


def other_configs(args, harness, data_rank):
    """BASELINE.json's other configurations, each as a short run of the same contract (value, ms_per_step, roofline, cpu_baseline), so
    that the driver's one default invocation records them all: c1 (the reference's CPU-runnable case), c2, c3 and c5 on one GPU."""
    import copy
    import gc
    import torch
    res = {}
    for name in ("c1", "c2", "c3", "c5", "headline_dense"):
        a = copy.copy(args)
        a.config, a.steps, a.warmup, a.no_vdp = name, args.other_steps, 2, True
        a.B, a.T, a.d, a.lr_data, a.lr_girsanov = 64, 100000, 6, 0.5, 0.1      # (c1 / c3 set their own sizes)
        a.dense = a.through_trainer = False
        if name == "headline_dense":
            # the headline workload on the dense-naturals route (no cq state): the path of every model without that structure
            a.config, a.dense, a.no_cpu_baseline = "headline", True, True
        h = Harness(a, harness.rank, harness.world, harness.device, harness.dist, harness.vdist)
        fn = bench_cvidp if name in ("c1", "headline_dense") else OTHER_CONFIGS[name]
        try:
            o = fn(h, data_rank)
            res[name] = {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "elbo_last", "roofline", "cpu_baseline")
                         if k in o}
        except Exception as e:      # one configuration failing must not lose the headline line
            res[name] = {"error": f"{type(e).__name__}: {e}"}
        gc.collect()
        torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--B", type=int, default=64, help="trajectories per GPU")
    ap.add_argument("--T", type=int, default=100000)
    ap.add_argument("--d", type=int, default=6)
    ap.add_argument("--lr-data", type=float, default=0.5)
    ap.add_argument("--lr-girsanov", type=float, default=0.1)
    ap.add_argument("--obs-every", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vdp", action="store_true", help="skip the secondary VDP step measurement")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="headline run only: do not append the short runs of BASELINE.json's other configurations (other_configs)")
    ap.add_argument("--dense", action="store_true", help="CVI-DP on the dense-naturals route (the cq state off, as VIDP_CQ=0)")
    ap.add_argument("--through-trainer", action="store_true",
                    help="also time the loop through CVISitesTrainer (cvi_dp_trainer.py:63-95) on the same model: `trainer` in the line")
    ap.add_argument("--trainer-sync-every", type=int, default=8, help="iterations between host synchronisations of the trainer loop")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps of each configuration under other_configs")
    ap.add_argument("--config", default="headline", choices=["headline", "c1", "c2", "c3", "c5"],
                    help="BASELINE.json configuration (default: the size the metric is quoted on)")
    ap.add_argument("--c5-M", type=int, default=200000, help="inducing states of config c5 (tests use a smaller chain)")
    ap.add_argument("--data-rank", type=int, default=None,
                    help="generate the synthetic trajectories of this rank (default: the process's own rank); lets a single-rank run "
                         "reproduce one shard of a multi-rank run")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)          # does not return
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != env_world:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {env_world} rank(s); refusing to report a mislabelled number",
              file=sys.stderr)
        sys.exit(2)

    import torch
    import vidp_amd
    from vidp_amd import distributed as vdist
    # "nccl" is RCCL on ROCm; VIDP_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path only)
    rank, world = vdist.init_from_env(backend=os.environ.get("VIDP_DIST_BACKEND", "nccl"))
    if world > 1:
        import torch.distributed as dist
    else:
        dist = None
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())

    data_rank = rank if args.data_rank is None else args.data_rank
    harness = Harness(args, rank, world, device, dist, vdist)
    fn = bench_cvidp if args.config in ("headline", "c1") else OTHER_CONFIGS[args.config]
    out = fn(harness, data_rank)
    if rank == 0 and world == 1 and args.config == "headline" and not args.no_other_configs:
        out["other_configs"] = other_configs(args, harness, data_rank)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()



if __name__ == "__main__":
    main()
