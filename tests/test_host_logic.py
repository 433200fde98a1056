"""
Host-side logic that needs no GPU: parameter blocks handed to the C ABI, kernel specifications, the experiment file schema,
the trainers' Adam step, shard arithmetic of the time-sharded chain.
"""
import math

import numpy as np
import pytest
import torch


def test_sde_parameter_block():
    from vidp_amd import sde
    dw = sde.DoubleWellSDE(q=torch.diag(torch.tensor([0.5, 2.0], dtype=torch.float64)), scale=3.0, c=0.8, scale_trainable=True)
    prm = dw.params(0.01, np.array([0.1, -0.2]), np.array([[2.0, 0.3], [0.3, 1.0]]), lr=0.25, clip=(-1.0, 1.0))
    assert prm.kind == 0 and prm.dt == 0.01
    np.testing.assert_allclose([prm.alpha[0], prm.beta[0]], [1 + 0.01 * 3.0 * 0.8, 0.01 * 3.0])
    np.testing.assert_allclose([prm.W[0], prm.W[1]], [1 / (0.01 * 0.5), 1 / (0.01 * 2.0)])
    P0inv = np.linalg.inv(np.array([[2.0, 0.3], [0.3, 1.0]]))
    np.testing.assert_allclose([prm.P0inv[0], prm.P0inv[1], prm.P0inv[2]], [P0inv[0, 0], P0inv[1, 0], P0inv[1, 1]])
    np.testing.assert_allclose(prm.logdetQp, math.log(0.005) + math.log(0.02))
    assert (prm.lr, prm.clip_lo, prm.clip_hi) == (0.25, -1.0, 1.0)
    assert dw.trainable_variables == ["scale"]
    np.testing.assert_allclose(dw.cubic_jacobian(0.01)["scale"], (0.01 * 0.8, 0.01))
    np.testing.assert_allclose(dw.drift_cubic_jacobian()["c"], (3.0, 0.0))
    ou = sde.OrnsteinUhlenbeckSDE(1.5, torch.eye(1, dtype=torch.float64), trainable=True)
    assert ou.cubic(0.1) == (1 - 0.15, 0.0) and ou.trainable_variables == ["decay"]
    b = sde.BenesSDE(1.3)
    assert b.kind == 1 and b.params(0.02, np.zeros(1), np.eye(1)).theta[0] == 1.3
    with pytest.raises(NotImplementedError):
        b.cubic(0.02)
    full = sde.DoubleWellSDE(q=torch.tensor([[1.0, 0.2], [0.2, 1.0]], dtype=torch.float64))      # non-diagonal diffusion for d > 1
    with pytest.raises(ValueError):     # ... is not for the closed-form kernels
        full.params(0.01, np.zeros(2), np.eye(2))
    qp = full.quad_params(0.01, np.array([0.1, -0.2]), np.eye(2), clip=(-1.0, 1.0))     # but for the quadrature kernels
    Winv = np.linalg.inv(0.01 * np.array([[1.0, 0.2], [0.2, 1.0]]))
    assert (qp.kind, qp.d) == (12, 2) and list(qp.theta)[:2] == [4.0, 4.0]
    np.testing.assert_allclose([qp.W[0], qp.W[1], qp.W[2]], [Winv[0, 0], Winv[1, 0], Winv[1, 1]])
    np.testing.assert_allclose(qp.logdetQp, np.linalg.slogdet(0.01 * np.array([[1.0, 0.2], [0.2, 1.0]]))[1])
    vdp = sde.VanderPolOscillatorSDE(1.3, 0.9, trainable=True)
    assert vdp.quad_params(0.05, np.zeros(2), np.eye(2)).kind == 10 and vdp.trainable_variables == ["a", "tau"]
    mlp = sde.MLPDrift(seed=3)
    th, nh = mlp.quad_theta()
    assert nh == 3 and len(th) == 10 and mlp.quad_params(0.05, np.zeros(1), np.eye(1)).nh == 3
    mlp.assign("weights", np.arange(10.0))
    assert mlp.quad_theta()[0] == list(np.arange(10.0))
    with pytest.raises(ValueError):
        sde.SineDiffusionSDE(0.1, torch.eye(5, dtype=torch.float64))


def test_kernel_spec_and_closed_forms():
    from oracle import np_kernels
    from vidp_amd import kernels as K
    k = K.Sum([K.Matern52(0.7, 1.3), K.Matern12(0.4, 0.9)], jitter=1e-6)
    spec = k._spec()
    assert spec.ncomp == 2 and list(spec.order)[:2] == [3, 1] and list(spec.offset)[:2] == [0, 3] and spec.jitter == 1e-6
    np.testing.assert_allclose(spec.lam[0], math.sqrt(5.0) / 0.7)
    ok = np_kernels.Sum([np_kernels.Matern52(0.7, 1.3), np_kernels.Matern12(0.4, 0.9)])
    np.testing.assert_allclose(k.steady_state_covariance.numpy(), ok.steady_state_covariance(), rtol=1e-13)
    A, Q = k.transition_statistics_local(torch.tensor([0.05, 0.3], dtype=torch.float64))
    oA, oQ = ok.transition_statistics(np.array([0.05, 0.3]))
    np.testing.assert_allclose(A.numpy(), oA, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(Q.numpy() - 1e-6 * np.eye(4), oQ, rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        K.Matern32(-1.0, 1.0)


def test_exp_data_schema(tmp_path):
    from vidp_amd import exp_io
    path = str(tmp_path / "d.npz")
    grid = np.linspace(0, 1, 11)
    exp_io.save_exp_data(path, Q=np.eye(1), x0=np.ones(1), sigma=0.2, latent_process=np.zeros((11, 1)), observation_grid=grid[::2],
                         observations=np.ones((6, 1)), test_grid=grid[1::4], test_observations=np.zeros((3, 1)), time_grid=grid)
    assert sorted(np.load(path).files) == sorted(["Q", "x0", "sigma", "latent_process", "observation_grid", "observations", "test_grid",
                                                  "test_observations", "time_grid"])
    Q, x0, noise, latent, obs, tg, test = exp_io.load_exp_data(path, device="cpu")
    assert noise.shape == (1, 1) and noise.item() == 0.2 and obs[0].dtype == torch.float64 and tuple(test[1].shape) == (3, 1)


def test_adam_matches_the_documented_update():
    from vidp_amd.trainers import _Adam
    opt = _Adam(0.1, 2)
    x, m, v = [1.0, -2.0], [0.0, 0.0], [0.0, 0.0]
    for t, g in enumerate(([0.3, -0.1], [0.2, 0.4], [-0.5, 0.1]), start=1):
        want = []
        for i in range(2):
            m[i] = 0.9 * m[i] + 0.1 * g[i]
            v[i] = 0.999 * v[i] + 0.001 * g[i] ** 2
            a = 0.1 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
            want.append(x[i] - a * m[i] / (math.sqrt(v[i]) + 1e-7))
        x = opt.step(x, g)
        np.testing.assert_allclose(x, want, rtol=1e-14)


def test_shard_bounds_cover_every_segment():
    from vidp_amd.distributed import shard_bounds
    for total in (1, 7, 64, 1021):
        for world in (1, 2, 3, 8):
            if world > total:
                continue
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 3, 3)


def test_bench_refuses_a_mislabelled_world_size():
    """`bench.py --gpus N` must never print an n_gpus the launcher did not provide: with a torch.distributed environment of another
    size it exits non-zero before touching the GPU (and with none, it starts the N ranks itself: tests/test_gpu_api.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()


def test_one_chain_exchange_volume_at_config5():
    """SURVEY 8e / config 5 (one chain, T = 200 000, d = 16, 8 ranks): the only per-time-step data that crosses ranks is the input of
    the exchange level of the partition -- n_X x (3 d^2 + 2 d) doubles with n_X = 32 (the highest level with at least 4 nodes per rank,
    distributed.ChainShard), i.e. four interface-block triples per rank, 205 KB (host-side plan logic, no GPU needed)."""
    import ctypes
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = ctypes.CDLL(os.path.join(root, "vi-diffusion-processes_amd", "csrc", "libmfgm.so"))
    lib.mfgm_plan_create.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_void_p)]
    lib.mfgm_plan_level.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.mfgm_plan_set_shard_level.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.mfgm_plan_exchange_region.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    lib.mfgm_plan_describe.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.mfgm_plan_destroy.argtypes = [ctypes.c_void_p]
    T, d, world = 200000, 16, 8
    h = ctypes.c_void_p()
    assert lib.mfgm_plan_create(1, T, d, 0, 0, ctypes.byref(h)) == 0
    desc = (ctypes.c_int * 6)()
    lib.mfgm_plan_describe(h, desc)
    lev, levels = (ctypes.c_int * 4)(), []
    for l in range(desc[0]):
        assert lib.mfgm_plan_level(h, l, lev) == 0
        levels.append(tuple(lev))
    X = max(l for l in range(1, desc[0]) if levels[l][0] >= 4 * world)
    assert levels[X][0] == 32
    covered = 0
    for rank in range(world):
        base, rem = divmod(levels[X][0], world)
        lo = rank * base + min(rank, rem)
        hi = lo + base + (1 if rank < rem else 0)
        assert lib.mfgm_plan_set_shard_level(h, X, lo, hi) == 0
        off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        assert lib.mfgm_plan_exchange_region(h, ctypes.byref(off), ctypes.byref(cnt)) == 0
        assert cnt.value <= levels[X][0] * (3 * d * d + 2 * d), (cnt.value, levels[X][0] * (3 * d * d + 2 * d))
        covered += hi - lo
    assert covered == levels[X][0]
    assert lib.mfgm_plan_set_shard_level(h, desc[0], 0, 1) == 1          # no such level
    lib.mfgm_plan_destroy(h)


def test_observation_aligned_segment_length():
    """packed.aligned_segment_length / observation_period (host logic): equally spaced observations move the automatic level-0 segment
    length up to the next multiple of their spacing; irregular observations, small problems, d > 8 and costly multiples keep the
    automatic choice (0)."""
    import numpy as np
    import torch
    import vidp_amd  # noqa: F401
    from vidp_amd.packed import aligned_segment_length, observation_period
    assert observation_period(torch.arange(49, 100000, 50)) == 50
    assert observation_period(np.arange(5, 1000, 9)) == 9
    assert observation_period(torch.tensor([3, 8, 14])) == 0 and observation_period(torch.tensor([7])) == 0
    assert observation_period(torch.arange(49, 1000, 50).expand(4, -1)) == 50          # [B, n], the same grid for every trajectory
    assert observation_period(np.stack([np.arange(5, 100, 9), np.arange(6, 101, 9)])) == 0     # different grids: no common alignment
    assert observation_period(torch.arange(3, 40, 4)[None]) == 4
    assert aligned_segment_length(64, 100000, 6, 50) == 100          # headline: ceil(6.4e6 / 65536) = 98 -> 100
    assert aligned_segment_length(64, 50000, 6, 50) == 50            # config 3: 49 -> 50
    assert aligned_segment_length(64, 100000, 6, 7) == 98            # already a multiple
    assert aligned_segment_length(64, 100000, 6, 0) == 0 and aligned_segment_length(64, 100000, 6, 1) == 0
    assert aligned_segment_length(1, 1001, 1, 31) == 0               # small problem: the automatic partition
    assert aligned_segment_length(64, 100000, 16, 50) == 0           # wavefront-per-segment plans
    assert aligned_segment_length(64, 100000, 6, 97) == 0            # 194 would leave half the lanes idle


def test_no_shadowed_test_functions():
    """A second top-level `def test_x` in a module silently replaces the first (a Benes / sine parity test was lost that way in
    round 3): every test module must define each top-level name once."""
    import ast
    import glob
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "*.py"))):
        tree = ast.parse(open(path).read())
        seen = {}
        for node in tree.body:
            if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
                assert node.name not in seen, f"{os.path.basename(path)}: {node.name} defined at lines {seen[node.name]} and {node.lineno}"
                seen[node.name] = node.lineno


def test_general_inverse():
    """linalg.general_inverse / small_inverse (cofactors for d <= 3, Gauss-Jordan with partial pivoting above) on non-symmetric blocks,
    including ones whose leading entry is zero (pivoting needed)."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(here, "vi-diffusion-processes_amd", "linalg.py")).read()
    ns = {"torch": torch}
    exec(src[src.index("def small_inverse"):], ns)          # the two functions are plain torch: no GPU library needed
    g = torch.Generator().manual_seed(3)
    for d in (1, 2, 3, 4, 6, 8):
        M = torch.randn(5, d, d, generator=g, dtype=torch.float64) + 2.0 * torch.eye(d, dtype=torch.float64)
        if d > 1:
            M[0, 0, 0] = 0.0
        X = ns["general_inverse"](M)
        np.testing.assert_allclose((X @ M).numpy(), np.broadcast_to(np.eye(d), (5, d, d)), atol=1e-11)
        np.testing.assert_allclose(X.numpy(), np.linalg.inv(M.numpy()), rtol=1e-9, atol=1e-11)

