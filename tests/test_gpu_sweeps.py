"""
GPU parity tests (run with -m gpu on an MI355X): the HIP sweeps, called through the C ABI, against the
NumPy oracle on the same seeded inputs.  Tolerance: fp64, 1e-7 relative on random blocks (partitioned elimination order differs from the sequential oracle)
(the north-star bound is 1e-5 relative).
"""
import numpy as np
import pytest

from oracle import np_btd
from tests.helpers import assert_close, random_dominant_btd, random_spd_btd

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-7, 1e-9


@pytest.fixture(scope="module")
def amd():
    import torch
    import vidp_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    vidp_amd._lib.load()
    return vidp_amd


def _dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("B,T,d,R0", [(1, 1, 1, 0), (3, 17, 3, 4), (2, 130, 6, 8), (5, 64, 2, 64), (70, 9, 4, 2)])
def test_pack_round_trip(amd, rng, kind, B, T, d, R0):
    plan = amd.Plan(B, T, d, R0=R0)
    for n_nodes in (T, T - 1):
        if n_nodes < 1:
            continue
        shape = (B, n_nodes, d) if kind == 0 else (B, n_nodes, d, d)
        x = rng.normal(size=shape)
        back = plan.unpack(kind, plan.pack(kind, _dev(x)), n_nodes).cpu().numpy()
        if kind == 2:
            low = np.tril(x)
            ref = low + np.swapaxes(np.tril(x, -1), -1, -2)
        elif kind == 3:
            ref = np.tril(x)
        else:
            ref = x
        np.testing.assert_array_equal(back, ref)


CASES = [
    # B, T, d, R0, Rup
    (1, 1, 1, 0, 0),
    (1, 4, 1, 0, 0),
    (3, 4, 3, 0, 0),
    (2, 5, 2, 2, 2),        # many tiny levels
    (3, 37, 3, 4, 3),       # ragged last segment, 3+ levels
    (4, 64, 5, 8, 4),
    (2, 129, 6, 8, 8),
    (65, 20, 4, 5, 0),      # more than one wavefront of lanes
    (1, 1000, 1, 0, 0),     # config-1 size
    (2, 300, 6, 16, 0),
    (1, 200, 8, 10, 0),
    (1, 150, 7, 10, 0),
]


@pytest.mark.parametrize("B,T,d,R0,Rup", CASES)
@pytest.mark.parametrize("with_rhs", [True, False])
def test_factor_and_selinv(amd, rng, B, T, d, R0, Rup, with_rhs):
    if T <= 5:
        diag, sub, _, _ = random_spd_btd(rng, (B,), T, d)   # the reference's own generator (KA1)
    else:
        diag, sub = random_dominant_btd(rng, (B,), T, d)
    r = rng.normal(size=(B, T, d))
    plan = amd.Plan(B, T, d, R0=R0, Rup=Rup)
    Dp = plan.pack(amd.SYM, _dev(diag))
    Sp = plan.pack(amd.FULL, _dev(sub)) if T > 1 else plan.zeros(amd.FULL)
    rp = plan.pack(amd.VEC, _dev(r)) if with_rhs else None
    f = plan.factor(Dp, Sp, rp, want_logdet=True, want_quad=True)
    plan.check_info()

    Ld, Ls = np_btd.cholesky(diag, sub)
    assert_close(plan.unpack(amd.TRI, f["L"]).cpu().numpy(), Ld)
    if T > 1:
        assert_close(plan.unpack(amd.FULL, f["G"], T - 1).cpu().numpy(), Ls)
    np.testing.assert_allclose(f["logdet"].cpu().numpy(), np_btd.abs_log_det(Ld), rtol=1e-8, atol=1e-8)
    if with_rhs:
        y = np_btd.solve(Ld, Ls, r)
        assert_close(plan.unpack(amd.VEC, f["y"]).cpu().numpy(), y)
        np.testing.assert_allclose(f["quad"].cpu().numpy(), np.sum(y * y, axis=(-1, -2)), rtol=1e-7)

    s = plan.selinv(f["L"], f["G"], f["y"], want_sub=True)
    Sd, Ss = np_btd.inverse_blocks(Ld, Ls)
    assert_close(plan.unpack(amd.SYM, s["Sig"]).cpu().numpy(), Sd)
    if T > 1:
        assert_close(plan.unpack(amd.FULL, s["Sub"], T - 1).cpu().numpy(), Ss)
    if with_rhs:
        x = np_btd.solve(Ld, Ls, np_btd.solve(Ld, Ls, r), transpose_left=True)
        assert_close(plan.unpack(amd.VEC, s["x"]).cpu().numpy(), x)


def test_scaled_inputs_and_not_pd(amd, rng):
    """aD / aS / aR scale on load (natural parameters -> precision), and the non-PD flag."""
    B, T, d = 2, 40, 3
    diag, sub = random_dominant_btd(rng, (B,), T, d)
    r = rng.normal(size=(B, T, d))
    plan = amd.Plan(B, T, d, R0=8)
    f = plan.factor(plan.pack(amd.SYM, _dev(-0.5 * diag)), plan.pack(amd.FULL, _dev(-sub)), plan.pack(amd.VEC, _dev(2 * r)),
                    aD=-2.0, aS=-1.0, aR=0.5)
    plan.check_info()
    Ld, Ls = np_btd.cholesky(diag, sub)
    assert_close(plan.unpack(amd.TRI, f["L"]).cpu().numpy(), Ld)
    assert_close(plan.unpack(amd.VEC, f["y"]).cpu().numpy(), np_btd.solve(Ld, Ls, r))
    plan.factor(plan.pack(amd.SYM, _dev(-diag)), plan.pack(amd.FULL, _dev(sub)))
    with pytest.raises(ArithmeticError):
        plan.check_info()


def test_partition_invariance_large(amd, rng):
    """Size-independent property at a size the dense oracle cannot reach: different partitions agree, and
    K * (K^{-1} r) == r through the block-tri-diagonal product."""
    B, T, d = 4, 20000, 6
    diag, sub = random_dominant_btd(rng, (B,), T, d)
    r = rng.normal(size=(B, T, d))
    outs = []
    for R0, Rup in ((0, 0), (50, 8), (T, 0)):
        plan = amd.Plan(B, T, d, R0=R0, Rup=Rup)
        f = plan.factor(plan.pack(amd.SYM, _dev(diag)), plan.pack(amd.FULL, _dev(sub)), plan.pack(amd.VEC, _dev(r)))
        plan.check_info()
        s = plan.selinv(f["L"], f["G"], f["y"])
        outs.append((f["logdet"].cpu().numpy(), plan.unpack(amd.VEC, s["x"]).cpu().numpy(),
                     plan.unpack(amd.SYM, s["Sig"]).cpu().numpy(), plan.unpack(amd.FULL, s["Sub"], T - 1).cpu().numpy()))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert_close(a, b)
    x = outs[0][1]
    np.testing.assert_allclose(np_btd.dense_mult(diag, sub, x, symmetric=True), r, rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("B,T,d", [(3, 57, 3), (2, 41, 12), (1, 30, 20)])
def test_natural_layout_entry_points(amd, rng, B, T, d):
    """mfgm_btd_cholesky / mfgm_btd_posterior: the natural-layout C entry points a TF custom-op kernel would call
    (lane-per-segment, MFMA and row-per-lane kernel families)."""
    import torch
    from vidp_amd.packed import _ptr, _stream
    diag, sub = random_dominant_btd(rng, (B,), T, d)
    r = rng.normal(size=(B, T, d))
    plan = amd.Plan(B, T, d, R0=8, Rup=4)
    lib = plan.lib
    nws = torch.empty(lib.mfgm_natural_workspace_bytes(plan.h) // 8, dtype=torch.float64, device="cuda")
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
    Ld, Ls, logdet = z(B, T, d, d), z(B, T - 1, d, d), z(B)
    # natural parameters in, precision on load: theta_diag = -K/2, theta_sub = -K_sub
    th_d, th_s, dg, sb, rr = _dev(-0.5 * diag), _dev(-sub), _dev(diag), _dev(sub), _dev(r)   # keep the inputs alive across the calls
    rc = lib.mfgm_btd_cholesky(plan.h, _ptr(th_d), _ptr(th_s), -2.0, -1.0, _ptr(Ld), _ptr(Ls), _ptr(logdet), _ptr(nws),
                               _ptr(plan.info), _stream())
    assert rc == 0
    plan.check_info()
    oLd, oLs = np_btd.cholesky(diag, sub)
    assert_close(Ld.cpu().numpy(), oLd)
    assert_close(Ls.cpu().numpy(), oLs)
    assert_close(logdet.cpu().numpy(), np_btd.abs_log_det(oLd))
    x, Sd, Ss = z(B, T, d), z(B, T, d, d), z(B, T - 1, d, d)
    rc = lib.mfgm_btd_posterior(plan.h, _ptr(dg), _ptr(sb), _ptr(rr), 1.0, 1.0, 1.0, _ptr(logdet), _ptr(x), _ptr(Sd),
                                _ptr(Ss), _ptr(nws), _ptr(plan.info), _stream())
    assert rc == 0
    oSd, oSs = np_btd.inverse_blocks(oLd, oLs)
    assert_close(Sd.cpu().numpy(), oSd)
    assert_close(Ss.cpu().numpy(), oSs)
    assert_close(x.cpu().numpy(), np_btd.solve(oLd, oLs, np_btd.solve(oLd, oLs, r), transpose_left=True))


@pytest.mark.parametrize("name,B,T,d", [("C2", 1, 100000, 3), ("C3", 64, 50000, 6), ("headline", 64, 100000, 6), ("C5", 1, 200000, 16)])
def test_full_size_properties(amd, name, B, T, d):
    """BASELINE.json's full sizes, through size-independent properties (inputs generated on the device):
    K (K^{-1} r) = r,  tr(K Sigma) = sum_t <D_t, Sigma_tt> + 2 <S_t, Sigma_{t+1,t}> = T d  (only the selected-inverse blocks enter),
    log det and the solve independent of the time partition, and L L^T reproducing the diagonal blocks."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(71892305)
    rnd = lambda *s: torch.randn(*s, generator=g, device="cuda", dtype=torch.float64)
    D = 0.1 * rnd(B, T, d, d)
    D = D + D.transpose(-1, -2) + (4.0 + d ** 0.5) * torch.eye(d, device="cuda", dtype=torch.float64)
    S = (0.3 / d ** 0.5) * rnd(B, T - 1, d, d)
    r = rnd(B, T, d)
    plan = amd.Plan(B, T, d)
    Dp, Sp, rp = plan.pack(amd.SYM, D), plan.pack(amd.FULL, S), plan.pack(amd.VEC, r)
    f = plan.factor(Dp, Sp, rp, want_logdet=True, want_quad=True)
    s = plan.selinv(f["L"], f["G"], f["y"], want_sub=True)
    plan.check_info()
    x, Sig, Sub = plan.unpack(amd.VEC, s["x"]), plan.unpack(amd.SYM, s["Sig"]), plan.unpack(amd.FULL, s["Sub"], T - 1)
    # K x = r
    Kx = (D @ x[..., None])[..., 0]
    Kx[:, 1:] += (S @ x[:, :-1, :, None])[..., 0]
    Kx[:, :-1] += (S.transpose(-1, -2) @ x[:, 1:, :, None])[..., 0]
    assert float((Kx - r).abs().max()) < 1e-10 * float(r.abs().max()) * 100
    # tr(K Sigma) = T d per chain
    tr = (D * Sig).sum(dim=(1, 2, 3)) + 2.0 * (S * Sub).sum(dim=(1, 2, 3))
    np.testing.assert_allclose(tr.cpu().numpy(), T * d, rtol=1e-10)
    # r^T K^{-1} r = |L^{-1} r|^2
    np.testing.assert_allclose((r * x).sum(dim=(1, 2)).cpu().numpy(), f["quad"].cpu().numpy(), rtol=1e-10)
    # the factor reproduces the diagonal blocks: L_tt L_tt^T + L_{t,t-1} L_{t,t-1}^T = D_t
    L, G = plan.unpack(amd.TRI, f["L"]), plan.unpack(amd.FULL, f["G"], T - 1)
    rec = L @ L.transpose(-1, -2)
    rec[:, 1:] += G @ G.transpose(-1, -2)
    assert float((rec - D).abs().max()) < 1e-12 * float(D.abs().max()) * 100
    del rec, L, G, Kx
    if d > 8:
        # the inverse-form sweeps (mfgm_packed_factor_form / _selinv_form, form 1) deliver the same moments
        f3 = plan.factor(Dp, Sp, rp, want_logdet=True, want_quad=True, moments_only=True)
        s3 = plan.selinv(f3["L"], f3["G"], f3["y"], want_sub=True, form=f3["form"])
        plan.check_info()
        assert f3["form"] == 1
        np.testing.assert_allclose(f3["logdet"].cpu().numpy(), f["logdet"].cpu().numpy(), rtol=1e-12)
        np.testing.assert_allclose(f3["quad"].cpu().numpy(), f["quad"].cpu().numpy(), rtol=1e-11)
        assert float((plan.unpack(amd.VEC, s3["x"]) - x).abs().max()) < 1e-11
        assert float((plan.unpack(amd.SYM, s3["Sig"]) - Sig).abs().max()) < 1e-11
        assert float((plan.unpack(amd.FULL, s3["Sub"], T - 1) - Sub).abs().max()) < 1e-11
        del f3, s3
    # another time partition gives the same log-determinant and solve
    plan2 = amd.Plan(B, T, d, R0=max(8, plan.R * 3 + 1), Rup=5)
    f2 = plan2.factor(plan2.pack(amd.SYM, D), plan2.pack(amd.FULL, S), plan2.pack(amd.VEC, r), want_logdet=True)
    s2 = plan2.selinv(f2["L"], f2["G"], f2["y"], want_sub=False)
    plan2.check_info()
    np.testing.assert_allclose(f2["logdet"].cpu().numpy(), f["logdet"].cpu().numpy(), rtol=1e-12)
    assert float((plan2.unpack(amd.VEC, s2["x"]) - x).abs().max()) < 1e-11


def test_random_partitions(amd):
    """Randomised shapes and partitions (all three kernel families): chains shorter than a segment, one segment per chain,
    ragged last segments on several levels, Rup = 2, more chains than lanes in a wavefront -- against the oracle."""
    rs = np.random.default_rng(71892305 + 17)
    cases = []
    for _ in range(40):
        d = int(rs.integers(1, 9))
        cases.append((int(rs.integers(1, 6)), int(rs.integers(1, 90)), d, int(rs.integers(0, 20)), int(rs.integers(0, 7))))
    for _ in range(14):
        d = int(rs.integers(9, 33))
        cases.append((int(rs.integers(1, 4)), int(rs.integers(1, 50)), d, int(rs.integers(0, 12)), int(rs.integers(0, 5))))
    cases += [(130, 3, 2, 2, 2), (1, 2, 8, 1, 2), (1, 49, 3, 0, 0), (2, 97, 16, 2, 2)]
    for B, T, d, R0, Rup in cases:
        diag, sub = random_dominant_btd(rs, (B,), T, d)
        r = rs.normal(size=(B, T, d))
        plan = amd.Plan(B, T, d, R0=R0, Rup=Rup)
        Sp = plan.pack(amd.FULL, _dev(sub)) if T > 1 else plan.zeros(amd.FULL)
        f = plan.factor(plan.pack(amd.SYM, _dev(diag)), Sp, plan.pack(amd.VEC, _dev(r)), want_logdet=True, want_quad=True)
        s = plan.selinv(f["L"], f["G"], f["y"], want_sub=True)
        plan.check_info()
        Ld, Ls = np_btd.cholesky(diag, sub)
        Sd, Ss = np_btd.inverse_blocks(Ld, Ls)
        ctx = f"B={B} T={T} d={d} R0={R0} Rup={Rup} levels={plan.nlevels}"
        try:
            assert_close(plan.unpack(amd.TRI, f["L"]).cpu().numpy(), Ld)
            np.testing.assert_allclose(f["logdet"].cpu().numpy(), np_btd.abs_log_det(Ld), rtol=1e-9, atol=1e-9)
            assert_close(plan.unpack(amd.SYM, s["Sig"]).cpu().numpy(), Sd)
            if T > 1:
                assert_close(plan.unpack(amd.FULL, f["G"], T - 1).cpu().numpy(), Ls)
                assert_close(plan.unpack(amd.FULL, s["Sub"], T - 1).cpu().numpy(), Ss)
            x = np_btd.solve(Ld, Ls, np_btd.solve(Ld, Ls, r), transpose_left=True)
            assert_close(plan.unpack(amd.VEC, s["x"]).cpu().numpy(), x)
            if d > 8:      # the same moments from the inverse-form sweeps
                f = plan.factor(plan.pack(amd.SYM, _dev(diag)), Sp, plan.pack(amd.VEC, _dev(r)), want_logdet=True, moments_only=True)
                s = plan.selinv(f["L"], f["G"], f["y"], want_sub=True, form=f["form"])
                plan.check_info()
                np.testing.assert_allclose(f["logdet"].cpu().numpy(), np_btd.abs_log_det(Ld), rtol=1e-9, atol=1e-9)
                assert_close(plan.unpack(amd.SYM, s["Sig"]).cpu().numpy(), Sd)
                assert_close(plan.unpack(amd.VEC, s["x"]).cpu().numpy(), x)
                if T > 1:
                    assert_close(plan.unpack(amd.FULL, s["Sub"], T - 1).cpu().numpy(), Ss)
        except AssertionError as e:
            raise AssertionError(ctx + "\n" + str(e)) from None


@pytest.mark.parametrize("d,moments_only", [(3, False), (6, False), (16, False), (16, True)])
def test_not_pd_report_names_the_first_failing_chain_and_nodes(amd, rng, d, moments_only):
    """SURVEY 8b's error contract: a pivot block that is not positive definite is reported with its location -- status 2 from
    mfgm_plan_check_info and (chain, node range) of the FIRST failure, not a bare flag (TF's Cholesky op fails the step in the reference,
    block_tri_diag.py:428-440).  One indefinite diagonal block is planted in chain 1 (and a later one in chain 2)."""
    import ctypes
    B, T = 3, 203
    diag, sub = random_dominant_btd(rng, (B,), T, d)
    diag[1, 77] = -np.eye(d)
    diag[2, 150] = -np.eye(d)
    plan = amd.Plan(B, T, d, R0=8)
    plan.factor(plan.pack(amd.SYM, _dev(diag)), plan.pack(amd.FULL, _dev(sub)), moments_only=moments_only)
    out = (ctypes.c_int * 4)()
    from vidp_amd.packed import _ptr, _stream
    assert plan.lib.mfgm_plan_check_info(plan.h, _ptr(plan.info), out, _stream()) == 2
    b, lo, hi, level = tuple(out)
    assert b == 1 and lo <= 77 < hi and hi - lo <= 8 and level == 0
    with pytest.raises(ArithmeticError, match=r"chain 1, a node in \[72, 80\)") as ei:
        plan.check_info()
    assert ei.value.location[:3] == (1, 72, 80)
    # the word is cleared: a clean factorisation on the same plan passes, status 0
    diag2, sub2 = random_dominant_btd(rng, (B,), T, d)
    plan.factor(plan.pack(amd.SYM, _dev(diag2)), plan.pack(amd.FULL, _dev(sub2)), moments_only=moments_only)
    assert plan.lib.mfgm_plan_check_info(plan.h, _ptr(plan.info), out, _stream()) == 0
    plan.check_info()


@pytest.mark.parametrize("B,T,d,R0,Rup", [(3, 700, 6, 7, 4), (2, 333, 8, 5, 3), (5, 260, 1, 4, 4), (2, 500, 3, 6, 2), (1, 900, 7, 9, 4),
                                           (4, 150, 2, 3, 4), (3, 410, 4, 5, 4), (2, 290, 5, 4, 5)])
def test_coarse_row_bodies_agree_with_lane_bodies(amd, rng, monkeypatch, B, T, d, R0, Rup):
    """Levels whose chains have at most 16 segments run, inside the fused coarse kernels, on 16 lanes per segment (mfgm_rows.h); with
    MFGM_COARSE_ROWS=0 they run on the lane-per-segment bodies the larger levels use.  Same arrays in, same arrays out: every output of
    the factorisation and of the selected inverse must agree to rounding (and both with the oracle, by the tests above)."""
    diag, sub = random_dominant_btd(rng, (B,), T, d)
    r = rng.normal(size=(B, T, d))
    plan = amd.Plan(B, T, d, R0=R0, Rup=Rup)
    assert plan.nlevels >= 3
    Dp, Sp, rp = plan.pack(amd.SYM, _dev(diag)), plan.pack(amd.FULL, _dev(sub)), plan.pack(amd.VEC, _dev(r))
    outs = []
    for rows in ("16", "0"):
        monkeypatch.setenv("MFGM_COARSE_ROWS", rows)
        f = plan.factor(Dp, Sp, rp, want_logdet=True, want_quad=True)
        plan.check_info()
        s = plan.selinv(f["L"], f["G"], f["y"], want_sub=True)
        outs.append([plan.unpack(amd.TRI, f["L"]), plan.unpack(amd.FULL, f["G"], T - 1), plan.unpack(amd.VEC, f["y"]), f["logdet"].clone(),
                     f["quad"].clone(), plan.unpack(amd.SYM, s["Sig"]), plan.unpack(amd.FULL, s["Sub"], T - 1), plan.unpack(amd.VEC, s["x"])])
    differs = False
    for a, b in zip(*outs):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-9, atol=1e-11)
        differs |= bool((a != b).any())
    import os
    fused_off = os.environ.get("MFGM_COARSE_FUSED") == "0"       # the row bodies live in the fused kernels
    assert differs or d == 1 or fused_off, "the two routes produced bit-identical arrays: the row bodies did not run"   # 1 x 1 blocks: same operations
    Ld, Ls = np_btd.cholesky(diag, sub)
    assert_close(outs[0][0].cpu().numpy(), Ld)
