"""The plain-C CPU port (oracle/csrc) against the NumPy oracle: it is the cpu_baseline, so it must be right."""
import os

import numpy as np
import pytest

from oracle import c_ref, np_btd, np_models, np_ssm, np_transforms
from tests.helpers import random_dominant_btd, random_ssm_params

pytestmark = pytest.mark.skipif(not os.path.exists(c_ref.LIB), reason="oracle/csrc/libbtdref.so not built (run __graft_entry__.build())")


@pytest.mark.parametrize("T,d", [(1, 1), (7, 1), (12, 3), (30, 6)])
def test_c_btd(rng, T, d):
    diag, sub = random_dominant_btd(rng, (), T, d)
    sub = np.zeros((0, d, d)) if sub is None else sub
    Ld, Ls = c_ref.btd_cholesky(diag, sub)
    oLd, oLs = np_btd.cholesky(diag, sub if T > 1 else None)
    np.testing.assert_allclose(Ld, oLd, rtol=1e-10, atol=1e-12)
    r = rng.normal(size=(T, d))
    if T > 1:
        np.testing.assert_allclose(Ls, oLs, rtol=1e-10, atol=1e-12)
    for tr in (False, True):
        np.testing.assert_allclose(c_ref.btd_solve(Ld, Ls, r, tr), np_btd.solve(oLd, oLs, r, tr), rtol=1e-9, atol=1e-11)
    Sd, Ss = c_ref.btd_inverse_blocks(Ld, Ls)
    oSd, oSs = np_btd.inverse_blocks(oLd, oLs)
    np.testing.assert_allclose(Sd, oSd, rtol=1e-9, atol=1e-11)
    if T > 1:
        np.testing.assert_allclose(Ss, oSs, rtol=1e-9, atol=1e-11)


def test_c_ssm_to_naturals(rng):
    ssm = np_ssm.StateSpaceModel(*random_ssm_params(rng, (), 11, 3))
    got = c_ref.ssm_to_naturals(ssm.A, ssm.concatenated_state_offsets, ssm.concatenated_cholesky_process_covariance)
    for a, b in zip(got, np_transforms.ssm_to_naturals(ssm)):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)


def test_c_cvi_step(rng):
    B, T, d, n = 3, 40, 2, 6
    prm = random_ssm_params(rng, (B,), T, d)
    idx = np.sort(rng.choice(T, size=n, replace=False))
    y = rng.normal(size=(B, n, d))
    cholR = 0.4 * np.eye(d)
    Rinv = np.linalg.inv(cholR @ cholR.T)
    ssms = [np_ssm.StateSpaceModel(*[p[b] for p in prm]) for b in range(B)]
    nats = [np_transforms.ssm_to_naturals(s) for s in ssms]
    st = c_ref.CviStepState(np.stack([n_[0] for n_ in nats]), np.stack([n_[1] for n_ in nats]), np.stack([n_[2] for n_ in nats]),
                            np.stack([s.marginal_means for s in ssms]), np.array([-0.5 * s.log_det_precision() for s in ssms]),
                            idx, y, Rinv, 2 * np.sum(np.log(np.diag(cholR))))
    models = [np_models.CVISitesSSM(ssms[b], np.arange(T) * 0.1, idx, y[b], np_models.MultivariateGaussianLik(cholR)) for b in range(B)]
    for lr_d, lr_g in ((1.0, 1.0), (0.5, 0.3)):
        total = st.step(lr_d, lr_g)
        ref = []
        for m in models:
            m.update_data_sites(lr_d)
            m.update_girsanov_sites(lr_g)
            ref.append(m.classic_elbo())
        np.testing.assert_allclose(st.elbo, ref, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(total, np.sum(ref), rtol=1e-8)


def test_c_sde_kl(rng):
    from oracle import np_sde
    T, d, dt = 9, 2, 0.05
    q = np_ssm.StateSpaceModel(*random_ssm_params(rng, (), T, d, scale_A=0.8))
    mu, cov = q.marginals
    sub = q.subsequent_covariances(cov)
    qd = 0.5 + rng.random(d)
    init_mu, init_cov = 0.1 * rng.normal(size=d), 0.7 * np.eye(d) + 0.1
    kl, grads = c_ref.sde_kl(mu, cov, sub, 1.2, 0.2, qd, dt, init_mu, init_cov)
    okl, ograds = np_sde.sde_ssm_kl_closed_form(mu, cov, sub, 1.2, 0.2, qd, dt, init_mu, init_cov)
    np.testing.assert_allclose(kl, okl, rtol=1e-12)
    for a, b in zip(grads, ograds):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("d,T", [(2, 40), (6, 70)])
def test_c_cvi_dp_step(rng, d, T):
    """The C port of the bench step against the NumPy oracle model; at d = 6 (the bench's state dimension) the oracle runs its
    closed-form route (pinned to the quadrature route at d <= 2 in tests/test_oracle_models.py)."""
    from oracle import np_sde
    B, n, dt = 2, 5, 0.02
    sde = np_sde.DoubleWellSDE(np.eye(d))
    grid = np.arange(T) * dt
    idx = np.sort(rng.choice(np.arange(1, T), size=n, replace=False))
    y = np.sign(rng.normal(size=(B, n, d))) + 0.2 * rng.normal(size=(B, n, d))
    cholR = 0.3 * np.eye(d)
    init = (np.zeros(d), 0.5 * np.eye(d))
    models = [np_models.CVISitesSDE(sde, grid, idx, y[b], np_models.MultivariateGaussianLik(cholR), *init, closed_form=d > 2)
              for b in range(B)]
    nats = [np_transforms.ssm_to_naturals(m.dist_p) for m in models]
    al, be = sde.cubic(dt)
    st = c_ref.CviDpStepState(np.stack([n_[0] for n_ in nats]), np.stack([n_[1] for n_ in nats]), np.stack([n_[2] for n_ in nats]),
                              idx, y, np.linalg.inv(cholR @ cholR.T), 2 * np.sum(np.log(np.diag(cholR))), al, be, np.ones(d), dt, *init)
    for lr_d, lr_g in ((0.5, 0.2), (0.3, 0.1)):
        total = st.step(lr_d, lr_g)
        ref = []
        for m in models:
            m.update_data_sites(lr_d)
            m.update_girsanov_sites(lr_g)
            ref.append(m.classic_elbo())
        np.testing.assert_allclose(st.elbo, ref, rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(total, np.sum(ref), rtol=1e-7)


@pytest.mark.parametrize("d,T,stabilize", [(2, 40, False), (6, 60, True)])
def test_c_vdp_step(rng, d, T, stabilize):
    """The C port of the VDP inference step (ref_vdp_step: update_lagrange + update_param, forward_pass, elbo -- bench.py's config-3
    cpu_baseline) against the NumPy oracle model (oracle/np_models.VariationalMarkovGP, closed-form cubic-drift moments, pinned to the
    reference's quadrature at d <= 2 in tests/test_oracle_sde.py), started at the OU drift -4 x as the bench starts it."""
    from oracle import np_sde
    B, n, dt = 2, 6, 0.01
    sde = np_sde.DoubleWellSDE(np.eye(d))
    grid = np.arange(T) * dt
    idx = np.sort(rng.choice(np.arange(1, T - 1), size=n, replace=False))
    y = np.sign(rng.normal(size=(B, n, d))) + 0.2 * rng.normal(size=(B, n, d))
    cholR = 0.3 * (np.eye(d) + 0.3 * np.eye(d, k=-1))
    init = (np.zeros(d), np.eye(d))
    models = [np_models.VariationalMarkovGP(idx, y[b], sde, grid, np_models.MultivariateGaussianLik(cholR), *init,
                                            stabilize_system=stabilize, closed_form=True) for b in range(B)]
    for m in models:
        m.A = np.broadcast_to(4.0 * np.eye(d), m.A.shape).copy()
    af, bf = np_sde.drift_cubic(sde)
    st = c_ref.VdpStepState(np.stack([m.A for m in models]), np.stack([m.b for m in models]), idx, y, np.linalg.inv(cholR @ cholR.T),
                            2 * np.sum(np.log(np.diag(cholR))), af, bf, np.ones(d), dt, *init, stabilize=stabilize)
    for b, m in enumerate(models):
        mm, SS = m.forward_pass()
        np.testing.assert_allclose(st.m[b], mm, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(st.S[b], SS, rtol=1e-9, atol=1e-12)
    for lr in (0.05, 0.02, 0.02):
        total = st.step(lr)
        ref = []
        for b, m in enumerate(models):
            mm, SS = m.forward_pass()
            m.update_lagrange(mm, SS)
            m.update_param(mm, SS, lr)
            ref.append(m.elbo())
            np.testing.assert_allclose(st.A[b], m.A, rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(st.b[b], m.b, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(st.elbo, ref, rtol=1e-8)
        np.testing.assert_allclose(total, np.sum(ref), rtol=1e-8)


@pytest.mark.parametrize("kname,T", [("m52", 60), ("m32", 41), ("m12", 25)])
def test_c_cvigp_step(rng, kname, T):
    """The C port of the CVI-GP step (ref_cvigp_step: update_sites + elbo on a state-space kernel -- bench.py's config-2 cpu_baseline)
    against the NumPy oracle model (oracle/np_models.CVIGaussianProcess, itself pinned to the reference's KA7 known answers), over
    damped steps: ELBO and both site arrays."""
    from oracle import np_kernels
    k = {"m52": np_kernels.Matern52(0.7, 1.3), "m32": np_kernels.Matern32(1.1, 0.8), "m12": np_kernels.Matern12(2.0, 2.25)}[kname]
    t = np.linspace(0, 6, T) + rng.uniform(0, 0.03, size=T)
    y = np.sin(2 * t)[:, None] + 0.2 * rng.normal(size=(T, 1))
    noise, lr = 0.3, 0.6
    o = np_models.CVIGaussianProcess(t, y, k, np_models.GaussianLik(noise), learning_rate=lr)
    H = k.emission_matrix(t)
    assert H.shape[-2] == 1 and np.allclose(H, H[0])
    st = c_ref.CviGpStepState(k.state_space_model(t), H[0, 0], y, noise, lr)
    for _ in range(4):
        e = st.step()
        o.update_sites()
        np.testing.assert_allclose(e, o.elbo(), rtol=1e-10)
        np.testing.assert_allclose(st.nat1, o.nat1[:, 0], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(st.nat2, o.nat2[:, 0, 0], rtol=1e-10)


@pytest.mark.parametrize("kname,M", [("sum", 12), ("m52", 25), ("m12", 7)])
def test_c_sparse_cvi_step(rng, kname, M):
    """The C port of the sparse / inducing-state CVI step (ref_sparse_cvi_step: update_sites + classic_elbo -- bench.py's config-5
    cpu_baseline) against the NumPy oracle model (oracle/np_conditionals.SparseCVIGaussianProcess, pinned to the reference's KA8 known
    answers), over damped steps, with data before the first and after the last inducing point: ELBO and both site arrays."""
    from oracle import np_conditionals as npc, np_kernels
    k = {"sum": np_kernels.Sum([np_kernels.Matern32(1.1, 0.7), np_kernels.Matern12(0.5, 1.2)]), "m52": np_kernels.Matern52(0.7, 1.3),
         "m12": np_kernels.Matern12(2.0, 2.25)}[kname]
    z = np.linspace(0.0, 5.0, M) + rng.uniform(0, 0.05, size=M)
    t = np.sort(np.concatenate([rng.uniform(-0.4, 5.5, size=3 * M), [-0.3, -0.1, 5.2, 5.45]]))
    y = np.sin(2 * t)[:, None] + 0.2 * rng.normal(size=(t.size, 1))
    noise, lr = 0.3, 0.6
    o = npc.SparseCVIGaussianProcess(k, z, np_models.GaussianLik(noise), learning_rate=lr)
    st = c_ref.SparseCviStepState(k, z, t, y, noise, lr)
    assert st.idx.min() == 0 and st.idx.max() == M
    for _ in range(4):
        e = st.step()
        o.update_sites(t, y)
        np.testing.assert_allclose(e, o.classic_elbo(t, y), rtol=1e-9)
        np.testing.assert_allclose(st.nat1, o.nat1, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(st.nat2, o.nat2, rtol=1e-9, atol=1e-12)
