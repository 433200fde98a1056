"""
GPU parity tests of the reference-named host classes (StateSpaceModel, block_tri_diag, the SSM <-> eta/theta
transformations, CVISitesSSM) against the NumPy oracle.  fp64; tolerance 1e-6 relative (north-star: 1e-5).
"""
import numpy as np
import pytest

from oracle import np_btd, np_models, np_ssm, np_transforms
from tests.helpers import assert_close, random_dominant_btd, random_spd_btd, random_ssm_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import torch
    import vidp_amd
    assert torch.cuda.is_available()
    vidp_amd._lib.load()
    return vidp_amd


def dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(x):
    return x.detach().cpu().numpy()


@pytest.mark.parametrize("d,T", [(1, 2), (3, 4), (5, 6), (3, 70)])
def test_state_space_model(amd, rng, batch_shape, d, T):
    from vidp_amd.state_space_model import StateSpaceModel
    prm = random_ssm_params(rng, batch_shape, T, d)
    o = np_ssm.StateSpaceModel(*prm)
    g = StateSpaceModel(*[dev(p) for p in prm])
    od, os_ = o.precision()
    gp = g.precision
    assert_close(host(gp.block_diagonal), od)
    assert_close(host(gp.block_sub_diagonal), os_)
    assert_close(host(g.marginal_means), o.marginal_means)
    assert_close(host(g.marginal_covariances), o.marginal_covariances)
    assert_close(host(g.subsequent_covariances()), o.subsequent_covariances(o.marginal_covariances))
    assert_close(host(g.log_det_precision()), o.log_det_precision())
    prm2 = random_ssm_params(rng, batch_shape, T, d)
    o2 = np_ssm.StateSpaceModel(*prm2)
    g2 = StateSpaceModel(*[dev(p) for p in prm2], plan=g.plan)
    assert_close(host(g.kl_divergence(g2)), o.kl_divergence(o2), rtol=1e-6)
    np.testing.assert_allclose(host(g.kl_divergence(g)), 0.0, atol=1e-6)


def test_zero_transitions(amd):
    import torch
    from vidp_amd.state_space_model import StateSpaceModel
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):
        StateSpaceModel(z(2), z(2, 2), z(0, 2, 2), z(0, 2), z(0, 2, 2))


@pytest.mark.parametrize("d,T", [(1, 1), (1, 4), (3, 1), (3, 4), (2, 4), (3, 5)])
@pytest.mark.parametrize("with_sub", [True, False])
def test_block_tri_diag(amd, rng, batch_shape, d, T, with_sub):
    """Mirrors the reference's tests/unit/test_block_tri_diag.py (KA1)."""
    from vidp_amd.block_tri_diag import LowerTriangularBlockTriDiagonal, SymmetricBlockTriDiagonal
    diag, sub, Ld0, Ls0 = random_spd_btd(rng, batch_shape, T, d, with_sub)
    sym = SymmetricBlockTriDiagonal(dev(diag), None if sub is None else dev(sub))
    dense = np_btd.to_dense(diag, sub)
    assert_close(host(sym.to_dense()), dense)
    chol = sym.cholesky
    Ld, Ls = np_btd.cholesky(diag, sub)
    assert_close(host(chol.block_diagonal), Ld)
    if Ls is not None:
        assert_close(host(chol.block_sub_diagonal), Ls)
    assert_close(host(chol.abs_log_det()), 0.5 * np.linalg.slogdet(dense)[1])
    Sd, _ = np_btd.inverse_blocks(Ld, Ls)
    assert_close(host(chol.block_diagonal_of_inverse()), Sd)
    x = rng.normal(size=batch_shape + (T, d))
    low = LowerTriangularBlockTriDiagonal(dev(Ld0), None if Ls0 is None else dev(Ls0))
    for tr in (False, True):
        assert_close(host(low.solve(dev(x), transpose_left=tr)), np_btd.solve(Ld0, Ls0, x, tr), rtol=1e-5)
        assert_close(host(low.dense_mult(dev(x), transpose_left=tr)), np_btd.dense_mult(Ld0, Ls0, x, False, tr))
    assert_close(host(sym.dense_mult(dev(x))), np_btd.dense_mult(diag, sub, x, True))
    assert_close(host(low.block_diagonal_of_inverse()), np_btd.inverse_blocks(Ld0, Ls0)[0], rtol=1e-5)
    both = sym + sym
    assert_close(host(both.block_diagonal), 2 * diag)


def test_not_positive_definite(amd):
    import torch
    from vidp_amd.block_tri_diag import SymmetricBlockTriDiagonal
    diag = -torch.eye(2, dtype=torch.float64, device="cuda").expand(3, 2, 2).contiguous()
    with pytest.raises(ArithmeticError):
        SymmetricBlockTriDiagonal(diag).cholesky


def test_transformations(amd, rng, batch_shape):
    from vidp_amd import ssm_gaussian_transformations as tr
    from vidp_amd.state_space_model import StateSpaceModel
    prm = random_ssm_params(rng, batch_shape, 9, 3)
    o = np_ssm.StateSpaceModel(*prm)
    g = StateSpaceModel(*[dev(p) for p in prm])
    for a, b in zip(tr.ssm_to_expectations(g), np_transforms.ssm_to_expectations(o)):
        assert_close(host(a), b)
    for a, b in zip(tr.ssm_to_naturals(g), np_transforms.ssm_to_naturals(o)):
        assert_close(host(a), b)
    for a, b in zip(tr.ssm_to_naturals_no_smoothing(g), np_transforms.ssm_to_naturals_no_smoothing(o)):
        assert_close(host(a), b)
    ref = (o.A, o.b, o.cholP0, o.cholQ, o.mu0)
    for fwd, bwd in ((tr.ssm_to_expectations, tr.expectations_to_ssm_params), (tr.ssm_to_naturals, tr.naturals_to_ssm_params),
                     (tr.ssm_to_naturals_no_smoothing, tr.naturals_to_ssm_params_no_smoothing)):
        back = bwd(*fwd(g))
        for a, b in zip(back, ref):
            assert_close(host(a), b)


@pytest.mark.parametrize("d,B,T", [(1, 1, 60), (2, 3, 45), (3, 2, 130)])
def test_cvi_sites_ssm(amd, rng, d, B, T):
    """CVISitesSSM (linear prior): data-site / Girsanov-site updates and classic_elbo against the oracle, per chain."""
    from vidp_amd.likelihoods import MultivariateGaussian
    from vidp_amd.state_space_model import StateSpaceModel
    from vidp_amd.variational_cvi_sde import CVISitesSSM
    prm = random_ssm_params(rng, (B,), T, d)
    grid = np.arange(T) * 0.01
    idx = np.sort(rng.choice(T, size=7, replace=False))
    y = rng.normal(size=(B, 7, d))
    cholR = 0.4 * np.eye(d) + 0.05 * np.tril(rng.normal(size=(d, d)), -1)
    import vidp_amd
    plan = vidp_amd.Plan(B, T, d, R0=8, Rup=4)
    g = CVISitesSSM(StateSpaceModel(*[dev(p) for p in prm], plan=plan), grid, (grid[idx], dev(y)), MultivariateGaussian(dev(cholR)))
    os_ = [np_models.CVISitesSSM(np_ssm.StateSpaceModel(*[p[b] for p in prm]), grid, idx, y[b], np_models.MultivariateGaussianLik(cholR))
           for b in range(B)]
    for it, (lr_d, lr_g) in enumerate(((1.0, 1.0), (0.5, 0.3), (0.7, 0.9))):
        g.update_data_sites(lr_d)
        g.update_girsanov_sites(lr_g)
        e = host(g.classic_elbo_per_trajectory())
        for b, o in enumerate(os_):
            o.update_data_sites(lr_d)
            o.update_girsanov_sites(lr_g)
            np.testing.assert_allclose(e[b], o.classic_elbo(), rtol=1e-6, atol=1e-6)
        mu = host(g.fx_mus)
        cov = host(g.fx_covs)
        for b, o in enumerate(os_):
            assert_close(mu[b], o.fx_mus)
            assert_close(cov[b], o.fx_covs)
    q = g.dist_q
    oq = os_[0].dist_q
    assert_close(host(q.state_transitions)[0], oq.A)
    assert_close(host(q.cholesky_process_covariances)[0], oq.cholQ)
    assert_close(host(q.state_offsets)[0], oq.b)
