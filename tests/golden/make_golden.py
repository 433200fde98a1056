"""
Generate the golden vectors under tests/golden/ from the reference's own TF-free test tools.

Run ONLY in the build container (needs /root/reference):  python tests/golden/make_golden.py
The reference modules are loaded by file path (they import nothing but numpy / scipy):
    tests/tools/numpy_kalman_filter.py   (filter + RTS smoother + per-step log-lik)
    tests/tools/generate_random_objects.py
    tests/tools/kernels/kernels.py       (Matern via scipy.linalg.expm)
Fixture recipes follow tests/integration/test_kalman_filter.py:30-102 (KA2),
tests/integration/test_kalman_filter_with_sites.py:41-115 (KA3) and tests/unit/test_matern.py (KA6);
seed 71892305 as tests/conftest.py:22.  Only inputs and expected outputs are stored (data, no code).
"""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/tests/tools"
HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 71892305


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def kalman_fixture(nkf_mod, gro, batch_shape, rng_seed):
    np.random.seed(rng_seed)
    num_transitions, d, o = 7, 3, 2
    A = np.random.normal(size=(d, d))
    cholQ = gro.generate_random_lower_triangular_matrix(d)
    H = np.random.normal(size=(o, d))
    R = gro.generate_random_pos_def_matrix(o)
    mu0 = np.random.normal(size=d)
    b = np.random.normal(size=d)
    cholP0 = gro.generate_random_lower_triangular_matrix(d)
    kf = nkf_mod.NumpyKalmanFilter(
        num_timesteps=num_transitions + 1, transition_matrix=A, transition_mean=b,
        transition_noise=cholQ @ cholQ.T, observation_matrix=H, observation_noise=R,
        initial_state_prior_mean=mu0, initial_state_prior_cov=cholP0 @ cholP0.T)
    y = kf.generate_trajectories(batch_shape)
    log_liks, f_mu, f_cov, p_mu, p_cov = kf.forward_filter(y)
    s_mu, s_cov = kf.backward_smoothing_pass(f_mu, f_cov, p_mu, p_cov)
    return dict(A=A, cholQ=cholQ, H=H, R=R, mu0=mu0, b=b, cholP0=cholP0, y=y,
                log_lik_total=np.sum(log_liks), log_liks=log_liks,
                smooth_means=s_mu, smooth_covs=s_cov, filter_means=f_mu, filter_covs=f_cov)


def sites_fixture(nkf_mod, gro, rng_seed):
    np.random.seed(rng_seed)
    num_transitions, d, o = 6, 2, 1
    A = np.random.normal(size=(d, d))
    cholQ = gro.generate_random_lower_triangular_matrix(d)
    means = np.random.normal(size=(num_transitions + 1, o))
    H = np.random.normal(size=(o, d))
    covs = gro.generate_random_pos_def_matrix(o, (num_transitions + 1,))
    mu0 = np.random.normal(size=d) * 0.0
    b = np.random.normal(size=d) * 0.0
    cholP0 = gro.generate_random_lower_triangular_matrix(d)
    kf = nkf_mod.NumpyKalmanFilterWithSites(
        num_timesteps=num_transitions + 1, transition_matrix=A, transition_mean=b,
        transition_noise=cholQ @ cholQ.T, observation_matrix=H, observation_covariances=covs,
        observation_means=means, initial_state_prior_mean=mu0, initial_state_prior_cov=cholP0 @ cholP0.T)
    log_liks, f_mu, f_cov, p_mu, p_cov = kf.forward_filter(means)
    s_mu, s_cov = kf.backward_smoothing_pass(f_mu, f_cov, p_mu, p_cov)
    return dict(A=A, cholQ=cholQ, H=H, site_means=means, site_covs=covs, mu0=mu0, b=b, cholP0=cholP0,
                log_lik_total=np.sum(log_liks), smooth_means=s_mu, smooth_covs=s_cov)


def matern_fixture(kern_mod):
    rng = np.random.default_rng(SEED)
    out = {}
    dts = rng.exponential(scale=0.3, size=(2, 9))
    out["time_deltas"] = dts
    shape = kern_mod.DataShape(batch_shape=(2,), time_dim=10)
    for name, cls in (("m12", kern_mod.Matern12Test), ("m32", kern_mod.Matern32Test),
                      ("m52", kern_mod.Matern52Test)):
        for i, (var, ls) in enumerate(((1.3, 0.7), (0.4, 2.1))):
            k = cls(variance=var, length_scale=ls, data_shape=shape)
            out[f"{name}_{i}_params"] = np.array([var, ls])
            out[f"{name}_{i}_A"] = k.state_transitions(None, dts)
            out[f"{name}_{i}_Q"] = k.process_covariances(None, dts)
            out[f"{name}_{i}_Pinf"] = k.steady_state_covariance()
    return out


def main():
    gro = _load("ref_generate_random_objects", os.path.join(REF, "generate_random_objects.py"))
    nkf = _load("ref_numpy_kalman_filter", os.path.join(REF, "numpy_kalman_filter.py"))
    kern = _load("ref_kernels", os.path.join(REF, "kernels", "kernels.py"))

    for tag, bs in (("b0", ()), ("b3", (3,)), ("b21", (2, 1))):
        np.savez(os.path.join(HERE, f"kalman_filter_{tag}.npz"), **kalman_fixture(nkf, gro, bs, SEED))
    np.savez(os.path.join(HERE, "kalman_filter_sites.npz"), **sites_fixture(nkf, gro, SEED))
    np.savez(os.path.join(HERE, "matern_expm.npz"), **matern_fixture(kern))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
