"""
ctypes binding of libmfgm.so (C ABI declared in include/mfgm.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``.  There is no CPU
fallback: when the shared object is missing every call raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFGM_LIB selects another build of the same library (A/B timing of kernel variants); the default is the in-tree one
LIB_PATH = os.environ.get("MFGM_LIB") or os.path.join(_HERE, "csrc", "libmfgm.so")

VEC, FULL, SYM, TRI = 0, 1, 2, 3

_lib = None

EXPORTS = {
    # name: (restype, argtypes)
    "mfgm_version": (ctypes.c_char_p, []),
    "mfgm_plan_create": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_void_p)]),
    "mfgm_plan_destroy": (None, [ctypes.c_void_p]),
    "mfgm_plan_describe": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "mfgm_plan_workspace_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "mfgm_packed_doubles": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_int]),
    "mfgm_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                 ctypes.c_void_p]),
    "mfgm_unpack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_void_p]),
    "mfgm_packed_factor": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_double] * 3 + [ctypes.c_void_p] * 8),
    "mfgm_packed_selinv": (ctypes.c_int, [ctypes.c_void_p] * 9),
    "mfgm_packed_factor_form": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_double] * 3
                                + [ctypes.c_void_p] * 8),
    "mfgm_packed_selinv_form": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8),
    "mfgm_combine_terms": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_double,
                                          ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "mfgm_lincomb": (ctypes.c_int, [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_double,
                                    ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    "mfgm_congruence_scan_workspace_doubles": (ctypes.c_size_t, [ctypes.c_void_p]),
    "mfgm_congruence_scan": (ctypes.c_int, [ctypes.c_void_p] * 6),
    "mfgm_band_workspace_doubles": (ctypes.c_size_t, [ctypes.c_void_p]),
    "mfgm_band_sigma_dP_sigma": (ctypes.c_int, [ctypes.c_void_p] * 9),
    "mfgm_wband_workspace_doubles": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "mfgm_wband_sigma_dP_sigma": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.c_void_p] * 9),
    "mfgm_site_lerp": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_double, ctypes.c_void_p]),
    "mfgm_site_lerp_to": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_double,
                                         ctypes.c_void_p]),
    "mfgm_cq_elbo": (ctypes.c_int, [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_double] + [ctypes.c_void_p] * 4),
    "mfgm_node_io": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]),
    "mfgm_node_io_pair": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]),
    "mfgm_packed_ssm_to_naturals": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_double] * 2 + [ctypes.c_void_p] * 6),
    "mfgm_packed_factor_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
                                 + [ctypes.c_double] * 3 + [ctypes.c_void_p] * 6),
    "mfgm_packed_selinv_level": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8),
    "mfgm_packed_sde_kl": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 14),
    "mfgm_packed_linearize_cubic": (ctypes.c_int, [ctypes.c_void_p] * 8),
    "mfgm_packed_stationary_ssm": (ctypes.c_int, [ctypes.c_void_p] * 8),
    "mfgm_vdp_workspace_doubles": (ctypes.c_size_t, [ctypes.c_void_p]),
    "mfgm_packed_vdp_to_ssm": (ctypes.c_int, [ctypes.c_void_p] * 8),
    "mfgm_packed_vdp_to_naturals": (ctypes.c_int, [ctypes.c_void_p] * 10),
    "mfgm_packed_vdp_marginals": (ctypes.c_int, [ctypes.c_void_p] * 12),
    "mfgm_packed_vdp_marginals_products": (ctypes.c_int, [ctypes.c_void_p] * 17),
    "mfgm_packed_vdp_esde": (ctypes.c_int, [ctypes.c_void_p] * 11),
    "mfgm_packed_vdp_lagrange": (ctypes.c_int, [ctypes.c_void_p] * 14),
    "mfgm_packed_vdp_update_param": (ctypes.c_int, [ctypes.c_void_p] * 9),
    "mfgm_packed_vdp_lagrange_update": (ctypes.c_int, [ctypes.c_void_p] * 14),
    "mfgm_packed_vdp_lagrange_update_final": (ctypes.c_int, [ctypes.c_void_p] * 14),
    "mfgm_packed_vdp_lagrange_update0": (ctypes.c_int, [ctypes.c_void_p] * 13 + [ctypes.c_int, ctypes.c_void_p]),
    "mfgm_packed_selinv_mom": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 9),
    "mfgm_packed_selinv_mom_s": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
                                 + [ctypes.c_void_p] * 6),
    "mfgm_packed_sde_lean": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 9),
    "mfgm_site_update_pair": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_double, ctypes.c_void_p]),
    "mfgm_mvn_obs_ve": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
                        + [ctypes.c_void_p] * 4),
    "mfgm_packed_selinv_kl": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
                              + [ctypes.c_void_p] * 7),
    "mfgm_packed_selinv_girsanov": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
                                    + [ctypes.c_void_p] * 9),
    "mfgm_cq_dyn_doubles": (ctypes.c_size_t, [ctypes.c_void_p]),
    "mfgm_cq_slot_ints": (ctypes.c_size_t, [ctypes.c_void_p]),
    "mfgm_cq_pack": (ctypes.c_int, [ctypes.c_void_p] * 7),
    "mfgm_cq_unpack": (ctypes.c_int, [ctypes.c_void_p] * 6),
    "mfgm_cq_slots": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "mfgm_cq_factor": (ctypes.c_int, [ctypes.c_void_p] * 9),
    "mfgm_cq_factor_pipelined": (ctypes.c_int, [ctypes.c_void_p] * 8 + [ctypes.c_int] + [ctypes.c_void_p] * 3),
    "mfgm_quad_linearize": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6),
    "mfgm_quad_kl_scratch_doubles": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "mfgm_quad_kl": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 11),
    "mfgm_quad_vdp_lagrange": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.c_double] * 2 + [ctypes.c_void_p] * 8),
    "mfgm_quad_esde": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 12),
    "mfgm_cq_factor_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6),
    "mfgm_cq_selinv_girsanov": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 7),
    "mfgm_cq_selinv_kl": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 11),
    "mfgm_mvn_ve_compact": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    "mfgm_kf_sites_loglik": (ctypes.c_int, [ctypes.c_void_p] * 15),
    "mfgm_kf_sites_predict": (ctypes.c_int, [ctypes.c_void_p] * 16),
    "mfgm_kf_sites_elbo": (ctypes.c_int, [ctypes.c_void_p] * 9 + [ctypes.c_double] + [ctypes.c_void_p] * 6),
    "mfgm_kf_sites_predict_factored": (ctypes.c_int, [ctypes.c_void_p] * 11),
    "mfgm_sparse_theta": (ctypes.c_int, [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 9),
    "mfgm_sparse_predict": (ctypes.c_int, [ctypes.c_void_p] * 7),
    "mfgm_cond_predict": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.c_void_p] * 11),
    "mfgm_sparse_site_update": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_double] + [ctypes.c_void_p] * 3),
    "mfgm_packed_naturals_to_ssm": (ctypes.c_int, [ctypes.c_void_p] * 11),
    "mfgm_bidiag_scratch_doubles": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "mfgm_bidiag_solve": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "mfgm_btd_matvec": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "mfgm_natural_workspace_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "mfgm_btd_cholesky": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_double] * 2 + [ctypes.c_void_p] * 6),
    "mfgm_btd_posterior": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_double] * 3 + [ctypes.c_void_p] * 7),
    "mfgm_packed_kl_terms": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_double] * 2 + [ctypes.c_void_p] * 5),
    "mfgm_unpack_moments": (ctypes.c_int, [ctypes.c_void_p] * 4),
    "mfgm_plan_set_shard": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "mfgm_plan_set_shard_level": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mfgm_plan_decode_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "mfgm_plan_check_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "mfgm_plan_level": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "mfgm_plan_exchange_region": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "mfgm_packed_factor_phase": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_double] * 3
                                 + [ctypes.c_void_p] * 8),
    "mfgm_packed_factor_phase_form": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
                                      + [ctypes.c_double] * 3 + [ctypes.c_void_p] * 8),
    "mfgm_wide_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_double] * 3
                        + [ctypes.c_void_p] * 11),
    "mfgm_sparse_predict_kl": (ctypes.c_int, [ctypes.c_void_p] * 9 + [ctypes.c_double] * 2 + [ctypes.c_void_p] * 5),
    "mfgm_sparse_factor": (ctypes.c_int, [ctypes.c_void_p] * 14),
    "mfgm_sparse_factor_q": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 13),
    "mfgm_sparse_site_update_q": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_double] + [ctypes.c_void_p] * 3),
    "mfgm_wide_stage_q": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_double] * 3 + [ctypes.c_void_p] * 8),
    "mfgm_sparse_factor_phase": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 13),
    "mfgm_plan_shard_left_marginal": (ctypes.c_int, [ctypes.c_void_p] * 5),
    "mfgm_batched_cholesky": (ctypes.c_int, [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4),
    "mfgm_batched_trsm": (ctypes.c_int, [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]),
}


class SdeParams(ctypes.Structure):
    """mfgm_sde_params (include/mfgm.h)."""
    _fields_ = [("alpha", ctypes.c_double * 8), ("beta", ctypes.c_double * 8), ("W", ctypes.c_double * 8),
                ("P0inv", ctypes.c_double * 36), ("mu0", ctypes.c_double * 8), ("logdetQp", ctypes.c_double),
                ("logdetP0", ctypes.c_double), ("lr", ctypes.c_double), ("clip_lo", ctypes.c_double),
                ("clip_hi", ctypes.c_double), ("sq_dtq", ctypes.c_double * 8), ("cholP0", ctypes.c_double * 36),
                ("theta", ctypes.c_double * 8), ("dt", ctypes.c_double), ("kind", ctypes.c_int), ("pad_", ctypes.c_int)]


class CqState(ctypes.Structure):
    """mfgm_cq_state (include/mfgm.h)."""
    _fields_ = [("dyn", ctypes.c_void_p), ("d_off", ctypes.c_double), ("s_off", ctypes.c_double), ("p0_off", ctypes.c_void_p),
                ("slot", ctypes.c_void_p), ("site_lin", ctypes.c_void_p), ("site_sym", ctypes.c_void_p)]


QUAD_NTHETA = 40


class QuadDrift(ctypes.Structure):
    """mfgm_quad_drift (include/mfgm.h)."""
    _fields_ = [("kind", ctypes.c_int), ("d", ctypes.c_int), ("nh", ctypes.c_int), ("pad_", ctypes.c_int),
                ("theta", ctypes.c_double * QUAD_NTHETA), ("dt", ctypes.c_double), ("W", ctypes.c_double * 6),
                ("logdetQp", ctypes.c_double), ("mu0", ctypes.c_double * 3), ("P0inv", ctypes.c_double * 6),
                ("logdetP0", ctypes.c_double), ("clip_lo", ctypes.c_double), ("clip_hi", ctypes.c_double)]


class SparseData(ctypes.Structure):
    """mfgm_sparse_data (include/mfgm.h)."""
    _fields_ = [("M", ctypes.c_int), ("d", ctypes.c_int), ("N", ctypes.c_int), ("seg", ctypes.c_void_p), ("w", ctypes.c_void_p),
                ("c", ctypes.c_void_p), ("prior_mean", ctypes.c_void_p), ("prior_cov", ctypes.c_void_p), ("m_lo", ctypes.c_int),
                ("m_hi", ctypes.c_int)]


class KfSites(ctypes.Structure):
    """mfgm_kf_sites (include/mfgm.h)."""
    _fields_ = [("H", ctypes.c_double * 32), ("o", ctypes.c_int), ("site_batch", ctypes.c_int), ("nat1", ctypes.c_void_p),
                ("nat2", ctypes.c_void_p), ("Hmu", ctypes.c_void_p)]


class KernelSpec(ctypes.Structure):
    """mfgm_kernel_spec (include/mfgm.h)."""
    _fields_ = [("ncomp", ctypes.c_int), ("order", ctypes.c_int * 8), ("offset", ctypes.c_int * 8), ("lam", ctypes.c_double * 8),
                ("var", ctypes.c_double * 8), ("mean", ctypes.c_double * 8), ("jitter", ctypes.c_double)]


class VdpParams(ctypes.Structure):
    """mfgm_vdp_params (include/mfgm.h)."""
    _fields_ = [("af", ctypes.c_double * 8), ("bf", ctypes.c_double * 8), ("q", ctypes.c_double * 8), ("mu0", ctypes.c_double * 8),
                ("chol0", ctypes.c_double * 36), ("dt", ctypes.c_double), ("lr", ctypes.c_double), ("clip", ctypes.c_double)]


class MfgmError(RuntimeError):
    pass


def load():
    """Load libmfgm.so (once) and declare every exported signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MfgmError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    msg = {1: "bad argument / unsupported shape", 3: "HIP runtime error"}.get(rc, f"error {rc}")
    if rc == 1:
        raise ValueError(f"{what}: {msg}")
    raise MfgmError(f"{what}: {msg}")
