// Launchers of the MFMA sweeps for one tile count MFGM_MFMA_NT (1: 8 < d <= 16, 2: 16 < d <= 32), Cholesky form (mfgm_mfma.h) and
// inverse form (mfgm_mfma_inv.h).  Included by mfgm_api_mfma_t1.hip / mfgm_api_mfma_t2.hip: the two tile counts are separate translation
// units because the 1 x 1 kernels are compiled with the MFMA accumulators in VGPRs (no v_accvgpr moves), which the 2 x 2 kernels, whose
// tiles need the AGPR half of the register file, cannot be.
#include "mfgm_internal.h"
#include "mfgm_mfma_inv.h"
#include "mfgm_wband.h"

#define MFGM_CAT_(a, b) a##b
#define MFGM_CAT(a, b) MFGM_CAT_(a, b)

namespace mfgm {

int MFGM_CAT(mfma_launch_, MFGM_MFMA_NT)(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    constexpr int NT = MFGM_MFMA_NT;
    dim3 grid((a.lv.L / a.lv.P) * a.nseg), block(64);   // chains x covered segments
#define KM(K) hipLaunchKernelGGL((K), grid, block, 0, st, a)
    if (which == 0) {
        if (has_rhs) { if (has_corr) KM((km_reduce<NT, true, true>)); else KM((km_reduce<NT, true, false>)); }
        else { if (has_corr) KM((km_reduce<NT, false, true>)); else KM((km_reduce<NT, false, false>)); }
    } else if (which == 1) {
        if (has_rhs) {
            if (has_corr) { if (has_up) KM((km_forward<NT, true, true, true>)); else KM((km_forward<NT, true, true, false>)); }
            else { if (has_up) KM((km_forward<NT, true, false, true>)); else KM((km_forward<NT, true, false, false>)); }
        } else {
            if (has_corr) { if (has_up) KM((km_forward<NT, false, true, true>)); else KM((km_forward<NT, false, true, false>)); }
            else { if (has_up) KM((km_forward<NT, false, false, true>)); else KM((km_forward<NT, false, false, false>)); }
        }
    } else {
        if (has_rhs) {
            if (has_up) { if (want_sub) KM((km_backward<NT, true, true, true>)); else KM((km_backward<NT, true, true, false>)); }
            else { if (want_sub) KM((km_backward<NT, true, false, true>)); else KM((km_backward<NT, true, false, false>)); }
        } else {
            if (has_up) { if (want_sub) KM((km_backward<NT, false, true, true>)); else KM((km_backward<NT, false, true, false>)); }
            else { if (want_sub) KM((km_backward<NT, false, false, true>)); else KM((km_backward<NT, false, false, false>)); }
        }
    }
#undef KM
    MFGM_CHECK_LAUNCH();
    return 0;
}

int MFGM_CAT(mfma_inv_launch_, MFGM_MFMA_NT)(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    constexpr int NT = MFGM_MFMA_NT;
    dim3 grid((a.lv.L / a.lv.P) * a.nseg), block(64);   // chains x covered segments
#define KM(K) hipLaunchKernelGGL((K), grid, block, 0, st, a)
    if (a.site2 && which < 2) {
        // sparse-CVI inputs: level 0 of one chain, with a right-hand side
        if (has_corr || !has_rhs || a.lv.L != a.lv.P) return 1;
        if (which == 0) KM((kmi_reduce<NT, true, false, true>));
        else if (has_up) KM((kmi_forward<NT, true, false, true, true>));
        else KM((kmi_forward<NT, true, false, false, true>));
        MFGM_CHECK_LAUNCH();
        return 0;
    }
    if (which == 0) {
        if (has_rhs) { if (has_corr) KM((kmi_reduce<NT, true, true>)); else KM((kmi_reduce<NT, true, false>)); }
        else { if (has_corr) KM((kmi_reduce<NT, false, true>)); else KM((kmi_reduce<NT, false, false>)); }
    } else if (which == 1) {
        if (has_rhs) {
            if (has_corr) { if (has_up) KM((kmi_forward<NT, true, true, true>)); else KM((kmi_forward<NT, true, true, false>)); }
            else { if (has_up) KM((kmi_forward<NT, true, false, true>)); else KM((kmi_forward<NT, true, false, false>)); }
        } else {
            if (has_corr) { if (has_up) KM((kmi_forward<NT, false, true, true>)); else KM((kmi_forward<NT, false, true, false>)); }
            else { if (has_up) KM((kmi_forward<NT, false, false, true>)); else KM((kmi_forward<NT, false, false, false>)); }
        }
    } else {
        if (has_rhs) {
            if (has_up) { if (want_sub) KM((kmi_backward<NT, true, true, true>)); else KM((kmi_backward<NT, true, true, false>)); }
            else { if (want_sub) KM((kmi_backward<NT, true, false, true>)); else KM((kmi_backward<NT, true, false, false>)); }
        } else {
            if (has_up) { if (want_sub) KM((kmi_backward<NT, false, true, true>)); else KM((kmi_backward<NT, false, true, false>)); }
            else { if (want_sub) KM((kmi_backward<NT, false, false, true>)); else KM((kmi_backward<NT, false, false, false>)); }
        }
    }
#undef KM
    MFGM_CHECK_LAUNCH();
    return 0;
}

int MFGM_CAT(mfma_ssm_to_naturals_, MFGM_MFMA_NT)(int B, int T, int d, const double* A, const double* off, const double* chol, double cD, double cS,
                                                   double* lin, double* diag, double* sub, double* part, hipStream_t st) {
    dim3 grid(B * T), block(64);
    if (lin) hipLaunchKernelGGL((km_ssm_to_naturals<MFGM_MFMA_NT, true>), grid, block, 0, st, B, T, d, A, off, chol, cD, cS, lin, diag, sub, part);
    else hipLaunchKernelGGL((km_ssm_to_naturals<MFGM_MFMA_NT, false>), grid, block, 0, st, B, T, d, A, off, chol, cD, cS, lin, diag, sub, part);
    MFGM_CHECK_LAUNCH();
    return 0;
}

// band of Sigma dP Sigma (mfgm_wband.h): node pass, the two recurrences (three passes each when a chain has more than one segment), node pass
int MFGM_CAT(wband_run_, MFGM_MFMA_NT)(const WBandArgs& a, const WScanArgs& s0, hipStream_t st) {
    constexpr int NT = MFGM_MFMA_NT;
    const dim3 nodes(a.B * a.T), block(64);
    hipLaunchKernelGGL((kwb_prepare<NT>), nodes, block, 0, st, a);
    MFGM_CHECK_LAUNCH();
    WScanPair w;
    for (int rev = 0; rev < 2; ++rev) {
        w.s[rev] = s0;
        w.s[rev].reverse = rev;
        w.s[rev].PhiT = rev ? a.PhiR : a.PhiL;
        w.s[rev].Q = rev ? a.QR : a.QL;
        w.s[rev].X = rev ? a.Rr : a.Lr;
        // each recurrence has its own third of the segment arrays
        const size_t m = (size_t)a.B * s0.P * a.d * a.d;
        w.s[rev].segT = s0.segT + (size_t)rev * 3 * m;
        w.s[rev].segQ = w.s[rev].segT + m;
        w.s[rev].segX = w.s[rev].segQ + m;
    }
    if (s0.P > 1) {
        hipLaunchKernelGGL((kwb_scan_maps<NT>), dim3(a.B * s0.P, 2), block, 0, st, w);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL((kwb_scan_tops<NT>), dim3(a.B, 2), block, 0, st, w);
        MFGM_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL((kwb_scan_sweep<NT>), dim3(a.B * s0.P, 2), block, 0, st, w);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL((kwb_finish<NT>), nodes, block, 0, st, a);
    MFGM_CHECK_LAUNCH();
    return 0;
}

}  // namespace mfgm
