// Launchers of the MFMA sweeps (8 < d <= 16, v_mfma_f64_16x16x4_f64 tiles; mfgm_mfma.h).
#include "mfgm_internal.h"
#include "mfgm_mfma.h"

namespace mfgm {

int mfma_launch(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    dim3 grid((a.lv.L / a.lv.P) * a.nseg), block(64);   // chains x covered segments
#define KM(K) hipLaunchKernelGGL((K), grid, block, 0, st, a)
    if (which == 0) {
        if (has_rhs) { if (has_corr) KM((km_reduce<true, true>)); else KM((km_reduce<true, false>)); }
        else { if (has_corr) KM((km_reduce<false, true>)); else KM((km_reduce<false, false>)); }
    } else if (which == 1) {
        if (has_rhs) {
            if (has_corr) { if (has_up) KM((km_forward<true, true, true>)); else KM((km_forward<true, true, false>)); }
            else { if (has_up) KM((km_forward<true, false, true>)); else KM((km_forward<true, false, false>)); }
        } else {
            if (has_corr) { if (has_up) KM((km_forward<false, true, true>)); else KM((km_forward<false, true, false>)); }
            else { if (has_up) KM((km_forward<false, false, true>)); else KM((km_forward<false, false, false>)); }
        }
    } else {
        if (has_rhs) {
            if (has_up) { if (want_sub) KM((km_backward<true, true, true>)); else KM((km_backward<true, true, false>)); }
            else { if (want_sub) KM((km_backward<true, false, true>)); else KM((km_backward<true, false, false>)); }
        } else {
            if (has_up) { if (want_sub) KM((km_backward<false, true, true>)); else KM((km_backward<false, true, false>)); }
            else { if (want_sub) KM((km_backward<false, false, true>)); else KM((km_backward<false, false, false>)); }
        }
    }
#undef KM
    MFGM_CHECK_LAUNCH();
    return 0;
}

}  // namespace mfgm
