// Launchers of the MFMA sweeps (8 < d <= 32 as 1 x 1 or 2 x 2 tiles of v_mfma_f64_16x16x4_f64; mfgm_mfma.h).
#include "mfgm_internal.h"
#include "mfgm_mfma.h"

namespace mfgm {

namespace {
template <int NT>
int mfma_launch_nt(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
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
}  // namespace

int mfma_ssm_to_naturals(int B, int T, int d, const double* A, const double* off, const double* chol, double cD, double cS,
                         double* lin, double* diag, double* sub, double* part, hipStream_t st) {
    dim3 grid(B * T), block(64);
#define S2N(NT_, LIN_) hipLaunchKernelGGL((km_ssm_to_naturals<NT_, LIN_>), grid, block, 0, st, B, T, d, A, off, chol, cD, cS, lin, diag, sub, part)
    if (d <= 16) { if (lin) S2N(1, true); else S2N(1, false); }
    else { if (lin) S2N(2, true); else S2N(2, false); }
#undef S2N
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfma_launch(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    if (a.d <= 16) return mfma_launch_nt<1>(which, a, has_rhs, has_corr, has_up, want_sub, st);
    return mfma_launch_nt<2>(which, a, has_rhs, has_corr, has_up, want_sub, st);
}

}  // namespace mfgm
