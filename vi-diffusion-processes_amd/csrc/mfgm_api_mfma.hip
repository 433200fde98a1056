// Dispatch of the MFMA sweeps (8 < d <= 32 as 1 x 1 or 2 x 2 tiles of v_mfma_f64_16x16x4_f64) by tile count; the kernels are
// instantiated in mfgm_api_mfma_t1.hip / mfgm_api_mfma_t2.hip (mfgm_mfma_launch.h).
#include "mfgm_internal.h"
#include "mfgm_wide.h"

namespace mfgm {

#define MFGM_DECL(NT_)                                                                                                                          \
    int mfma_launch_##NT_(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st);              \
    int mfma_inv_launch_##NT_(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st);          \
    int mfma_ssm_to_naturals_##NT_(int B, int T, int d, const double* A, const double* off, const double* chol, double cD, double cS,          \
                                   double* lin, double* diag, double* sub, double* part, hipStream_t st);
MFGM_DECL(1)
MFGM_DECL(2)
#undef MFGM_DECL

int mfma_launch(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    return a.d <= 16 ? mfma_launch_1(which, a, has_rhs, has_corr, has_up, want_sub, st) : mfma_launch_2(which, a, has_rhs, has_corr, has_up, want_sub, st);
}

int mfma_inv_launch(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    return a.d <= 16 ? mfma_inv_launch_1(which, a, has_rhs, has_corr, has_up, want_sub, st)
                     : mfma_inv_launch_2(which, a, has_rhs, has_corr, has_up, want_sub, st);
}

int mfma_ssm_to_naturals(int B, int T, int d, const double* A, const double* off, const double* chol, double cD, double cS,
                         double* lin, double* diag, double* sub, double* part, hipStream_t st) {
    return d <= 16 ? mfma_ssm_to_naturals_1(B, T, d, A, off, chol, cD, cS, lin, diag, sub, part, st)
                   : mfma_ssm_to_naturals_2(B, T, d, A, off, chol, cD, cS, lin, diag, sub, part, st);
}

}  // namespace mfgm
