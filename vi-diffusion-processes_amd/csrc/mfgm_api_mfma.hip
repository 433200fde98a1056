// Dispatch of the MFMA sweeps (8 < d <= 32 as 1 x 1 or 2 x 2 tiles of v_mfma_f64_16x16x4_f64) by tile count; the kernels are
// instantiated in mfgm_api_mfma_t1.hip / mfgm_api_mfma_t2.hip (mfgm_mfma_launch.h).
#include "mfgm_internal.h"
#include "mfgm_wide.h"
#include "mfgm_wband.h"
#include "../../include/mfgm.h"

namespace mfgm {

#define MFGM_DECL(NT_)                                                                                                                          \
    int mfma_launch_##NT_(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st);              \
    int mfma_inv_launch_##NT_(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st);          \
    int mfma_ssm_to_naturals_##NT_(int B, int T, int d, const double* A, const double* off, const double* chol, double cD, double cS,          \
                                   double* lin, double* diag, double* sub, double* part, hipStream_t st);                                   \
    int wband_run_##NT_(const WBandArgs& a, const WScanArgs& s0, hipStream_t st);
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

namespace {
// segments of a recurrence over T positions: passes 1 and 3 walk R positions (2.5 block steps each), pass 2 walks the P segments
void wband_partition(int T, int& R, int& P) {
    R = T;
    if (T > 32) {
        R = 8;
        while ((long)R * R * 5 < (long)T * 2) ++R;
    }
    P = (T + R - 1) / R;
}
}  // namespace

extern "C" size_t mfgm_wband_workspace_doubles(int B, int T, int d) {
    if (B < 1 || T < 1 || d < 1) return 0;
    int R, P;
    wband_partition(T, R, P);
    return ((size_t)7 * B * T + (size_t)6 * B * P) * d * d;
}

extern "C" int mfgm_wband_sigma_dP_sigma(int B, int T, int d, const double* Sig, const double* Sub, const double* dPd, const double* dPs,
                                         double* Xd, double* Xs, double* work, int* info, void* stream) {
    if (B < 1 || T < 2 || d < 1 || d > 32 || !Sig || !Sub || !dPd || !dPs || !Xd || !Xs || !work || !info) return 1;
    using namespace mfgm;
    const size_t n = (size_t)B * T * d * d;
    WBandArgs a;
    a.B = B; a.T = T; a.d = d;
    a.Sig = Sig; a.Sub = Sub; a.dPd = dPd; a.dPs = dPs;
    a.PhiL = work; a.QL = work + n; a.PhiR = work + 2 * n; a.QR = work + 3 * n; a.loc = work + 4 * n; a.Lr = work + 5 * n; a.Rr = work + 6 * n;
    a.Xd = Xd; a.Xs = Xs; a.info = info;
    WScanArgs s;
    s.B = B; s.T = T; s.d = d;
    wband_partition(T, s.R, s.P);
    s.segT = work + 7 * n; s.segQ = nullptr; s.segX = nullptr;      // 2 x (segT, segQ, segX), laid out by the launcher
    s.reverse = 0; s.PhiT = nullptr; s.Q = nullptr; s.X = nullptr;
    hipStream_t st = (hipStream_t)stream;
    return d <= 16 ? wband_run_1(a, s, st) : wband_run_2(a, s, st);
}

