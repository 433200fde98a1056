// Local (per time step) kernels on packed arrays: SSM parameters -> naturals / precision, KL terms, stationary kernels.
#include "mfgm_internal.h"
#include "mfgm_sweeps.h"
#include "mfgm_local.h"
#include "mfgm_nat2ssm.h"
#include "mfgm_bidiag.h"

using namespace mfgm;

namespace {
template <int D>
int s2n_impl(const Plan& P, const double* A, const double* off, const double* chol, double cD, double cS, double* lin,
             double* diag, double* sub, double* sumlogchol, double* ws, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    double* part = sumlogchol ? ws + P.off_part[0] : nullptr;
    if (lin) hipLaunchKernelGGL((k_ssm_to_naturals<D, true>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, A, off, chol, cD, cS, lin, diag, sub, part);
    else hipLaunchKernelGGL((k_ssm_to_naturals<D, false>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, A, off, chol, cD, cS, lin, diag, sub, part);
    MFGM_CHECK_LAUNCH();
    if (sumlogchol) {
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, part, lv.P, 0, sumlogchol, (double*)nullptr);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}
template <int D>
int kl_impl(const Plan& P, const double* Sig, const double* Sub, const double* mu, const double* Pd, const double* Ps, double aD,
            double aS, const double* mup, double* trace, double* maha, double* ws, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    double* part = ws + P.off_part[0];
    hipLaunchKernelGGL((k_kl_terms<D>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, Sig, Sub, mu, Pd, Ps, aD, aS, mup, part);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, part, lv.P, lv.Lpad, trace, maha);
    MFGM_CHECK_LAUNCH();
    return 0;
}
template <int D>
int n2s_impl(const Plan& P, const double* Sig, const double* Sub, const double* mu, const double* td, const double* ts, double* A,
             double* off, double* chol, int* info, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    hipLaunchKernelGGL((k_naturals_to_ssm<D>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, Sig, Sub, mu, td, ts, A, off, chol, info);
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

int mfgm_packed_naturals_to_ssm(const mfgm_plan* plan, const double* Sig, const double* Sub, const double* mu, const double* theta_diag,
                                const double* theta_sub, double* A, double* off, double* chol, int* info, void* stream) {
    if (!plan || !Sig || !Sub || !mu || !theta_diag || !theta_sub || !A || !off || !chol || !info) return 1;
    const Plan& P = plan->p;
    if (P.wide) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (n2s_impl<DD>(P, Sig, Sub, mu, theta_diag, theta_sub, A, off, chol, info, st)));
}

size_t mfgm_bidiag_scratch_doubles(int B, int T, int d) {
    if (B < 1 || T < 1 || d < 1) return 0;
    const size_t P = (size_t)(T + kBidiagR - 1) / kBidiagR;
    return (size_t)B * P * ((size_t)d * d + 2 * d);
}

int mfgm_bidiag_solve(int B, int T, int d, const double* Ld, const double* Ls, const double* r, double* x, int transpose, double* scratch,
                      void* stream) {
    if (B < 1 || T < 1 || d < 1 || d > 32 || !Ld || !r || !x || !scratch || (T > 1 && !Ls)) return 1;
    const int P = (T + kBidiagR - 1) / kBidiagR;
    double* Phi = scratch;
    double* cvec = Phi + (size_t)B * P * d * d;
    double* xin = cvec + (size_t)B * P * d;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B * P), block(64);
#define BIDIAG(DM_)                                                                                                                    \
    if (transpose) {                                                                                                                   \
        hipLaunchKernelGGL((k_bidiag<DM_, true, 0>), grid, block, 0, st, T, d, P, Ld, Ls, r, Phi, cvec, (const double*)nullptr, x);      \
        hipLaunchKernelGGL((k_bidiag_scan<true>), dim3(B), block, 0, st, d, P, (const double*)Phi, (const double*)cvec, xin);            \
        hipLaunchKernelGGL((k_bidiag<DM_, true, 1>), grid, block, 0, st, T, d, P, Ld, Ls, r, Phi, cvec, (const double*)xin, x);          \
    } else {                                                                                                                           \
        hipLaunchKernelGGL((k_bidiag<DM_, false, 0>), grid, block, 0, st, T, d, P, Ld, Ls, r, Phi, cvec, (const double*)nullptr, x);     \
        hipLaunchKernelGGL((k_bidiag_scan<false>), dim3(B), block, 0, st, d, P, (const double*)Phi, (const double*)cvec, xin);           \
        hipLaunchKernelGGL((k_bidiag<DM_, false, 1>), grid, block, 0, st, T, d, P, Ld, Ls, r, Phi, cvec, (const double*)xin, x);         \
    }
    if (d <= 8) { BIDIAG(8) } else if (d <= 16) { BIDIAG(16) } else { BIDIAG(32) }
#undef BIDIAG
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_btd_matvec(int B, int T, int d, const double* diag, const double* sub, const double* x, double* out, int symmetric,
                    int transpose, void* stream) {
    if (B < 1 || T < 1 || d < 1 || !diag || !x || !out || x == out) return 1;
    const size_t total = (size_t)B * T * d;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(k_btd_matvec, dim3(blocks), dim3(256), 0, (hipStream_t)stream, B, T, d, diag, sub, x, out, symmetric, transpose);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_packed_ssm_to_naturals(const mfgm_plan* plan, const double* A, const double* off, const double* chol, double cD,
                                double cS, double* lin, double* diag, double* sub, double* sumlogchol, void* ws,
                                void* stream) {
    if (!plan || !chol || !diag || !sub || !ws) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !A) return 1;
    if ((lin != nullptr) != (off != nullptr)) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (P.wide) return wide_ssm_to_naturals(P, A, off, chol, cD, cS, lin, diag, sub, sumlogchol, (double*)ws, st);
    MFGM_DISPATCH_D(P.d, (s2n_impl<DD>(P, A, off, chol, cD, cS, lin, diag, sub, sumlogchol, (double*)ws, st)));
}

int mfgm_packed_kl_terms(const mfgm_plan* plan, const double* Sig, const double* Sub, const double* mu, const double* Pd,
                         const double* Ps, double aD, double aS, const double* mup, double* trace, double* maha, void* ws,
                         void* stream) {
    if (!plan || !Sig || !Sub || !mu || !Pd || !Ps || !mup || !trace || !maha || !ws) return 1;
    const Plan& P = plan->p;
    hipStream_t st = (hipStream_t)stream;
    if (P.wide) return wide_kl_terms(P, Sig, Sub, mu, Pd, Ps, aD, aS, mup, trace, maha, (double*)ws, st);
    MFGM_DISPATCH_D(P.d, (kl_impl<DD>(P, Sig, Sub, mu, Pd, Ps, aD, aS, mup, trace, maha, (double*)ws, st)));
}

}  // extern "C"

static_assert(sizeof(mfgm_kernel_spec) == sizeof(mfgm::KernelSpec), "public and internal kernel spec structs must match");

namespace {
template <int D>
int stationary_impl(const Plan& P, const KernelSpec& ks, const double* dts, double* A, double* off, double* chol, int* info,
                    hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    hipLaunchKernelGGL((k_stationary_ssm<D>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, ks, dts, A, off, chol, info);
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int mfgm_packed_stationary_ssm(const mfgm_plan* plan, const mfgm_kernel_spec* spec, const double* time_deltas,
                                          double* A, double* off, double* chol, int* info, void* stream) {
    if (!plan || !spec || !A || !off || !chol || !info) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !time_deltas) return 1;
    KernelSpec ks;
    memcpy(&ks, spec, sizeof(ks));
    if (ks.ncomp < 1 || ks.ncomp > 8) return 1;
    int dim = 0;
    for (int c = 0; c < ks.ncomp; ++c) {
        if (ks.order[c] < 1 || ks.order[c] > 3 || ks.offset[c] != dim) return 1;
        dim += ks.order[c];
    }
    if (dim != P.d) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (stationary_impl<DD>(P, ks, time_deltas, A, off, chol, info, st)));
}

