// Tensor-product Gauss-Hermite kernels (mfgm_quad.h): coupled / non-polynomial drifts and full diffusion matrices, natural layout.
#include "mfgm_internal.h"
#include "mfgm_quad.h"

using namespace mfgm;

namespace {
bool quad_ok(const mfgm_quad_drift* q) {
    if (!q || q->d < 1 || q->d > kQD) return false;
    if (q->kind == 10) return q->d == 2;
    if (q->kind == 11) return q->nh >= 1 && 3 * q->nh + 1 <= kQP;
    return q->kind >= 12 && q->kind <= 15;
}
int nparam(const mfgm_quad_drift* q) { return q->kind == 11 ? 3 * q->nh + 1 : 2; }
}  // namespace

extern "C" {

int mfgm_quad_linearize(const mfgm_quad_drift* drift, int N, const double* mean, const double* cov, double* A, double* b, int* info,
                        void* stream) {
    if (!quad_ok(drift) || N < 0 || !mean || !cov || !A || !b || !info) return 1;
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_quad_linearize, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, *drift, N, mean, cov, A, b, info);
    MFGM_CHECK_LAUNCH();
    return 0;
}

size_t mfgm_quad_kl_scratch_doubles(int B, int T, int d, int nh) {
    if (B < 1 || T < 2 || d < 1 || d > kQD) return 0;
    const size_t nt = (size_t)B * (T - 1);
    return nt * (1 + (2 * d + 3 * d * d) + (size_t)std::max(2, 3 * nh + 1)) + B;
}

int mfgm_quad_kl(const mfgm_quad_drift* drift, int B, int T, const double* mu, const double* Sig, const double* Sub, double* kl,
                 double* g1, double* gd, double* gs, double* gtheta, double* scratch, int* info, void* stream) {
    if (!quad_ok(drift) || B < 1 || T < 2 || !mu || !Sig || !Sub || !kl || !scratch || !info) return 1;
    const bool grad = g1 != nullptr;
    if ((gd != nullptr) != grad || (gs != nullptr) != grad || (gtheta && !grad)) return 1;
    const int d = drift->d, np = nparam(drift);
    const size_t nt = (size_t)B * (T - 1);
    hipStream_t st = (hipStream_t)stream;
    double* klt = scratch;
    double* pieces = klt + nt;
    double* gth = pieces + nt * (2 * d + 3 * d * d);
    double* kl0 = gth + nt * np;
    const int blocks = (int)((nt + 63) / 64);
    if (grad) hipLaunchKernelGGL((k_quad_kl_transitions<true>), dim3(blocks), dim3(64), 0, st, *drift, B, T, mu, Sig, Sub, klt, pieces, gth, info);
    else hipLaunchKernelGGL((k_quad_kl_transitions<false>), dim3(blocks), dim3(64), 0, st, *drift, B, T, mu, Sig, Sub, klt, pieces, gth, info);
    MFGM_CHECK_LAUNCH();
    // node pass: KL[q(x0) || p(x0)] (+ the gradient assembly); without gradients only node 0 of every chain has work, but the pass is tiny
    if (grad) {
        hipLaunchKernelGGL(k_quad_kl_assemble, dim3((B * T + 63) / 64), dim3(64), 0, st, *drift, B, T, mu, Sig, pieces, kl0, g1, gd, gs, info);
    } else {
        hipLaunchKernelGGL(k_quad_kl0, dim3((B + 63) / 64), dim3(64), 0, st, *drift, B, T, mu, Sig, kl0, info);
    }
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_quad_sum, dim3(B), dim3(256), 0, st, T - 1, 1, klt, kl0, kl);
    MFGM_CHECK_LAUNCH();
    if (gtheta) {
        hipLaunchKernelGGL(k_quad_sum, dim3(B), dim3(256), 0, st, T - 1, np, gth, (const double*)nullptr, gtheta);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}

int mfgm_quad_esde(const mfgm_quad_drift* drift, int N, const double* mean, const double* cov, const double* A, const double* b, double* E,
                   double* dEdm, double* dEdS, double* dEdA, double* dEdb, double* gtheta, int* info, void* stream) {
    if (!quad_ok(drift) || N < 0 || !mean || !cov || !A || !b || !E || !info) return 1;
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_quad_esde, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, *drift, N, mean, cov, A, b, E, dEdm, dEdS, dEdA,
                       dEdb, gtheta, info);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_quad_vdp_lagrange(int B, int N, int d, double dt, double clip, const double* A, const double* dEdm, const double* dEdS,
                           const double* dobsm, const double* dobsS, double* psi, double* lam, void* stream) {
    if (B < 1 || N < 1 || d < 1 || d > kQD || !A || !dEdm || !dEdS || !dobsm || !dobsS || !psi || !lam) return 1;
    hipLaunchKernelGGL(k_quad_vdp_lagrange, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, N, d, dt, clip, A, dEdm, dEdS, dobsm,
                       dobsS, psi, lam);
    MFGM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
