// Closed-form SDE kernels of the CVI-DP path on packed arrays (KL and its gradients, linearisation, moment-array variants).
#include "mfgm_internal.h"
#include "mfgm_sweeps.h"
#include "mfgm_sde.h"

using namespace mfgm;

static_assert(sizeof(mfgm_sde_params) == sizeof(mfgm::SdeParams), "public and internal SDE parameter structs must match");

namespace {
template <int D, int KIND>
int sde_kl_impl(const Plan& P, int mode, const SdeParams& pr, const double* mu, const double* Sig, const double* Sub, double* kl,
                double* o1, double* od, double* os, double* q1, double* qd, double* qs, double* ws, int* info, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    double* part = kl ? ws + P.off_part[0] : nullptr;
    dim3 grid(lv.Lpad / 64), block(64);
    if (mode == 0) hipLaunchKernelGGL((k_sde_kl<D, 0, KIND>), grid, block, 0, st, lv, pr, mu, Sig, Sub, part, o1, od, os, q1, qd, qs, info);
    else if (mode == 1) hipLaunchKernelGGL((k_sde_kl<D, 1, KIND>), grid, block, 0, st, lv, pr, mu, Sig, Sub, part, o1, od, os, q1, qd, qs, info);
    else if (mode == 2) hipLaunchKernelGGL((k_sde_kl<D, 2, KIND>), grid, block, 0, st, lv, pr, mu, Sig, Sub, part, o1, od, os, q1, qd, qs, info);
    else hipLaunchKernelGGL((k_sde_kl<D, 3, KIND>), grid, block, 0, st, lv, pr, mu, Sig, Sub, part, o1, od, os, q1, qd, qs, info);
    MFGM_CHECK_LAUNCH();
    if (kl) {
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, part, lv.P, 0, kl, (double*)nullptr);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}
template <int D, int KIND>
int linearize_impl(const Plan& P, const SdeParams& pr, const double* mu, const double* Sig, double* A, double* off, double* chol,
                   hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    hipLaunchKernelGGL((k_linearize_cubic<D, KIND>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, pr, mu, Sig, A, off, chol);
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

int mfgm_packed_sde_kl(const mfgm_plan* plan, int mode, const mfgm_sde_params* prm, const double* mu, const double* Sig,
                       const double* Sub, double* kl, double* o1, double* od, double* os, double* q1, double* qd, double* qs,
                       void* ws, int* info, void* stream) {
    if (!plan || !prm || !mu || !Sig || !Sub || !info || !ws || mode < 0 || mode > 3) return 1;
    if (mode == 0 && !kl) return 1;
    if ((mode == 1 || mode == 2) && (!o1 || !od || !os)) return 1;
    if (mode >= 2 && (!q1 || !qd || !qs)) return 1;
    const Plan& P = plan->p;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    if (pr.kind != 0) { MFGM_DISPATCH_D4(P.d, (sde_kl_impl<DD, 1>(P, mode, pr, mu, Sig, Sub, kl, o1, od, os, q1, qd, qs, (double*)ws, info, st))); }
    MFGM_DISPATCH_D(P.d, (sde_kl_impl<DD, 0>(P, mode, pr, mu, Sig, Sub, kl, o1, od, os, q1, qd, qs, (double*)ws, info, st)));
}

int mfgm_packed_linearize_cubic(const mfgm_plan* plan, const mfgm_sde_params* prm, const double* mu, const double* Sig,
                                double* A, double* off, double* chol, void* stream) {
    if (!plan || !prm || !mu || !Sig || !A || !off || !chol) return 1;
    const Plan& P = plan->p;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    if (pr.kind != 0) { MFGM_DISPATCH_D4(P.d, (linearize_impl<DD, 1>(P, pr, mu, Sig, A, off, chol, st))); }
    MFGM_DISPATCH_D(P.d, (linearize_impl<DD, 0>(P, pr, mu, Sig, A, off, chol, st)));
}

}  // extern "C"

namespace {
template <int D, int KIND>
int sde_lean_impl(const Plan& P, int mode, const SdeParams& pr, const double* mom, const double* Sig, double* out, double* q1,
                  double* qd, double* qs, double* ws, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    dim3 grid(lv.Lpad / 64), block(64);
    if (mode == 0) {
        double* part = ws + P.off_part[0];
        hipLaunchKernelGGL((k_sde_lean<D, 0, KIND>), grid, block, 0, st, lv, pr, mom, Sig, part, q1, qd, qs);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, part, lv.P, 0, out, (double*)nullptr);
    } else {
        hipLaunchKernelGGL((k_sde_lean<D, 3, KIND>), grid, block, 0, st, lv, pr, mom, Sig, (double*)nullptr, q1, qd, qs);
    }
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

int mfgm_packed_sde_lean(const mfgm_plan* plan, int mode, const mfgm_sde_params* prm, const double* mom, const double* Sig,
                         double* kl_part, double* q1, double* qd, double* qs, void* ws, void* stream) {
    if (!plan || !prm || !mom || !ws || (mode != 0 && mode != 3)) return 1;
    if (mode == 0 && (!kl_part || !Sig)) return 1;
    if (mode == 3 && (!q1 || !qd || !qs)) return 1;
    const Plan& P = plan->p;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    if (pr.kind != 0) { MFGM_DISPATCH_D4(P.d, (sde_lean_impl<DD, 1>(P, mode, pr, mom, Sig, kl_part, q1, qd, qs, (double*)ws, st))); }
    MFGM_DISPATCH_D(P.d, (sde_lean_impl<DD, 0>(P, mode, pr, mom, Sig, kl_part, q1, qd, qs, (double*)ws, st)));
}

}  // extern "C"
