// Shared by the translation units of libmfgm: the opaque plan, launch-check and block-size dispatch macros, and the
// cross-unit entry points of the wide (8 < d <= 32) drivers.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mfgm.h"
#include "mfgm_layout.h"

struct mfgm_plan {
    mfgm::Plan p;
    // events of the cross-step pipelining of the CVI-DP loop (mfgm_cq_factor_pipelined), created on first use; 0: main stream reached
    // the point after which the side stream may run, 1: side stream finished the record the main stream is about to consume
    mutable hipEvent_t ev[2] = {nullptr, nullptr};
    mutable int ahead_region = 1;      // region (0: the plan's own level-1 inputs, 1: off_alt) the record made ahead lives in
    ~mfgm_plan() {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

namespace mfgm {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

#define MFGM_CHECK_LAUNCH()                        \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return 3;           \
    } while (0)

#define MFGM_DISPATCH_D(d, CALL)             \
    switch (d) {                             \
        case 1: { constexpr int DD = 1; return CALL; } \
        case 2: { constexpr int DD = 2; return CALL; } \
        case 3: { constexpr int DD = 3; return CALL; } \
        case 4: { constexpr int DD = 4; return CALL; } \
        case 5: { constexpr int DD = 5; return CALL; } \
        case 6: { constexpr int DD = 6; return CALL; } \
        case 7: { constexpr int DD = 7; return CALL; } \
        case 8: { constexpr int DD = 8; return CALL; } \
        default: return 1;                   \
    }

// the quadrature (non-polynomial drift) kernels are instantiated for d <= 4 only
#define MFGM_DISPATCH_D4(d, CALL)            \
    switch (d) {                             \
        case 1: { constexpr int DD = 1; return CALL; } \
        case 2: { constexpr int DD = 2; return CALL; } \
        case 3: { constexpr int DD = 3; return CALL; } \
        case 4: { constexpr int DD = 4; return CALL; } \
        default: return 1;                   \
    }

// mfgm_api_wide.hip
int wide_factor(const Plan& P, const double* Dg, const double* Sg, const double* rg, double aD, double aS, double aR,
                double* Lg, double* Gg, double* yg, double* logdet, double* quad, double* ws, int* info, hipStream_t st,
                int phase = -1, int form = 0, const double* site1 = nullptr, const double* site2 = nullptr, int site_packed = 0);
int wide_selinv(const Plan& P, const double* Lg, const double* Gg, const double* yg, double* Sig, double* Sub, double* x,
                double* ws, hipStream_t st, int form = 0);
int wide_stage(const Plan& P, int form, int which, const double* Dg, const double* Sg, const double* rg, double aD, double aS, double aR,
               double* Lg, double* Gg, double* yg, double* Sig, double* Sub, double* x, double* ws, int* info, hipStream_t st,
               const double* site1 = nullptr, const double* site2 = nullptr, int site_packed = 0);
int wide_ssm_to_naturals(const Plan& P, const double* A, const double* off, const double* chol, double cD, double cS, double* lin,
                         double* diag, double* sub, double* sumlogchol, double* ws, hipStream_t st);
int wide_kl_terms(const Plan& P, const double* Sig, const double* Sub, const double* mu, const double* Pd, const double* Ps,
                  double aD, double aS, const double* mup, double* trace, double* maha, double* ws, hipStream_t st);

// mfgm_api_mfma.hip: which = 0 reduce, 1 forward, 2 backward
struct WideArgs;
int mfma_launch(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st);
// mfgm_api_mfma_inv.hip: the same passes in inverse form (mfgm_mfma_inv.h)
int mfma_inv_launch(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st);
int mfma_ssm_to_naturals(int B, int T, int d, const double* A, const double* off, const double* chol, double cD, double cS,
                         double* lin, double* diag, double* sub, double* part, hipStream_t st);

}  // namespace mfgm
