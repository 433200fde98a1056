// Drivers of the wide (8 < d <= 32) path: one wavefront per segment, same level recursion as the narrow sweeps.
#include "mfgm_internal.h"
#include "mfgm_sweeps.h"
#include "mfgm_wide.h"

using namespace mfgm;

namespace {

// ---- wide (8 < d <= 32) drivers: same level recursion, one wavefront per segment ------------------------------------
template <int DM>
int wide_launch(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    dim3 grid((a.lv.L / a.lv.P) * a.nseg), block(64);   // chains x covered segments
#define KW(K) hipLaunchKernelGGL((K), grid, block, 0, st, a)
    if (which == 0) {
        if (has_rhs) { if (has_corr) KW((kw_reduce<DM, true, true>)); else KW((kw_reduce<DM, true, false>)); }
        else { if (has_corr) KW((kw_reduce<DM, false, true>)); else KW((kw_reduce<DM, false, false>)); }
    } else if (which == 1) {
        if (has_rhs) {
            if (has_corr) { if (has_up) KW((kw_forward<DM, true, true, true>)); else KW((kw_forward<DM, true, true, false>)); }
            else { if (has_up) KW((kw_forward<DM, true, false, true>)); else KW((kw_forward<DM, true, false, false>)); }
        } else {
            if (has_corr) { if (has_up) KW((kw_forward<DM, false, true, true>)); else KW((kw_forward<DM, false, true, false>)); }
            else { if (has_up) KW((kw_forward<DM, false, false, true>)); else KW((kw_forward<DM, false, false, false>)); }
        }
    } else {
        if (has_rhs) {
            if (has_up) { if (want_sub) KW((kw_backward<DM, true, true, true>)); else KW((kw_backward<DM, true, true, false>)); }
            else { if (want_sub) KW((kw_backward<DM, true, false, true>)); else KW((kw_backward<DM, true, false, false>)); }
        } else {
            if (has_up) { if (want_sub) KW((kw_backward<DM, false, true, true>)); else KW((kw_backward<DM, false, true, false>)); }
            else { if (want_sub) KW((kw_backward<DM, false, false, true>)); else KW((kw_backward<DM, false, false, false>)); }
        }
    }
#undef KW
    MFGM_CHECK_LAUNCH();
    return 0;
}

// MFMA tiles (mfgm_mfma.h) unless MFGM_WIDE_NO_MFMA=1 asks for the row-per-lane kernels (cross-checks)
bool use_mfma() {
    static const bool on = [] { const char* e = getenv("MFGM_WIDE_NO_MFMA"); return !(e && atoi(e) != 0); }();
    return on;
}

int wide_dispatch(int d, int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st, int form) {
    if (form == 1) return mfma_inv_launch(which, a, has_rhs, has_corr, has_up, want_sub, st);
    if (use_mfma()) return mfma_launch(which, a, has_rhs, has_corr, has_up, want_sub, st);
    if (d <= 16) return wide_launch<16>(which, a, has_rhs, has_corr, has_up, want_sub, st);
    return wide_launch<32>(which, a, has_rhs, has_corr, has_up, want_sub, st);
}

void wide_bind(const Plan& P, int l, double* ws, WideArgs& a) {
    const int K = P.nlevels - 1;
    a.lv = P.lv[l];
    a.d = P.d;
    const bool below = (P.shard_level > 0 && l < P.shard_level);      // a sharded chain works on its own segments below the exchange level
    a.seg_lo = below ? P.own_lo[l] : 0;
    a.nseg = below ? P.own_hi[l] - P.own_lo[l] : P.lv[l].P;
    a.store_left = below ? 1 : 0;
    if (l > 0) {
        a.Dg = ws + P.off_Dhat[l]; a.Dcorr = ws + P.off_Rsub[l]; a.Sg = ws + P.off_S[l];
        a.rg = ws + P.off_rhat[l]; a.rcorr = ws + P.off_rho[l];
        a.aD = a.aS = a.aR = 1.0;
        a.Lg = ws + P.off_L[l]; a.Gg = ws + P.off_G[l]; a.yg = ws + P.off_y[l];
        a.Sigg = ws + P.off_Sig[l]; a.Subg = nullptr; a.mug = ws + P.off_mu[l];
    }
    if (l < K) {
        a.up = P.lv[l + 1];
        a.uDhat = ws + P.off_Dhat[l + 1]; a.uRsub = ws + P.off_Rsub[l + 1]; a.uS = ws + P.off_S[l + 1];
        a.urhat = ws + P.off_rhat[l + 1]; a.urho = ws + P.off_rho[l + 1];
        a.uL = ws + P.off_L[l + 1]; a.uy = ws + P.off_y[l + 1]; a.uSig = ws + P.off_Sig[l + 1]; a.umu = ws + P.off_mu[l + 1];
    }
}

}  // namespace

namespace mfgm {

// carry-up of a sharded chain: the Gram corrections (Rsub, rho) this process produced, at the levels below the exchange level, for
// the separator on the left of its range belong to a node of the neighbouring process, which never sees them; that node is a
// separator at every level up to the exchange level, where its diagonal block only ever receives additive terms, so they are
// added to the exchange level's own correction of that node (one d x d block and one d-vector per level and chain).
struct CarryArgs {
    int nlev;                       // level slots 0 .. nlev-1 are the levels 1 .. exchange level; the last one receives the sums
    int nn[kMaxLevels], qq[kMaxLevels];
    double* Rsub[kMaxLevels];
    double* rho[kMaxLevels];
};
// save [B][EF + d]: the exchange level's own correction of that node before the sums went in; the forward sweep below the exchange
// level reconstructs its boundary state from exactly that (F_a = Ltil Ltil^T + R), so k_carry_restore puts it back once the levels
// from the exchange level up are factorised.
static __global__ void k_carry_up(int d, CarryArgs c, double* __restrict__ save) {
    const int b = blockIdx.x, EF = d * d, top = c.nlev - 1;
    for (int k = threadIdx.x; k < EF + d; k += blockDim.x) {
        double acc = 0.0;
        for (int j = 0; j < top; ++j) {
            acc += (k < EF) ? c.Rsub[j][((size_t)b * c.nn[j] + c.qq[j]) * EF + k] : c.rho[j][((size_t)b * c.nn[j] + c.qq[j]) * d + (k - EF)];
        }
        double* dst = (k < EF) ? &c.Rsub[top][((size_t)b * c.nn[top] + c.qq[top]) * EF + k]
                               : &c.rho[top][((size_t)b * c.nn[top] + c.qq[top]) * d + (k - EF)];
        save[(size_t)b * (EF + d) + k] = *dst;
        *dst += acc;
    }
}
static __global__ void k_carry_restore(int d, int n, int q, double* __restrict__ Rsub, double* __restrict__ rho,
                                       const double* __restrict__ save) {
    const int b = blockIdx.x, EF = d * d;
    for (int k = threadIdx.x; k < EF + d; k += blockDim.x) {
        const double v = save[(size_t)b * (EF + d) + k];
        if (k < EF) Rsub[((size_t)b * n + q) * EF + k] = v;
        else rho[((size_t)b * n + q) * d + (k - EF)] = v;
    }
}

// phase -1: the whole factorisation (plans that own every segment);  phase 0: zero the inputs of the levels 1 .. exchange level,
// run the reduces below the exchange level on the owned segments and carry the boundary corrections up (a sharded chain then sums
// the exchange level's inputs over the processes);  phase 1: everything after.
int wide_factor(const Plan& P, const double* Dg, const double* Sg, const double* rg, double aD, double aS, double aR,
                double* Lg, double* Gg, double* yg, double* logdet, double* quad, double* ws, int* info, hipStream_t st,
                int phase, int form, const double* site1, const double* site2, int site_packed) {
    if (form != 0 && form != 1) return 1;
    if ((site1 != nullptr) != (site2 != nullptr) || (site2 && (form != 1 || P.B != 1))) return 1;
    const bool has_rhs = (rg != nullptr) || (site1 != nullptr);
    const int K = P.nlevels - 1;
    const bool sharded = (P.shard_level > 0);
    const int X = sharded ? P.shard_level : 1;          // exchange level
    if (sharded && (phase < 0 || K == 0)) return 1;
    if (!sharded && phase >= 0) return 1;
    auto make = [&](int l) {
        WideArgs a;
        memset(&a, 0, sizeof(a));
        a.info = info;
        if (l == 0) {
            a.Dg = Dg; a.Sg = Sg; a.rg = rg; a.aD = aD; a.aS = aS; a.aR = aR;
            a.Lg = Lg; a.Gg = Gg; a.yg = yg;
            a.part = (logdet || quad) ? ws + P.off_part[0] : nullptr;
            a.site1 = site1; a.site2 = site2; a.site_packed = site_packed;
        }
        wide_bind(P, l, ws, a);
        return a;
    };
    if (phase <= 0) {
        if (phase == 0) {
            for (int l = 1; l <= X; ++l)
                if (hipMemsetAsync(ws + P.off_Dhat[l], 0, (P.off_L[l] - P.off_Dhat[l]) * sizeof(double), st) != hipSuccess) return 3;
        }
        const int upto = (phase == 0) ? X : std::min(1, K);      // reduces run here: levels 0 .. upto-1
        for (int l = 0; l < upto && l < K; ++l) {
            int rc = wide_dispatch(P.d, 0, make(l), has_rhs, l > 0, true, false, st, form);
            if (rc) return rc;
        }
        if (phase == 0) {
            if (X > 1 && P.own_lo[0] > 0) {
                // levels 1 .. X: node of the separator on the left = (first owned segment of the level below) - 1
                CarryArgs c;
                memset(&c, 0, sizeof(c));
                c.nlev = X;
                for (int l = 1; l <= X; ++l) {
                    c.nn[l - 1] = P.lv[l].n; c.qq[l - 1] = P.own_lo[l - 1] - 1;
                    c.Rsub[l - 1] = ws + P.off_Rsub[l]; c.rho[l - 1] = ws + P.off_rho[l];
                }
                hipLaunchKernelGGL(k_carry_up, dim3(P.B), dim3(256), 0, st, P.d, c, ws + P.off_part2);
                MFGM_CHECK_LAUNCH();
            }
            return 0;
        }
    }
    for (int l = X; l < K; ++l) {
        int rc = wide_dispatch(P.d, 0, make(l), has_rhs, true, true, false, st, form);
        if (rc) return rc;
    }
    if (sharded && (logdet || quad)) {
        if (hipMemsetAsync(ws + P.off_part[0], 0, 2 * (size_t)P.lv[0].Lpad * sizeof(double), st) != hipSuccess) return 3;
    }
    for (int l = K; l >= 0; --l) {
        if (sharded && l == X - 1 && X > 1 && P.own_lo[0] > 0) {
            // the levels from the exchange level up are factorised: back to this process's own correction of the separator on its left
            hipLaunchKernelGGL(k_carry_restore, dim3(P.B), dim3(256), 0, st, P.d, P.lv[X].n, P.own_lo[X - 1] - 1, ws + P.off_Rsub[X],
                               ws + P.off_rho[X], (const double*)(ws + P.off_part2));
            MFGM_CHECK_LAUNCH();
        }
        int rc = wide_dispatch(P.d, 1, make(l), has_rhs, l > 0, l < K, false, st, form);
        if (rc) return rc;
    }
    if (logdet || quad) {
        // on a sharded plan these are the partial sums over the owned segments
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, ws + P.off_part[0], P.lv[0].P, P.lv[0].Lpad, logdet, quad);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}

int wide_selinv(const Plan& P, const double* Lg, const double* Gg, const double* yg, double* Sig, double* Sub, double* x,
                double* ws, hipStream_t st, int form) {
    if (form != 0 && form != 1) return 1;
    const bool has_rhs = (yg != nullptr);
    const int K = P.nlevels - 1;
    for (int l = K; l >= 0; --l) {
        WideArgs a;
        memset(&a, 0, sizeof(a));
        if (l == 0) {
            a.Lg = const_cast<double*>(Lg); a.Gg = const_cast<double*>(Gg); a.yg = const_cast<double*>(yg);
            a.Sigg = Sig; a.Subg = Sub; a.mug = x;
        }
        wide_bind(P, l, ws, a);
        int rc = wide_dispatch(P.d, 2, a, has_rhs, false, l < K, l == 0 && Sub != nullptr, st, form);
        if (rc) return rc;
    }
    return 0;
}


// one level-0 kernel alone (which 0 reduce, 1 forward, 2 backward) with the coarser levels already in ws
int wide_stage(const Plan& P, int form, int which, const double* Dg, const double* Sg, const double* rg, double aD, double aS, double aR,
               double* Lg, double* Gg, double* yg, double* Sig, double* Sub, double* x, double* ws, int* info, hipStream_t st,
               const double* site1, const double* site2, int site_packed) {
    if ((form != 0 && form != 1) || which < 0 || which > 2) return 1;
    if ((site1 != nullptr) != (site2 != nullptr) || (site2 && (form != 1 || P.B != 1))) return 1;
    const int K = P.nlevels - 1;
    if (which == 0 && K == 0) return 1;
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.info = info;
    a.Dg = Dg; a.Sg = Sg; a.rg = rg; a.aD = aD; a.aS = aS; a.aR = aR;
    a.Lg = Lg; a.Gg = Gg; a.yg = yg;
    if (which == 2) { a.Sigg = Sig; a.Subg = Sub; a.mug = x; }
    a.site1 = site1; a.site2 = site2; a.site_packed = site_packed;
    wide_bind(P, 0, ws, a);
    if (which < 2) return wide_dispatch(P.d, which, a, rg != nullptr || site1 != nullptr, false, K > 0, false, st, form);
    return wide_dispatch(P.d, 2, a, yg != nullptr, false, K > 0, Sub != nullptr, st, form);
}

int wide_ssm_to_naturals(const Plan& P, const double* A, const double* off, const double* chol, double cD, double cS, double* lin,
                         double* diag, double* sub, double* sumlogchol, double* ws, hipStream_t st) {
    double* part = sumlogchol ? ws + P.off_part[0] : nullptr;
    if (use_mfma()) {
        int rc = mfma_ssm_to_naturals(P.B, P.T, P.d, A, off, chol, cD, cS, lin, diag, sub, part, st);
        if (rc) return rc;
    } else {
        dim3 grid(P.B * P.T), block(64);
#define S2N(DM_, LIN_) hipLaunchKernelGGL((kw_ssm_to_naturals<DM_, LIN_>), grid, block, 0, st, P.B, P.T, P.d, A, off, chol, cD, cS, lin, diag, sub, part)
        if (P.d <= 16) { if (lin) S2N(16, true); else S2N(16, false); }
        else { if (lin) S2N(32, true); else S2N(32, false); }
#undef S2N
        MFGM_CHECK_LAUNCH();
    }
    if (sumlogchol) {
        int rc = launch_sum_partials(part, P.T, 0, P.B, sumlogchol, nullptr, ws + P.off_part2, st);
        if (rc) return rc;
    }
    return 0;
}

int wide_kl_terms(const Plan& P, const double* Sig, const double* Sub, const double* mu, const double* Pd, const double* Ps,
                  double aD, double aS, const double* mup, double* trace, double* maha, double* ws, hipStream_t st) {
    double* part = ws + P.off_part[0];
    hipLaunchKernelGGL(kw_kl_terms, dim3(P.B * P.T), dim3(64), 0, st, P.B, P.T, P.d, Sig, Sub, mu, Pd, Ps, aD, aS, mup, part);
    MFGM_CHECK_LAUNCH();
    return launch_sum_partials(part, P.T, P.B * P.T, P.B, trace, maha, ws + P.off_part2, st);
}

}  // namespace mfgm
