// Multi-level drivers of the lane-per-segment sweeps (d <= 8): reduce / forward / backward over the partition levels.
#include "mfgm_internal.h"
#include "mfgm_sweeps.h"
#include "mfgm_rows.h"
#include "mfgm_girsanov.h"
#include "mfgm_cq.h"
#include "mfgm_kf.h"

using namespace mfgm;

namespace {

template <int D>
int launch_reduce(const SweepArgs& a, bool has_rhs, bool has_corr, hipStream_t st) {
    dim3 grid(a.lv.Lpad / 64), block(64);
    if (has_rhs) {
        if (has_corr) hipLaunchKernelGGL((k_reduce<D, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_reduce<D, true, false>), grid, block, 0, st, a);
    } else {
        if (has_corr) hipLaunchKernelGGL((k_reduce<D, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_reduce<D, false, false>), grid, block, 0, st, a);
    }
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D>
int launch_forward(const SweepArgs& a, bool has_rhs, bool has_corr, bool has_up, hipStream_t st) {
    dim3 grid(a.lv.Lpad / 64), block(64);
#define FW(R_, C_, U_) hipLaunchKernelGGL((k_forward<D, R_, C_, U_>), grid, block, 0, st, a)
    if (has_rhs) {
        if (has_corr) { if (has_up) FW(true, true, true); else FW(true, true, false); }
        else { if (has_up) FW(true, false, true); else FW(true, false, false); }
    } else {
        if (has_corr) { if (has_up) FW(false, true, true); else FW(false, true, false); }
        else { if (has_up) FW(false, false, true); else FW(false, false, false); }
    }
#undef FW
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D>
int launch_backward(const SweepArgs& a, bool has_rhs, bool has_up, bool want_sub, hipStream_t st, bool coarse = false) {
    dim3 grid(a.lv.Lpad / 64), block(64);
    const bool mom = (a.momg != nullptr);
    if (coarse) {
        // a level above the finest one: workspace arrays (coarse_off), marginals only
        if (want_sub || mom || a.Gg == nullptr) return 1;
#define BWC(R_, U_) hipLaunchKernelGGL((k_backward<D, R_, U_, false, false, false, true>), grid, block, 0, st, a)
        if (has_rhs) { if (has_up) BWC(true, true); else BWC(true, false); }
        else { if (has_up) BWC(false, true); else BWC(false, false); }
#undef BWC
        MFGM_CHECK_LAUNCH();
        return 0;
    }
#define BW(R_, U_, S_, M_) hipLaunchKernelGGL((k_backward<D, R_, U_, S_, M_>), grid, block, 0, st, a)
    if (a.Gg == nullptr && a.Sg != nullptr) {
        // level 0 rebuilt from the input sub-diagonal blocks (the forward pass stored no L_{t+1,t}); means wanted, no Sub
        if (want_sub || !has_rhs) return 1;
        if (mom) {
            if (has_up) hipLaunchKernelGGL((k_backward<D, true, true, false, true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_backward<D, true, false, false, true, true>), grid, block, 0, st, a);
        } else {
            if (has_up) hipLaunchKernelGGL((k_backward<D, true, true, false, false, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_backward<D, true, false, false, false, true>), grid, block, 0, st, a);
        }
    } else if (mom) {
        // the moment array needs the means: has_rhs is guaranteed by the entry point
        if (has_up) { if (want_sub) BW(true, true, true, true); else BW(true, true, false, true); }
        else { if (want_sub) BW(true, false, true, true); else BW(true, false, false, true); }
    } else if (has_rhs) {
        if (has_up) { if (want_sub) BW(true, true, true, false); else BW(true, true, false, false); }
        else { if (want_sub) BW(true, false, true, false); else BW(true, false, false, false); }
    } else {
        if (has_up) { if (want_sub) BW(false, true, true, false); else BW(false, true, false, false); }
        else { if (want_sub) BW(false, false, true, false); else BW(false, false, false, false); }
    }
#undef BW
    MFGM_CHECK_LAUNCH();
    return 0;
}

// First level handled by the fused coarse-level kernels (k_coarse_factor / k_coarse_backward): the lowest level >= 1 whose chains
// have at most `MFGM_FUSE_P` segments (default 128); nlevels when fusing is off (MFGM_COARSE_FUSED=0) or no such level exists.
// Levels 1 .. l0-1 keep one launch per level and pass.  The fused kernels could take 256 segments per chain (one lane each in a
// 256-thread workgroup), but four wavefronts of one workgroup share their CU's texture-address unit: by the per-phase stamps of
// DESIGN 5c a 256-segment level costs 1.6 x per block step what a 64-segment level costs inside the fused kernel, while as a launch of
// its own its wavefronts spread over the idle CUs -- three more launches per refresh, 14 us less at the headline size.
int coarse_fuse_from(const Plan& P) {
    static const int enabled = [] { const char* e = getenv("MFGM_COARSE_FUSED"); return (e && atoi(e) == 0) ? 0 : 1; }();
    static const int maxp = [] { const char* e = getenv("MFGM_FUSE_P"); int v = e ? atoi(e) : 0; return v > 0 ? std::min(v, 256) : 128; }();
    if (!enabled || P.nlevels < 2) return P.nlevels;
    for (int l = 1; l < P.nlevels; ++l)
        if (P.lv[l].P <= maxp) return l;
    return P.nlevels;
}

// Levels whose chains have at most this many segments run on the 16-lanes-per-segment row bodies (mfgm_rows.h) inside the fused
// kernels; MFGM_COARSE_ROWS=0 keeps every level on the lane-per-segment bodies (the cross-check of tests/test_gpu_sweeps.py), a value
// in 1 .. kRowsMaxP lowers the threshold.
int coarse_rows_p() {
    // read per launch (a getenv, next to a kernel launch): the cross-check test flips it inside one process
    const char* e = getenv("MFGM_COARSE_ROWS");
    if (!e) return kRowsMaxP;
    return std::max(0, std::min(atoi(e), kRowsMaxP));
}

template <int D>
int launch_coarse_backward(const Plan& P, int lf, bool has_rhs, double* ws, hipStream_t st) {
    if (has_rhs) hipLaunchKernelGGL((k_coarse_backward<D, true>), dim3(P.B), dim3(kCoarseBlock), 0, st, P, lf, ws, coarse_rows_p());
    else hipLaunchKernelGGL((k_coarse_backward<D, false>), dim3(P.B), dim3(kCoarseBlock), 0, st, P, lf, ws, coarse_rows_p());
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D>
int factor_impl(const Plan& P, const double* Dg, const double* Sg, const double* rg, double aD, double aS, double aR,
                double* Lg, double* Gg, double* yg, double* logdet, double* quad, double* ws, int* info,
                hipStream_t st, int only_stage = -1, int only_level = -1, bool partials_only = false) {
    // partials_only: the per-lane log|L| / |y|^2 partials are left in ws + off_part[0] for the caller to sum (kf_elbo_impl)
    const bool has_rhs = (rg != nullptr);
    const int K = P.nlevels - 1;  // top level index (single segment per chain)
    auto make = [&](int l) {
        SweepArgs a;
        memset(&a, 0, sizeof(a));
        a.lv = P.lv[l];
        a.info = info;
        if (l == 0) {
            a.Dg = Dg; a.Sg = Sg; a.rg = rg; a.aD = aD; a.aS = aS; a.aR = aR;
            a.Lg = Lg; a.Gg = Gg; a.yg = yg;
            a.part = (logdet || quad || partials_only) ? ws + P.off_part[0] : nullptr;
        } else {
            bind_level_inputs(P, l, ws, a);
        }
        if (l < K) bind_up(P, l, ws, a);
        return a;
    };
    // levels lf .. K in one launch (reduce lf .. K-1, forward K .. lf); single-kernel profiling calls keep one launch per level
    const int lf = (only_stage >= 0) ? P.nlevels : coarse_fuse_from(P);
    for (int l = 0; l < K && l < lf; ++l) {
        if (only_stage >= 0 && !(only_stage == 0 && only_level == l)) continue;
        SweepArgs a = make(l);
        int rc = launch_reduce<D>(a, has_rhs, l > 0, st);
        if (rc) return rc;
    }
    if (lf <= K) {
        if (has_rhs) hipLaunchKernelGGL((k_coarse_factor<D, true>), dim3(P.B), dim3(kCoarseBlock), 0, st, P, lf, ws, info, coarse_rows_p());
        else hipLaunchKernelGGL((k_coarse_factor<D, false>), dim3(P.B), dim3(kCoarseBlock), 0, st, P, lf, ws, info, coarse_rows_p());
        MFGM_CHECK_LAUNCH();
    }
    for (int l = std::min(K, lf - 1); l >= 0; --l) {
        if (only_stage >= 0 && !(only_stage == 1 && only_level == l)) continue;
        SweepArgs a = make(l);
        int rc = launch_forward<D>(a, has_rhs, l > 0, l < K, st);
        if (rc) return rc;
    }
    if (only_stage < 0 && (logdet || quad)) {
        if (launch_sum_partials(ws + P.off_part[0], P.lv[0].P, P.lv[0].Lpad, P.B, logdet, quad, P.B == 1 ? ws + P.off_part2 : nullptr, st))
            return 3;
    }
    return 0;
}

template <int D>
int selinv_impl(const Plan& P, const double* Lg, const double* Gg, const double* yg, double* Sig, double* Sub,
                double* x, double* ws, hipStream_t st, int only_level = -1, double* mom = nullptr, const double* Sg = nullptr,
                double aS = 1.0) {
    const bool has_rhs = (yg != nullptr);
    const int K = P.nlevels - 1;
    const int lf = (only_level >= 0) ? P.nlevels : coarse_fuse_from(P);
    if (lf <= K) {
        int rc = launch_coarse_backward<D>(P, lf, has_rhs, ws, st);
        if (rc) return rc;
    }
    for (int l = std::min(K, lf - 1); l >= 0; --l) {
        if (only_level >= 0 && only_level != l) continue;
        SweepArgs a;
        memset(&a, 0, sizeof(a));
        a.lv = P.lv[l];
        if (l == 0) {
            a.Lg = const_cast<double*>(Lg); a.Gg = const_cast<double*>(Gg); a.yg = const_cast<double*>(yg);
            a.Sigg = Sig; a.Subg = Sub; a.mug = x; a.momg = mom;
            a.Sg = Sg; a.aS = aS;          // non-null Sg: level 0 reads S instead of L_{t+1,t} (see k_backward, USE_S)
        } else {
            bind_level_inputs(P, l, ws, a);
        }
        if (l < K) bind_up(P, l, ws, a);
        int rc = launch_backward<D>(a, has_rhs, l < K, l == 0 && Sub != nullptr, st, l > 0);
        if (rc) return rc;
    }
    return 0;
}

template <int D>
int selinv_girsanov_impl(const Plan& P, const double* Lg, const double* Sg, double aS, const double* yg, const SdeParams& pr,
                         const GirsanovArgs& g, double* ws, hipStream_t st, int only_level) {
    // coarser levels exactly as in a plain selected inverse, then the fused level-0 sweep
    const int K = P.nlevels - 1;
    const int lf = (only_level >= 0) ? P.nlevels : coarse_fuse_from(P);
    if (lf <= K) {
        int rc = launch_coarse_backward<D>(P, lf, true, ws, st);
        if (rc) return rc;
    }
    for (int l = std::min(K, lf - 1); l >= 1; --l) {
        if (only_level >= 0 && only_level != l) continue;
        SweepArgs a;
        memset(&a, 0, sizeof(a));
        a.lv = P.lv[l];
        bind_level_inputs(P, l, ws, a);
        if (l < K) bind_up(P, l, ws, a);
        int rc = launch_backward<D>(a, true, l < K, false, st, true);
        if (rc) return rc;
    }
    if (only_level > 0) return 0;
    SweepArgs a;
    memset(&a, 0, sizeof(a));
    a.lv = P.lv[0];
    a.Lg = const_cast<double*>(Lg); a.yg = const_cast<double*>(yg); a.Sg = Sg; a.aS = aS;
    bind_up(P, 0, ws, a);
    GirsanovArgs ga = g;
    ga.fix = ws + P.off_part[0];
    dim3 grid(a.lv.Lpad / 64), block(64);
    hipLaunchKernelGGL((k_backward_girsanov<D>), grid, block, 0, st, a, pr, ga);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_girsanov_fixup<D>), grid, block, 0, st, a.lv, ga);
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D>
int selinv_kl_impl(const Plan& P, const double* Lg, const double* Sg, double aS, const double* yg, const SdeParams& pr, double* Sig,
                   double* x, double* kl, double* ws, hipStream_t st, int only_level) {
    const int K = P.nlevels - 1;
    const int lf = (only_level >= 0) ? P.nlevels : coarse_fuse_from(P);
    if (lf <= K) {
        int rc = launch_coarse_backward<D>(P, lf, true, ws, st);
        if (rc) return rc;
    }
    for (int l = std::min(K, lf - 1); l >= 1; --l) {
        if (only_level >= 0 && only_level != l) continue;
        SweepArgs a;
        memset(&a, 0, sizeof(a));
        a.lv = P.lv[l];
        bind_level_inputs(P, l, ws, a);
        if (l < K) bind_up(P, l, ws, a);
        int rc = launch_backward<D>(a, true, l < K, false, st, true);
        if (rc) return rc;
    }
    if (only_level > 0) return 0;
    SweepArgs a;
    memset(&a, 0, sizeof(a));
    a.lv = P.lv[0];
    a.Lg = const_cast<double*>(Lg); a.yg = const_cast<double*>(yg); a.Sg = Sg; a.aS = aS;
    a.Sigg = Sig; a.mug = x;
    a.part = ws + P.off_part[0];
    bind_up(P, 0, ws, a);
    hipLaunchKernelGGL((k_backward_kl<D>), dim3(a.lv.Lpad / 64), dim3(64), 0, st, a, pr);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, a.part, a.lv.P, 0, kl, (double*)nullptr);
    MFGM_CHECK_LAUNCH();
    return 0;
}

// ---- CVI-DP sweeps on the structured ("cq") state: level 0 from mfgm_cq.h, coarser levels as usual ------------------------------------
CqArgs cq_args(const mfgm_cq_state* q) {
    CqArgs c;
    memset(&c, 0, sizeof(c));
    c.dyn = q->dyn; c.dOff = q->d_off; c.sOff = q->s_off; c.p0off = q->p0_off;
    c.slot = q->slot; c.site_lin = q->site_lin; c.site_sym = q->site_sym;
    return c;
}

// Pipelining across the steps of the CVI-DP loop (mfgm_cq_factor_pipelined).  The workspace holds the level-1 INPUT region (Dhat,
// Rsub, S, rhat, rho: what a level-0 reduce writes) twice; a factorisation reads one copy while the level-0 reduce of the state the
// caller knows one factorisation ahead fills the other, on the stream `side`, next to this factorisation's level-0 forward sweep.
struct CqPipe {
    const mfgm_plan* owner = nullptr;
    bool use_ahead = false;             // this state's separator system was made ahead: skip the level-0 reduce
    const CqArgs* next = nullptr;       // make the separator system of this state ahead
    hipStream_t side = nullptr;
};
// the plan with its level-1 inputs bound to region r (0: its own, 1: off_alt)
Plan with_l1_region(const Plan& P, int r) {
    Plan Q = P;
    if (r) {
        const size_t sh = P.off_alt - P.off_Dhat[1];
        Q.off_Dhat[1] += sh; Q.off_Rsub[1] += sh; Q.off_S[1] += sh; Q.off_rhat[1] += sh; Q.off_rho[1] += sh;
    }
    return Q;
}

template <int D>
int cq_factor_impl(const Plan& P0, const CqArgs& q, double* Lg, double* yg, double* logdet, double* quad, double* ws, int* info,
                   hipStream_t st, int only_stage = -1, const CqPipe* pipe = nullptr) {
    const int region = (pipe && pipe->use_ahead) ? pipe->owner->ahead_region : 0;
    const Plan P = with_l1_region(P0, region);
    const int K = P.nlevels - 1;
    SweepArgs a0;
    memset(&a0, 0, sizeof(a0));
    a0.lv = P.lv[0];
    a0.info = info;
    a0.aD = -2.0; a0.aS = -1.0; a0.aR = 1.0;          // natural parameters -> precision
    a0.Lg = Lg; a0.yg = yg;
    a0.part = (logdet || quad) ? ws + P.off_part[0] : nullptr;
    bind_up(P, 0, ws, a0);
    dim3 grid(a0.lv.Lpad / 64), block(64);
    if (pipe && pipe->use_ahead) {
        // the separator system of this state was made ahead in the region bound above; when that happened on a side stream: wait for it
        if (pipe->side) {
            if (hipEventRecord(pipe->owner->ev[1], pipe->side) != hipSuccess) return 3;
            if (hipStreamWaitEvent(st, pipe->owner->ev[1], 0) != hipSuccess) return 3;
        }
    } else if (only_stage < 0 || only_stage == 0) {
        hipLaunchKernelGGL((k_reduce_cq<D>), grid, block, 0, st, a0, q);
        MFGM_CHECK_LAUNCH();
    }
    if (only_stage < 0 || only_stage == 2) {         // stage 2 (profiling): the levels above the finest one alone
        const int lf = coarse_fuse_from(P);
        for (int l = 1; l < K && l < lf; ++l) {
            SweepArgs a = coarse_level_args(P, l, ws, info);
            int rc = launch_reduce<D>(a, true, true, st);
            if (rc) return rc;
        }
        if (lf <= K) {
            hipLaunchKernelGGL((k_coarse_factor<D, true>), dim3(P.B), dim3(kCoarseBlock), 0, st, P, lf, ws, info, coarse_rows_p());
            MFGM_CHECK_LAUNCH();
        }
        for (int l = std::min(K, lf - 1); l >= 1; --l) {
            SweepArgs a = coarse_level_args(P, l, ws, info);
            int rc = launch_forward<D>(a, true, true, l < K, st);
            if (rc) return rc;
        }
    }
    if (pipe && pipe->next && !pipe->side && D > 6) {
        // d = 7, 8: the two-wavefront kernel does not fit 256 registers; without a side stream the reduce made ahead simply follows the
        // forward sweep on this stream (correct, nothing overlapped)
        SweepArgs an = a0;
        bind_up(with_l1_region(P0, 1 - region), 0, ws, an);
        pipe->owner->ahead_region = 1 - region;
        an.part = nullptr;
        if (a0.lv.nt >= 2) hipLaunchKernelGGL((k_forward_cq<D, 2>), grid, block, 0, st, a0, q);
        else hipLaunchKernelGGL((k_forward_cq<D>), grid, block, 0, st, a0, q);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL((k_reduce_cq_lean<D>), grid, block, 0, st, an, *pipe->next);
        MFGM_CHECK_LAUNCH();
        if (only_stage < 0 && (logdet || quad)) {
            hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, ws + P.off_part[0], P.lv[0].P, P.lv[0].Lpad, logdet, quad);
            MFGM_CHECK_LAUNCH();
        }
        return 0;
    }
    if (pipe && pipe->next && !pipe->side) {
        // forward sweep of this factorisation and level-0 reduce of the NEXT one as the two wavefronts of one workgroup per tile
        // (k_forward_reduce_cq): the records are read from HBM once; the reduce's separator system goes to the other level-1 region
        SweepArgs an = a0;
        bind_up(with_l1_region(P0, 1 - region), 0, ws, an);
        pipe->owner->ahead_region = 1 - region;
        CqArgs qf = q;
        qf.site_lin2 = pipe->next->site_lin; qf.site_sym2 = pipe->next->site_sym;
        qf.pre_Dhat = an.uDhat; qf.pre_Rsub = an.uRsub; qf.pre_S = an.uS; qf.pre_rhat = an.urhat; qf.pre_rho = an.urho;
        if (a0.lv.nt >= 1) hipLaunchKernelGGL((k_forward_reduce_cq<D, 1>), grid, dim3(128), 0, st, a0, qf);
        else hipLaunchKernelGGL((k_forward_reduce_cq<D>), grid, dim3(128), 0, st, a0, qf);
        MFGM_CHECK_LAUNCH();
        if (only_stage < 0 && (logdet || quad)) {
            hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, ws + P.off_part[0], P.lv[0].P, P.lv[0].Lpad, logdet, quad);
            MFGM_CHECK_LAUNCH();
        }
        return 0;
    }
    if (pipe && pipe->next) {
        // the level-0 reduce of the NEXT factorisation's state, into the other level-1 region, on the side stream: from here on the main
        // stream runs the bandwidth-bound level-0 forward sweep (193 registers), next to whose wavefronts the lean reduce (302) fits
        if (hipEventRecord(pipe->owner->ev[0], st) != hipSuccess) return 3;
        if (hipStreamWaitEvent(pipe->side, pipe->owner->ev[0], 0) != hipSuccess) return 3;
        SweepArgs an = a0;
        bind_up(with_l1_region(P0, 1 - region), 0, ws, an);
        pipe->owner->ahead_region = 1 - region;
        an.part = nullptr;
        hipLaunchKernelGGL((k_reduce_cq_lean<D>), grid, block, 0, pipe->side, an, *pipe->next);
        MFGM_CHECK_LAUNCH();
    }
    if (only_stage < 0 || only_stage == 1) {
        if (a0.lv.nt >= 2) hipLaunchKernelGGL((k_forward_cq<D, 2>), grid, block, 0, st, a0, q);
        else hipLaunchKernelGGL((k_forward_cq<D>), grid, block, 0, st, a0, q);
        MFGM_CHECK_LAUNCH();
    }
    if (only_stage < 0 && (logdet || quad)) {
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, ws + P.off_part[0], P.lv[0].P, P.lv[0].Lpad, logdet, quad);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}

template <int D>
int cq_coarse_backward(const Plan& P, double* ws, hipStream_t st) {
    const int K = P.nlevels - 1;
    const int lf = coarse_fuse_from(P);
    if (lf <= K) {
        int rc = launch_coarse_backward<D>(P, lf, true, ws, st);
        if (rc) return rc;
    }
    for (int l = std::min(K, lf - 1); l >= 1; --l) {
        SweepArgs a = coarse_level_args(P, l, ws, nullptr);
        int rc = launch_backward<D>(a, true, l < K, false, st, true);
        if (rc) return rc;
    }
    return 0;
}

template <int D>
int cq_selinv_girsanov_impl(const Plan& P, const CqArgs& q, const double* Lg, const double* yg, const SdeParams& pr, double* ws,
                            hipStream_t st, int only_level) {
    if (only_level != 0) {
        int rc = cq_coarse_backward<D>(P, ws, st);
        if (rc) return rc;
    }
    SweepArgs a;
    memset(&a, 0, sizeof(a));
    a.lv = P.lv[0];
    a.Lg = const_cast<double*>(Lg); a.yg = const_cast<double*>(yg); a.aS = -1.0;
    bind_up(P, 0, ws, a);
    double* fix = ws + P.off_part[0];
    dim3 grid(a.lv.Lpad / 64), block(64);
    if (a.lv.nt >= 2) hipLaunchKernelGGL((k_backward_girsanov_cq<D, 2>), grid, block, 0, st, a, pr, q, fix);
    else hipLaunchKernelGGL((k_backward_girsanov_cq<D>), grid, block, 0, st, a, pr, q, fix);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_girsanov_fixup_cq<D>), grid, block, 0, st, a.lv, q.dyn_out, (const double*)fix);
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D>
int cq_selinv_kl_impl(const Plan& P, const CqArgs& q, const double* Lg, const double* yg, const SdeParams& pr, double* Sig, double* x,
                      double* kl, double* ws, hipStream_t st, int only_level) {
    if (only_level != 0) {
        int rc = cq_coarse_backward<D>(P, ws, st);
        if (rc) return rc;
    }
    if (only_level > 0) return 0;                    // profiling: the levels above the finest one alone
    SweepArgs a;
    memset(&a, 0, sizeof(a));
    a.lv = P.lv[0];
    a.Lg = const_cast<double*>(Lg); a.yg = const_cast<double*>(yg); a.aS = -1.0;
    a.Sigg = Sig; a.mug = x;
    a.part = ws + P.off_part[0];
    bind_up(P, 0, ws, a);
    if (a.lv.nt >= 2) hipLaunchKernelGGL((k_backward_kl_cq<D, 2>), dim3(a.lv.Lpad / 64), dim3(64), 0, st, a, pr, q);
    else hipLaunchKernelGGL((k_backward_kl_cq<D>), dim3(a.lv.Lpad / 64), dim3(64), 0, st, a, pr, q);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, a.part, a.lv.P, 0, kl, (double*)nullptr);
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D>
int cq_pack_impl(const Plan& P, const double* lin, const double* diag, const double* sub, double* dyn, double* range, hipStream_t st) {
    hipLaunchKernelGGL((k_cq_pack<D>), dim3(P.lv[0].Lpad / 64), dim3(64), 0, st, P.lv[0], lin, diag, sub, dyn, range);
    MFGM_CHECK_LAUNCH();
    return 0;
}
template <int D>
int cq_unpack_impl(const Plan& P, const CqArgs& q, double* lin, double* diag, double* sub, hipStream_t st) {
    hipLaunchKernelGGL((k_cq_unpack<D>), dim3(P.lv[0].Lpad / 64), dim3(64), 0, st, P.lv[0], q, lin, diag, sub);
    MFGM_CHECK_LAUNCH();
    return 0;
}
template <int D>
int mvn_ve_compact_impl(int B, int n_per, const double* mu, const double* cov, const double* y, const double* Sinv, double cst,
                        double* ve, hipStream_t st) {
    const int nb = (n_per + 255) / 256;
    hipLaunchKernelGGL((k_mvn_ve_compact<D>), dim3(nb, B), dim3(256), 0, st, n_per, mu, cov, y, Sinv, cst, ve);
    MFGM_CHECK_LAUNCH();
    return 0;
}

// ---- Kalman filter with sites, time-invariant emission (mfgm_kf.h) ------------------------------------------------------------------
template <int D, int O>
int kf_assemble_impl(const Plan& P, const KfArgs& k, const double* Pd, const double* plin, double* Dg, double* rg, double* t1,
                     double* ldR, double* ws, hipStream_t st) {
    double* part = (t1 || ldR) ? ws + P.off_part[0] : nullptr;
    hipLaunchKernelGGL((k_kf_assemble<D, O>), dim3(P.lv[0].Lpad / 64), dim3(64), 0, st, P.lv[0], P.T, k, Pd, plin, Dg, rg, part);
    MFGM_CHECK_LAUNCH();
    if (part) {
        // (a single chain has thousands of per-lane partials: two deterministic stages when it pays)
        if (launch_sum_partials(part, P.lv[0].P, P.lv[0].Lpad, P.B, t1, ldR, P.B == 1 ? ws + P.off_part2 : nullptr, st)) return 3;
    }
    return 0;
}
// The four per-chain sums of the Kalman log-likelihood with sites and its assembly (kalman_filter.py:229-255) in two launches instead
// of five: part = [log|L|, |y|^2, t1, ldR] x [Lpad] per-lane partials; stage 1: kElboSplit blocks per chain sum a slice of each
// array (fixed order: deterministic), stage 2: one block adds those and forms
//   ll_b = cst - 1/2 t1 + 1/2 |y|^2 - sumlogchol_b - log|L| + 1/2 ldR      (NaN when a pivot block was not positive definite).
constexpr int kElboSplit = 64;
static __global__ __launch_bounds__(256) void k_kf_elbo_stage1(const double* __restrict__ part, int P, int Lpad, double* __restrict__ scratch) {
    __shared__ double sh[4][4];
    const int g = blockIdx.x, b = blockIdx.y, B = gridDim.y;
    const int slice = (P + kElboSplit - 1) / kElboSplit, lo = g * slice, hi = min(P, lo + slice);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p = lo + threadIdx.x; p < hi; p += blockDim.x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += part[(size_t)q * Lpad + (size_t)b * P + p];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off, 64);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = s[q];
    }
    __syncthreads();
    if (threadIdx.x < 4) scratch[((size_t)threadIdx.x * B + b) * kElboSplit + g] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}
static __global__ __launch_bounds__(64) void k_kf_elbo_stage2(const double* __restrict__ scratch, int B, double cst, const double* __restrict__ sumlogchol,
                                                             const int* __restrict__ info, double* __restrict__ terms, double* __restrict__ ll,
                                                             double* __restrict__ total) {
    // one wavefront; lane g holds the stage-1 partial g of each term, chains one after the other (B is small on this path)
    const bool bad = info && *info != 0;
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        double s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[q] = scratch[((size_t)q * B + b) * kElboSplit + threadIdx.x];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off, 64);
            s[q] = __shfl(s[q], 0, 64);
        }
        double v = cst - 0.5 * s[2] + 0.5 * s[1] - sumlogchol[b] - s[0] + 0.5 * s[3];
        if (bad) v = __builtin_nan("");
        if (threadIdx.x == 0) {
            if (terms) { terms[b] = s[2]; terms[B + b] = s[3]; terms[2 * B + b] = s[0]; terms[3 * B + b] = s[1]; }   // (t1, ldR, log|L|, |y|^2)
            if (ll) ll[b] = v;
        }
        tot += v;
    }
    if (threadIdx.x == 0 && total) *total = tot;
}

template <int D, int O>
int kf_elbo_impl(const Plan& P, const KfArgs& k, const double* Pd, const double* Ps, double* Dg, double* rg, double* L, double* y,
                 const double* sumlogchol, double cst, double* terms, double* ll, double* total, double* ws, int* info, hipStream_t st) {
    double* part = ws + P.off_part[0];
    hipLaunchKernelGGL((k_kf_assemble<D, O>), dim3(P.lv[0].Lpad / 64), dim3(64), 0, st, P.lv[0], P.T, k, Pd, (const double*)nullptr, Dg, rg,
                       part + 2 * (size_t)P.lv[0].Lpad);
    MFGM_CHECK_LAUNCH();
    int rc = factor_impl<D>(P, Dg, Ps, rg, 1.0, 1.0, 1.0, L, nullptr, y, nullptr, nullptr, ws, info, st, -1, -1, true);
    if (rc) return rc;
    double* scratch = ws + P.off_part2;
    hipLaunchKernelGGL(k_kf_elbo_stage1, dim3(kElboSplit, P.B), dim3(256), 0, st, (const double*)part, P.lv[0].P, P.lv[0].Lpad, scratch);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_kf_elbo_stage2, dim3(1), dim3(64), 0, st, (const double*)scratch, P.B, cst, sumlogchol, (const int*)info, terms, ll, total);
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D, int O>
int kf_loglik_impl(const Plan& P, const KfArgs& k, const double* Pd, const double* Ps, double* Dg, double* rg, double* L, double* y,
                   double* t1, double* ldR, double* logdet, double* quad, double* ws, int* info, hipStream_t st) {
    int rc = kf_assemble_impl<D, O>(P, k, Pd, nullptr, Dg, rg, t1, ldR, ws, st);
    if (rc) return rc;
    return factor_impl<D>(P, Dg, Ps, rg, 1.0, 1.0, 1.0, L, nullptr, y, logdet, quad, ws, info, st);
}
template <int D, int O>
int kf_predict_impl(const Plan& P, const KfArgs& k, const double* Pd, const double* Ps, const double* plin, double* Dg, double* rg,
                    double* L, double* y, double* Sig, double* x, double* Fmu, double* Fvar, double* ws, int* info, hipStream_t st) {
    int rc = kf_assemble_impl<D, O>(P, k, Pd, plin, Dg, rg, nullptr, nullptr, ws, st);
    if (rc) return rc;
    rc = factor_impl<D>(P, Dg, Ps, rg, 1.0, 1.0, 1.0, L, nullptr, y, nullptr, nullptr, ws, info, st);
    if (rc) return rc;
    rc = selinv_impl<D>(P, L, nullptr, y, Sig, nullptr, x, ws, st, -1, nullptr, Ps, 1.0);
    if (rc) return rc;
    hipLaunchKernelGGL((k_kf_project<D, O>), dim3(P.lv[0].Lpad / 64), dim3(64), 0, st, P.lv[0], P.T, k, (const double*)x,
                       (const double*)Sig, Fmu, Fvar);
    MFGM_CHECK_LAUNCH();
    return 0;
}
template <int D, int O>
int kf_predict_factored_impl(const Plan& P, const KfArgs& k, const double* Ps, const double* L, const double* y, double* Sig, double* x,
                             double* Fmu, double* Fvar, double* ws, hipStream_t st) {
    int rc = selinv_impl<D>(P, L, nullptr, y, Sig, nullptr, x, ws, st, -1, nullptr, Ps, 1.0);
    if (rc) return rc;
    hipLaunchKernelGGL((k_kf_project<D, O>), dim3(P.lv[0].Lpad / 64), dim3(64), 0, st, P.lv[0], P.T, k, (const double*)x,
                       (const double*)Sig, Fmu, Fvar);
    MFGM_CHECK_LAUNCH();
    return 0;
}
bool kf_args(const mfgm_plan* plan, const mfgm_kf_sites* sv, int mode, KfArgs& k) {
    if (!plan || !sv || !sv->nat1 || !sv->nat2 || plan->p.wide) return false;
    const Plan& P = plan->p;
    if (sv->o < 1 || sv->o > 2 || (sv->site_batch != 1 && sv->site_batch != P.B)) return false;
    memset(&k, 0, sizeof(k));
    memcpy(k.H, sv->H, sizeof(double) * sv->o * P.d);
    k.nat1 = sv->nat1; k.nat2 = sv->nat2; k.Hmu = sv->Hmu; k.site_batch = sv->site_batch; k.mode = mode;
    return true;
}
#define MFGM_DISPATCH_DO(d, o, CALL)                                      \
    if ((o) == 1) { constexpr int OO = 1; MFGM_DISPATCH_D(d, CALL); }     \
    else { constexpr int OO = 2; MFGM_DISPATCH_D(d, CALL); }

bool cq_ok(const mfgm_plan* plan, const mfgm_cq_state* q) {
    if (!plan || !q || !q->dyn) return false;
    const Plan& P = plan->p;
    if (P.wide || P.nlevels < 2) return false;
    if (q->slot && (!q->site_lin || !q->site_sym)) return false;
    return true;
}

}  // namespace

extern "C" {

size_t mfgm_cq_dyn_doubles(const mfgm_plan* plan) {
    if (!plan || plan->p.wide) return 0;
    return packed_elems(plan->p.lv[0], 3 * plan->p.d);
}
size_t mfgm_cq_slot_ints(const mfgm_plan* plan) {
    if (!plan || plan->p.wide) return 0;
    return (size_t)plan->p.lv[0].R * plan->p.lv[0].Lpad;
}

int mfgm_cq_pack(const mfgm_plan* plan, const double* lin, const double* diag, const double* sub, double* dyn, double* range,
                 void* stream) {
    if (!plan || !lin || !diag || !sub || !dyn || !range || plan->p.wide) return 1;
    const Plan& P = plan->p;
    MFGM_DISPATCH_D(P.d, (cq_pack_impl<DD>(P, lin, diag, sub, dyn, range, (hipStream_t)stream)));
}

int mfgm_cq_unpack(const mfgm_plan* plan, const mfgm_cq_state* q, double* lin, double* diag, double* sub, void* stream) {
    if (!plan || !q || !q->dyn || !lin || !diag || !sub || plan->p.wide) return 1;
    const Plan& P = plan->p;
    const CqArgs c = cq_args(q);
    MFGM_DISPATCH_D(P.d, (cq_unpack_impl<DD>(P, c, lin, diag, sub, (hipStream_t)stream)));
}

int mfgm_cq_slots(const mfgm_plan* plan, const long long* node_ids, int n, int* slot, int* dup, void* stream) {
    if (!plan || !node_ids || !slot || !dup || n < 0 || plan->p.wide) return 1;
    const Plan& P = plan->p;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(slot, 0xFF, mfgm_cq_slot_ints(plan) * sizeof(int), st) != hipSuccess) return 3;
    if (hipMemsetAsync(dup, 0, sizeof(int), st) != hipSuccess) return 3;
    if (n > 0) {
        hipLaunchKernelGGL(k_cq_slots, dim3((n + 255) / 256), dim3(256), 0, st, P.lv[0], P.T, node_ids, n, slot, dup);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}

int mfgm_cq_factor(const mfgm_plan* plan, const mfgm_cq_state* q, double* L, double* y, double* logdet, double* quad, void* ws,
                   int* info, void* stream) {
    if (!cq_ok(plan, q) || !L || !y || !ws || !info) return 1;
    const Plan& P = plan->p;
    const CqArgs c = cq_args(q);
    MFGM_DISPATCH_D(P.d, (cq_factor_impl<DD>(P, c, L, y, logdet, quad, (double*)ws, info, (hipStream_t)stream)));
}

int mfgm_cq_factor_pipelined(const mfgm_plan* plan, const mfgm_cq_state* q, double* L, double* y, double* logdet, double* quad, void* ws,
                             int* info, int use_ahead, const mfgm_cq_state* q_next, void* side_stream, void* stream) {
    if (!cq_ok(plan, q) || !L || !y || !ws || !info || plan->p.off_alt == 0) return 1;
    if (side_stream && side_stream == stream) return 1;
    if (q_next && (!cq_ok(plan, q_next) || !q_next->slot)) return 1;
    const Plan& P = plan->p;
    for (int i = 0; i < 2; ++i)
        if (!plan->ev[i] && hipEventCreateWithFlags(&plan->ev[i], hipEventDisableTiming) != hipSuccess) return 3;
    const CqArgs c = cq_args(q);
    CqArgs cn;
    CqPipe pipe;
    pipe.owner = plan; pipe.use_ahead = use_ahead != 0; pipe.side = (hipStream_t)side_stream;
    if (q_next) { cn = cq_args(q_next); pipe.next = &cn; }
    MFGM_DISPATCH_D(P.d, (cq_factor_impl<DD>(P, c, L, y, logdet, quad, (double*)ws, info, (hipStream_t)stream, -1, &pipe)));
}

int mfgm_cq_factor_stage(const mfgm_plan* plan, int stage, const mfgm_cq_state* q, double* L, double* y, void* ws, int* info,
                         void* stream) {
    if (!cq_ok(plan, q) || !L || !y || !ws || !info || stage < 0 || stage > 2) return 1;
    const Plan& P = plan->p;
    const CqArgs c = cq_args(q);
    MFGM_DISPATCH_D(P.d, (cq_factor_impl<DD>(P, c, L, y, nullptr, nullptr, (double*)ws, info, (hipStream_t)stream, stage)));
}

int mfgm_cq_selinv_girsanov(const mfgm_plan* plan, int only_level, const mfgm_cq_state* q, const double* L, const double* y,
                            const mfgm_sde_params* prm, double* dyn_out, void* ws, void* stream) {
    if (!cq_ok(plan, q) || !L || !y || !prm || !dyn_out || !ws || dyn_out == q->dyn || prm->kind != 0) return 1;
    const Plan& P = plan->p;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    CqArgs c = cq_args(q);
    c.dyn_out = dyn_out;
    MFGM_DISPATCH_D(P.d, (cq_selinv_girsanov_impl<DD>(P, c, L, y, pr, (double*)ws, (hipStream_t)stream, only_level)));
}

int mfgm_cq_selinv_kl(const mfgm_plan* plan, int only_level, const mfgm_cq_state* q, const double* L, const double* y,
                      const mfgm_sde_params* prm, double* Sig, double* x, double* kl_part, double* obs_mu, double* obs_cov, void* ws,
                      void* stream) {
    if (!cq_ok(plan, q) || !L || !y || !prm || !kl_part || !ws || prm->kind != 0) return 1;
    if ((obs_mu != nullptr) != (obs_cov != nullptr) || (Sig != nullptr) != (x != nullptr)) return 1;
    const Plan& P = plan->p;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    CqArgs c = cq_args(q);
    c.obs_mu = obs_mu; c.obs_cov = obs_cov;
    MFGM_DISPATCH_D(P.d, (cq_selinv_kl_impl<DD>(P, c, L, y, pr, Sig, x, kl_part, (double*)ws, (hipStream_t)stream, only_level)));
}

int mfgm_kf_sites_loglik(const mfgm_plan* plan, const mfgm_kf_sites* sites, const double* Pd, const double* Ps, double* D, double* r,
                         double* L, double* y, double* t1, double* ldR, double* logdet, double* quad, void* ws, int* info, void* stream) {
    KfArgs k;
    if (!kf_args(plan, sites, 0, k) || !Pd || !D || !r || !L || !y || !t1 || !ldR || !logdet || !quad || !ws || !info) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !Ps) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_DO(P.d, sites->o, (kf_loglik_impl<DD, OO>(P, k, Pd, Ps, D, r, L, y, t1, ldR, logdet, quad, (double*)ws, info, st)));
}

int mfgm_kf_sites_elbo(const mfgm_plan* plan, const mfgm_kf_sites* sites, const double* Pd, const double* Ps, double* D, double* r,
                       double* L, double* y, const double* sumlogchol, double cst, double* terms, double* ll, double* total, void* ws,
                       int* info, void* stream) {
    KfArgs k;
    if (!kf_args(plan, sites, 0, k) || !Pd || !D || !r || !L || !y || !sumlogchol || (!ll && !total) || !ws || !info) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !Ps) return 1;
    if ((size_t)4 * P.B * kElboSplit > std::max((size_t)2 * P.B * 128, (size_t)P.B * (P.d * P.d + P.d))) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_DO(P.d, sites->o, (kf_elbo_impl<DD, OO>(P, k, Pd, Ps, D, r, L, y, sumlogchol, cst, terms, ll, total, (double*)ws, info, st)));
}

int mfgm_kf_sites_predict(const mfgm_plan* plan, const mfgm_kf_sites* sites, const double* Pd, const double* Ps, const double* plin,
                          double* D, double* r, double* L, double* y, double* Sig, double* x, double* Fmu, double* Fvar, void* ws,
                          int* info, void* stream) {
    KfArgs k;
    if (!kf_args(plan, sites, 1, k) || !Pd || !D || !r || !L || !y || !Sig || !x || !Fmu || !Fvar || !ws || !info) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !Ps) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_DO(P.d, sites->o, (kf_predict_impl<DD, OO>(P, k, Pd, Ps, plin, D, r, L, y, Sig, x, Fmu, Fvar, (double*)ws, info, st)));
}

int mfgm_kf_sites_predict_factored(const mfgm_plan* plan, const mfgm_kf_sites* sites, const double* Ps, const double* L, const double* y,
                                   double* Sig, double* x, double* Fmu, double* Fvar, void* ws, void* stream) {
    KfArgs k;
    if (!kf_args(plan, sites, 1, k) || !L || !y || !Sig || !x || !Fmu || !Fvar || !ws) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !Ps) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_DO(P.d, sites->o, (kf_predict_factored_impl<DD, OO>(P, k, Ps, L, y, Sig, x, Fmu, Fvar, (double*)ws, st)));
}

int mfgm_mvn_ve_compact(int B, int n_per, int d, const double* mu, const double* cov, const double* y, const double* Sinv, double cst,
                        double* ve, void* stream) {
    if (B < 1 || n_per < 0 || !mu || !cov || !y || !Sinv || !ve) return 1;
    MFGM_DISPATCH_D(d, (mvn_ve_compact_impl<DD>(B, n_per, mu, cov, y, Sinv, cst, ve, (hipStream_t)stream)));
}


int mfgm_packed_factor(const mfgm_plan* plan, const double* D, const double* S, const double* r, double aD, double aS,
                       double aR, double* L, double* G, double* y, double* logdet, double* quad, void* ws, int* info,
                       void* stream) {
    if (!plan || !D || !L || !info) return 1;
    const Plan& P = plan->p;
    if (!G && P.wide) return 1;           // the lane-per-segment kernels may skip the L_{t+1,t} output (see mfgm_packed_selinv_mom_s)
    if (P.T > 1 && !S) return 1;
    if ((r != nullptr) != (y != nullptr)) return 1;
    if (!ws && P.ws_doubles > 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (P.wide) return wide_factor(P, D, S, r, aD, aS, aR, L, G, y, logdet, quad, (double*)ws, info, st);
    MFGM_DISPATCH_D(P.d, (factor_impl<DD>(P, D, S, r, aD, aS, aR, L, G, y, logdet, quad, (double*)ws, info, st)));
}

int mfgm_packed_selinv(const mfgm_plan* plan, const double* L, const double* G, const double* y, double* Sig,
                       double* Sub, double* x, void* ws, void* stream) {
    if (!plan || !L || !G || !Sig) return 1;
    const Plan& P = plan->p;
    if ((y != nullptr) != (x != nullptr)) return 1;
    if (!ws && P.ws_doubles > 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (P.wide) return wide_selinv(P, L, G, y, Sig, Sub, x, (double*)ws, st);
    MFGM_DISPATCH_D(P.d, (selinv_impl<DD>(P, L, G, y, Sig, Sub, x, (double*)ws, st)));
}


int mfgm_packed_factor_form(const mfgm_plan* plan, int form, const double* D, const double* S, const double* r, double aD, double aS,
                            double aR, double* L, double* G, double* y, double* logdet, double* quad, void* ws, int* info,
                            void* stream) {
    if (form == 0) return mfgm_packed_factor(plan, D, S, r, aD, aS, aR, L, G, y, logdet, quad, ws, info, stream);
    if (!plan || !D || !L || !G || !info || form != 1) return 1;
    const Plan& P = plan->p;
    if (!P.wide) return 1;
    if (P.T > 1 && !S) return 1;
    if ((r != nullptr) != (y != nullptr)) return 1;
    if (!ws && P.ws_doubles > 0) return 1;
    return wide_factor(P, D, S, r, aD, aS, aR, L, G, y, logdet, quad, (double*)ws, info, (hipStream_t)stream, -1, 1);
}

int mfgm_packed_selinv_form(const mfgm_plan* plan, int form, const double* L, const double* G, const double* y, double* Sig,
                            double* Sub, double* x, void* ws, void* stream) {
    if (form == 0) return mfgm_packed_selinv(plan, L, G, y, Sig, Sub, x, ws, stream);
    if (!plan || !L || !G || !Sig || form != 1) return 1;
    const Plan& P = plan->p;
    if (!P.wide) return 1;
    if ((y != nullptr) != (x != nullptr)) return 1;
    if (!ws && P.ws_doubles > 0) return 1;
    return wide_selinv(P, L, G, y, Sig, Sub, x, (double*)ws, (hipStream_t)stream, 1);
}


// Profiling / roofline entry points: launch exactly ONE kernel of a sweep (stage 0 = reduce, 1 = forward; level 0 is
// the finest).  The coarser levels must already be in `ws` from a full mfgm_packed_factor / mfgm_packed_selinv call
// with the same arguments; outputs are overwritten with identical values.
int mfgm_packed_factor_stage(const mfgm_plan* plan, int stage, int level, const double* D, const double* S, const double* r,
                             double aD, double aS, double aR, double* L, double* G, double* y, void* ws, int* info,
                             void* stream) {
    if (!plan || !D || !L || !info || stage < 0 || stage > 1) return 1;        // G may be NULL as in mfgm_packed_factor
    const Plan& P = plan->p;
    if (P.wide) return 1;
    if (level < 0 || level >= P.nlevels || (stage == 0 && level >= P.nlevels - 1)) return 1;
    if ((r != nullptr) != (y != nullptr)) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (factor_impl<DD>(P, D, S, r, aD, aS, aR, L, G, y, nullptr, nullptr, (double*)ws, info, st, stage, level)));
}

int mfgm_wide_stage(const mfgm_plan* plan, int form, int which, const double* D, const double* S, const double* r, double aD, double aS,
                    double aR, double* L, double* G, double* y, double* Sig, double* Sub, double* x, const double* site1,
                    const double* site2, void* ws, int* info, void* stream) {
    if (!plan || !L || !G || !info || !ws) return 1;
    const Plan& P = plan->p;
    if (!P.wide) return 1;
    if (which < 2 && (!D || !S || ((r != nullptr) || (site1 != nullptr)) != (y != nullptr))) return 1;
    if (which == 2 && (!Sig || (y != nullptr) != (x != nullptr))) return 1;
    return wide_stage(P, form, which, D, S, r, aD, aS, aR, L, G, y, Sig, Sub, x, (double*)ws, info, (hipStream_t)stream, site1, site2);
}

int mfgm_sparse_factor(const mfgm_plan* plan, const double* nat1, const double* nat2, const double* plin, const double* pdiag,
                       const double* psub, double* L, double* G, double* y, double* logdet, double* quad, void* ws, int* info,
                       void* stream) {
    if (!plan || !nat1 || !nat2 || !pdiag || !psub || !L || !G || !y || !info || !ws) return 1;
    const Plan& P = plan->p;
    if (!P.wide || P.B != 1 || P.shard_level > 0) return 1;
    return wide_factor(P, pdiag, psub, plin, -2.0, -1.0, 1.0, L, G, y, logdet, quad, (double*)ws, info, (hipStream_t)stream, -1, 1, nat1,
                       nat2);
}

int mfgm_sparse_factor_phase(const mfgm_plan* plan, int phase, const double* nat1, const double* nat2, const double* plin,
                             const double* pdiag, const double* psub, double* L, double* G, double* y, double* logdet, double* quad, void* ws,
                             int* info, void* stream) {
    if (!plan || !nat1 || !nat2 || !pdiag || !psub || !L || !G || !y || !info || !ws || (phase != 0 && phase != 1)) return 1;
    const Plan& P = plan->p;
    if (!P.wide || P.B != 1 || P.shard_level < 1) return 1;
    return wide_factor(P, pdiag, psub, plin, -2.0, -1.0, 1.0, L, G, y, logdet, quad, (double*)ws, info, (hipStream_t)stream, phase, 1, nat1,
                       nat2);
}

int mfgm_sparse_factor_q(const mfgm_plan* plan, int phase, const double* nat1, const double* nat2q, const double* plin, const double* pdiag,
                         const double* psub, double* L, double* G, double* y, double* logdet, double* quad, void* ws, int* info,
                         void* stream) {
    if (!plan || !nat1 || !nat2q || !pdiag || !psub || !L || !G || !y || !info || !ws || phase < -1 || phase > 1) return 1;
    const Plan& P = plan->p;
    if (!P.wide || P.B != 1 || (phase < 0) != (P.shard_level < 1)) return 1;
    return wide_factor(P, pdiag, psub, plin, -2.0, -1.0, 1.0, L, G, y, logdet, quad, (double*)ws, info, (hipStream_t)stream, phase, 1, nat1,
                       nat2q, 1);
}

int mfgm_wide_stage_q(const mfgm_plan* plan, int which, const double* D, const double* S, const double* r, double aD, double aS, double aR,
                      double* L, double* G, double* y, const double* site1, const double* site2q, void* ws, int* info, void* stream) {
    if (!plan || !L || !G || !y || !info || !ws || !D || !S || !site1 || !site2q || which < 0 || which > 1) return 1;
    const Plan& P = plan->p;
    if (!P.wide) return 1;
    return wide_stage(P, 1, which, D, S, r, aD, aS, aR, L, G, y, nullptr, nullptr, nullptr, (double*)ws, info, (hipStream_t)stream, site1,
                      site2q, 1);
}

int mfgm_plan_shard_left_marginal(const mfgm_plan* plan, double* Sig, double* x, const void* ws, void* stream) {
    if (!plan || !Sig || !ws) return 1;
    const Plan& P = plan->p;
    if (!P.wide || P.shard_level < 1) return 1;
    if (P.own_lo[0] == 0) return 0;                       // the first process has no left neighbour
    const int X = P.shard_level, q = P.own_lo[X - 1] - 1, nX = P.lv[X].n, EF = P.d * P.d;
    const size_t t = (size_t)P.own_lo[0] * P.lv[0].R - 1;
    const double* w = (const double*)ws;
    for (int b = 0; b < P.B; ++b) {
        if (hipMemcpyAsync(Sig + ((size_t)b * P.T + t) * EF, w + P.off_Sig[X] + ((size_t)b * nX + q) * EF, EF * sizeof(double),
                           hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return 3;
        if (x && hipMemcpyAsync(x + ((size_t)b * P.T + t) * P.d, w + P.off_mu[X] + ((size_t)b * nX + q) * P.d, P.d * sizeof(double),
                                hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return 3;
    }
    return 0;
}

int mfgm_packed_selinv_level(const mfgm_plan* plan, int level, const double* L, const double* G, const double* y, double* Sig,
                             double* Sub, double* x, void* ws, void* stream) {
    if (!plan || !L || !G || !Sig) return 1;
    const Plan& P = plan->p;
    if (level < 0 || level >= P.nlevels) return 1;
    if ((y != nullptr) != (x != nullptr)) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (selinv_impl<DD>(P, L, G, y, Sig, Sub, x, (double*)ws, st, level)));
}

int mfgm_packed_selinv_mom_s(const mfgm_plan* plan, int only_level, const double* L, const double* S, double aS, const double* y,
                             double* Sig, double* x, double* mom, void* ws, void* stream) {
    if (!plan || !L || !S || !Sig || !y || !x) return 1;          // mom may be NULL: marginals only
    const Plan& P = plan->p;
    if (P.wide || only_level >= P.nlevels) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (selinv_impl<DD>(P, L, nullptr, y, Sig, nullptr, x, (double*)ws, st, only_level, mom, S, aS)));
}

int mfgm_packed_selinv_girsanov(const mfgm_plan* plan, int only_level, const double* L, const double* S, double aS, const double* y,
                                const mfgm_sde_params* prm, const double* q1, const double* qd, double* n1, double* nd, double* ns,
                                void* ws, void* stream) {
    if (!plan || !L || !S || !y || !prm || !q1 || !qd || !n1 || !nd || !ns || !ws) return 1;
    if (n1 == q1 || nd == qd || ns == S) return 1;            // the update is out of place (see mfgm_girsanov.h)
    const Plan& P = plan->p;
    if (P.wide || P.nlevels < 2 || only_level >= P.nlevels || prm->kind != 0) return 1;
    static_assert(sizeof(mfgm_sde_params) == sizeof(mfgm::SdeParams), "public and internal SDE parameter structs must match");
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    GirsanovArgs g{q1, qd, n1, nd, ns, nullptr};
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (selinv_girsanov_impl<DD>(P, L, S, aS, y, pr, g, (double*)ws, st, only_level)));
}

int mfgm_packed_selinv_kl(const mfgm_plan* plan, int only_level, const double* L, const double* S, double aS, const double* y,
                          const mfgm_sde_params* prm, double* Sig, double* x, double* kl_part, void* ws, void* stream) {
    if (!plan || !L || !S || !y || !prm || !Sig || !x || !kl_part || !ws) return 1;
    const Plan& P = plan->p;
    if (P.wide || P.nlevels < 2 || only_level >= P.nlevels || prm->kind != 0) return 1;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (selinv_kl_impl<DD>(P, L, S, aS, y, pr, Sig, x, kl_part, (double*)ws, st, only_level)));
}

int mfgm_packed_selinv_mom(const mfgm_plan* plan, int only_level, const double* L, const double* G, const double* y, double* Sig,
                           double* Sub, double* x, double* mom, void* ws, void* stream) {
    if (!plan || !L || !G || !Sig || !y || !x || !mom) return 1;
    const Plan& P = plan->p;
    if (only_level >= P.nlevels) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (selinv_impl<DD>(P, L, G, y, Sig, Sub, x, (double*)ws, st, only_level, mom)));
}


int mfgm_packed_factor_phase(const mfgm_plan* plan, int phase, const double* D, const double* S, const double* r, double aD,
                             double aS, double aR, double* L, double* G, double* y, double* logdet, double* quad, void* ws,
                             int* info, void* stream) {
    if (!plan || !D || !L || !G || !info || !ws || (phase != 0 && phase != 1)) return 1;
    const Plan& P = plan->p;
    if (!P.wide || P.nlevels < 2 || !S) return 1;
    if ((r != nullptr) != (y != nullptr)) return 1;
    return wide_factor(P, D, S, r, aD, aS, aR, L, G, y, logdet, quad, (double*)ws, info, (hipStream_t)stream, phase);
}

int mfgm_packed_factor_phase_form(const mfgm_plan* plan, int form, int phase, const double* D, const double* S, const double* r, double aD,
                                  double aS, double aR, double* L, double* G, double* y, double* logdet, double* quad, void* ws,
                                  int* info, void* stream) {
    if (!plan || !D || !L || !G || !info || !ws || (phase != 0 && phase != 1) || (form != 0 && form != 1)) return 1;
    const Plan& P = plan->p;
    if (!P.wide || P.nlevels < 2 || !S) return 1;
    if ((r != nullptr) != (y != nullptr)) return 1;
    return wide_factor(P, D, S, r, aD, aS, aR, L, G, y, logdet, quad, (double*)ws, info, (hipStream_t)stream, phase, form);
}

}  // extern "C"
