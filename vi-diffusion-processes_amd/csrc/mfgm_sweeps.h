// The three sweep kernels of the partitioned block-tri-diagonal SPD solver (lane = one chain segment).
//
//   reduce   : eliminates the interior of every segment, emitting the separator (reduced) system
//   forward  : natural-order block Cholesky  L_tt, L_{t+1,t}, y = L^{-1} r   (+ log|L|, |y|^2 partials)
//   backward : Takahashi selected inverse   S_tt, S_{t+1,t}  and  x = L^{-T} y
//
// They replace, fused and block-specialised, what the reference gets from banded_matrices'
// cholesky_band / solve_triang_mat / inverse_from_cholesky_band (block_tri_diag.py:330-331,350,440;
// ssm_gaussian_transformations.py:443-444).  All arrays use the packed layout of mfgm_layout.h.
#pragma once
#include "mfgm_layout.h"
#include "mfgm_math.h"

namespace mfgm {

// ---- packed-layout element access --------------------------------------------------------------
// Element (lane, step s, e) of an array with E doubles per node and R steps per segment lives at
//   (((lane/64)*R + s)*E + e)*64 + lane%64
// For the wave's own lanes the tile (= blockIdx.x) is uniform, so the node pointer is scalar arithmetic and every
// element is reached with an immediate offset e*512 from it (no per-element vector address math).
struct LaneRef {
    int tile, l;
    int nt = 0;      // cache policy of the level-0 arrays this lane streams (LevelDesc::nt; a compile-time constant where it is not 0)
    MFGM_DEV static LaneRef of(int lane) { return LaneRef{lane >> 6, lane & 63}; }
};
// NM marks an array of a level above the finest one (plan workspace).  Those levels use the SAME lane-interleaved layout as level 0,
// with the level's own lanes and R:  element e of (lane, step s) at (((lane / 64) * R + s) * E + e) * 64 + lane % 64  -- a wavefront of the
// level reads 512 contiguous bytes per (step, element), and the 64 separators that consecutive lanes of the level below hand up land in
// R runs of 64 / R lanes each (whole cache lines for R <= 8).  Round 2 and most of round 3 kept these levels node-major (node = lane * R + s, element e at
// ((node / 64) * E + e) * 64 + node % 64: the hand-over is one 512-byte run, but a wavefront of the level itself then reads R-strided
// doubles, 32 cache lines per load instead of 8).  Per-phase cycle stamps of the fused coarse kernel showed where that goes: the level
// with 256 segments per chain -- four wavefronts sharing one CU's texture-address unit -- took 27 000 cycles per block step against
// 11 400 for the level with one wavefront.  Same box, headline: k_coarse_factor 0.143 -> 0.090 ms, k_coarse_backward 0.062 -> 0.047,
// level-0 kernels unchanged, step 2.96 -> 2.83 ms.  MFGM_COARSE_NODE_MAJOR=1 at compile time brings the node-major layout back.
// offset of element 0 of (lane, step s) of a level >= 1 array with E doubles per node (the elements follow at a stride of 64 doubles)
#ifndef MFGM_COARSE_NODE_MAJOR
#define MFGM_COARSE_NODE_MAJOR 0
#endif
template <int E>
MFGM_DEV size_t coarse_off(int lane, int R, int s) {
#if MFGM_COARSE_NODE_MAJOR
    const size_t node = (size_t)lane * R + s;
    return ((node >> 6) * E) * 64 + (node & 63);
#else
    return ((size_t)(lane >> 6) * R + s) * (size_t)(E * 64) + (lane & 63);
#endif
}
template <int E, bool NM = false>
MFGM_DEV void ld_node(const double* __restrict__ base, int R, int s, LaneRef w, double (&out)[E]) {
    if constexpr (NM) {
        const double* p = base + coarse_off<E>(w.tile * 64 + w.l, R, s);
#pragma unroll
        for (int e = 0; e < E; ++e) out[e] = p[e * 64];
    } else {
        // Level-0 arrays of the large models (GBs, written by one pass and read by the next after everything else has gone through the
        // caches) are streamed with the non-temporal policy; models that fit the caches keep the default one (config 2 is 3.5 % slower
        // without it).  Plan creation decides (LevelDesc::nt, mfgm_plan_create), the CVI-DP sweeps carry the value as a template
        // parameter: behind a run-time branch the compiler merges the two copies of a load and drops the hint (no `nt` in the ISA).
        const double* p = base + ((size_t)w.tile * R + s) * (size_t)(E * 64);
        if (w.nt >= 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) out[e] = __builtin_nontemporal_load(p + e * 64 + w.l);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) out[e] = p[e * 64 + w.l];
        }
    }
}
template <int E, bool NM = false>
MFGM_DEV void st_node(double* __restrict__ base, int R, int s, LaneRef w, const double (&v)[E]) {
    if constexpr (NM) {
        double* p = base + coarse_off<E>(w.tile * 64 + w.l, R, s);
#pragma unroll
        for (int e = 0; e < E; ++e) p[e * 64] = v[e];
    } else {
        double* p = base + ((size_t)w.tile * R + s) * (size_t)(E * 64);
        if (w.nt >= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e) __builtin_nontemporal_store(v[e], p + e * 64 + w.l);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) p[e * 64 + w.l] = v[e];
        }
    }
}
template <int E, bool NM = false>
MFGM_DEV void st_node_zero(double* __restrict__ base, int R, int s, LaneRef w) {
    if constexpr (NM) {
        double* p = base + coarse_off<E>(w.tile * 64 + w.l, R, s);
#pragma unroll
        for (int e = 0; e < E; ++e) p[e * 64] = 0.0;
    } else {
        double* p = base + ((size_t)w.tile * R + s) * (size_t)(E * 64);
#pragma unroll
        for (int e = 0; e < E; ++e) p[e * 64 + w.l] = 0.0;
    }
}

struct SweepArgs {
    LevelDesc lv;          // this level
    LevelDesc up;          // next (coarser) level, valid when the level has more than one segment
    // inputs of this level
    const double* Dg;      // tri-packed symmetric diagonal blocks
    const double* Sg;      // full sub-diagonal blocks, S at node t couples t -> t+1
    const double* rg;      // right-hand side (may be null when !HAS_RHS)
    const double* Dcorr;   // level >= 1: subtract (spike Gram term of the segment to the right)
    const double* rcorr;
    double aD, aS, aR;     // scales applied on load (e.g. -2, -1, 1 turn natural parameters into a precision)
    // factor outputs of this level
    double* Lg;            // tri-packed L_tt
    double* Gg;            // full L_{t+1,t} stored at node t
    double* yg;            // L^{-1} r
    double* part;          // [2*Lpad] per-lane partial log|L| and |y|^2 (may be null)
    // selected-inverse outputs of this level
    double* Sigg;          // tri-packed S_tt
    double* Subg;          // full S_{t+1,t} at node t (may be null)
    double* mug;           // L^{-T} y
    double* momg;          // optional [3d per node]: (mu, diag Sigma_tt, diag Sigma_{t+1,t}) for the local CVI-DP kernels
    // coarser level arrays
    double* uDhat; double* uRsub; double* uS; double* urhat; double* urho;   // written by reduce
    const double* uL; const double* uy;                                         // read by forward
    const double* uSig; const double* umu;                                      // read by backward
    int* info;             // set non-zero when a pivot block is not positive definite
};

// ---- reduce -------------------------------------------------------------------------------------
// The bodies are device functions of (lane, me): the per-level kernels call them with the wave's uniform tile (me = {blockIdx.x,
// threadIdx.x}: scalar node pointers), the fused coarse-level kernels (k_coarse_*) with arbitrary lanes of one chain.
// Keep a batch of loads a batch: an empty asm that "modifies" every value makes the loads complete where the batch was issued.  Without
// it the compiler, under register pressure, sinks each load to its point of use and waits for it there -- on the latency-bound coarse
// levels that turned one memory round trip per step into ~25 dependent ones (global_load; s_waitcnt vmcnt(0); use; ... in the ISA).
template <int N>
MFGM_DEV void pin_loaded(double (&v)[N]) {
#pragma unroll
    for (int e = 0; e < N; ++e) asm volatile("" : "+v"(v[e]));
}

template <int D, bool HAS_RHS, bool HAS_CORR>
MFGM_DEV void reduce_body(const SweepArgs& a, const int lane, const LaneRef me) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int P = a.lv.P, R = a.lv.R, Lp = a.lv.Lpad;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, a.lv.n - p * R);
    int bad = 0;

    double F[ET], W[EF], h[D], Racc[ET], rho[D];
    ld_node<ET, HAS_CORR>(a.Dg, R, 0, me, F);
#pragma unroll
    for (int e = 0; e < ET; ++e) F[e] *= a.aD;
    if (HAS_CORR) {
        double c[ET];
        ld_node<ET, HAS_CORR>(a.Dcorr, R, 0, me, c);
#pragma unroll
        for (int e = 0; e < ET; ++e) F[e] -= c[e];
    }
    if (p > 0) {
        ld_node<EF, HAS_CORR>(a.Sg, R, R - 1, LaneRef::of(lane - 1), W);
#pragma unroll
        for (int e = 0; e < EF; ++e) W[e] *= a.aS;
    } else {
#pragma unroll
        for (int e = 0; e < EF; ++e) W[e] = 0.0;
    }
    if (HAS_RHS) {
        ld_node<D, HAS_CORR>(a.rg, R, 0, me, h);
#pragma unroll
        for (int e = 0; e < D; ++e) h[e] *= a.aR;
        if (HAS_CORR) {
            double c[D];
            ld_node<D, HAS_CORR>(a.rcorr, R, 0, me, c);
#pragma unroll
            for (int e = 0; e < D; ++e) h[e] -= c[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < D; ++e) h[e] = 0.0;
    }
#pragma unroll
    for (int e = 0; e < ET; ++e) Racc[e] = 0.0;
#pragma unroll
    for (int e = 0; e < D; ++e) rho[e] = 0.0;

    // Cross-iteration register prefetch: the raw blocks of step s+1 are requested at the top of step s and consumed one
    // iteration later; scales / corrections are applied at the point of use, never at the load.  (Level >= 1 keeps the
    // simpler same-step loads: the correction arrays would not fit the register file next to a second buffer.)
    double Gn[EF], Dn[ET], rn[D];
    auto load_step = [&](int s, double (&Go)[EF], double (&Do)[ET], double (&ro)[D]) {
        ld_node<EF, HAS_CORR>(a.Sg, R, s, me, Go);
        ld_node<ET, HAS_CORR>(a.Dg, R, s + 1, me, Do);
        if (HAS_RHS) ld_node<D, HAS_CORR>(a.rg, R, s + 1, me, ro);
    };
    if (!HAS_CORR && len > 1) load_step(0, Gn, Dn, rn);
    for (int s = 0; s < R - 1; ++s) {
        if (s < len - 1) {
            double G[EF], Dc_[ET], rc_[D], Dcur[ET], rcur[D];
            if (HAS_CORR) {
                load_step(s, G, Dcur, rcur);
                ld_node<ET, HAS_CORR>(a.Dcorr, R, s + 1, me, Dc_);
                if (HAS_RHS) ld_node<D, HAS_CORR>(a.rcorr, R, s + 1, me, rc_);
                pin_loaded(G);
                pin_loaded(Dcur);
                pin_loaded(Dc_);
                if (HAS_RHS) { pin_loaded(rcur); pin_loaded(rc_); }
            } else {
#pragma unroll
                for (int e = 0; e < EF; ++e) G[e] = Gn[e];
#pragma unroll
                for (int e = 0; e < ET; ++e) Dcur[e] = Dn[e];
#pragma unroll
                for (int e = 0; e < D; ++e) rcur[e] = HAS_RHS ? rn[e] : 0.0;
                if (s + 1 < len - 1) load_step(s + 1, Gn, Dn, rn);
            }
            // eliminate interior node s
            double invd[D];
            chol_inplace<D>(F, invd, bad);
            trsm_left_lower<D>(F, invd, W);        // W := L^{-1} W   (spike towards the left separator)
            syrk_t_acc<D>(W, Racc);                // R += W^T W
            if (HAS_RHS) {
                trsv_lower<D>(F, invd, h);         // y := L^{-1} h
                double t[D];
                gemv_t<D>(W, h, t);
#pragma unroll
                for (int e = 0; e < D; ++e) rho[e] += t[e];
            }
#pragma unroll
            for (int e = 0; e < EF; ++e) G[e] *= a.aS;
            trsm_right_lower_t<D>(F, invd, G);     // G := S L^{-T}
            // Schur complement onto node s+1
            syrk_set<D>(G, F);
#pragma unroll
            for (int e = 0; e < ET; ++e) F[e] = __builtin_fma(a.aD, Dcur[e], -F[e]) - (HAS_CORR ? Dc_[e] : 0.0);
#pragma unroll
            for (int c = 0; c < D; ++c) {          // W := -G W, column by column in place
                double col[D];
#pragma unroll
                for (int k = 0; k < D; ++k) col[k] = W[k * D + c];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = __builtin_fma(G[i * D + k], col[k], t);
                    W[i * D + c] = -t;
                }
            }
            if (HAS_RHS) {
                double t[D];
                gemv<D>(G, h, t);
#pragma unroll
                for (int e = 0; e < D; ++e) h[e] = __builtin_fma(a.aR, rcur[e], -t[e]) - (HAS_CORR ? rc_[e] : 0.0);
            }
        }
    }
    // separator of this segment is node q = p of the coarser level
    const int uP = a.up.P, uR = a.up.R;
    {
        const int q = p, ul = b * uP + q / uR, us = q % uR;
        st_node<ET, true>(a.uDhat, uR, us, LaneRef::of(ul), F);
        st_node<D, true>(a.urhat, uR, us, LaneRef::of(ul), h);
        if (p == P - 1) {
            st_node_zero<ET, true>(a.uRsub, uR, us, LaneRef::of(ul));
            st_node_zero<D, true>(a.urho, uR, us, LaneRef::of(ul));
            st_node_zero<EF, true>(a.uS, uR, us, LaneRef::of(ul));
        }
    }
    if (p > 0) {
        const int q = p - 1, ul = b * uP + q / uR, us = q % uR;
        st_node<EF, true>(a.uS, uR, us, LaneRef::of(ul), W);      // couples separator p-1 -> p
        st_node<ET, true>(a.uRsub, uR, us, LaneRef::of(ul), Racc);
        st_node<D, true>(a.urho, uR, us, LaneRef::of(ul), rho);
    }
    if (bad) flag_not_pd(a.info, a.lv.level, lane);
}
template <int D, bool HAS_RHS, bool HAS_CORR>
static __global__ __launch_bounds__(64) void k_reduce(SweepArgs a) {
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    reduce_body<D, HAS_RHS, HAS_CORR>(a, lane, LaneRef{(int)blockIdx.x, (int)threadIdx.x});
}

// ---- forward ------------------------------------------------------------------------------------
template <int D, bool HAS_RHS, bool HAS_CORR, bool HAS_UP>
MFGM_DEV void forward_body(const SweepArgs& a, const int lane, const LaneRef me) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int P = a.lv.P, R = a.lv.R, Lp = a.lv.Lpad, n = a.lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    int bad = 0;

    double C[ET], c[D];
#pragma unroll
    for (int e = 0; e < ET; ++e) C[e] = 0.0;
#pragma unroll
    for (int e = 0; e < D; ++e) c[e] = 0.0;

    if (HAS_UP && p > 0) {
        // natural-order Cholesky state at the separator to the left:
        //   F_a = Ltil Ltil^T + R_p ,  h_a = Ltil ytil + rho_p
        const int uP = a.up.P, uR = a.up.R;
        const int q = p - 1, ul = b * uP + q / uR, us = q % uR;
        double Lt[ET], Fa[ET], ha[D], invd[D];
        ld_node<ET, true>(a.uL, uR, us, LaneRef::of(ul), Lt);
        ld_node<ET, true>(a.uRsub, uR, us, LaneRef::of(ul), Fa);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double t = Fa[tix(i, j)];
#pragma unroll
                for (int k = 0; k <= j; ++k) t = __builtin_fma(Lt[tix(i, k)], Lt[tix(j, k)], t);
                Fa[tix(i, j)] = t;
            }
        if (HAS_RHS) {
            double yt[D];
            ld_node<D, true>(a.uy, uR, us, LaneRef::of(ul), yt);
            ld_node<D, true>(a.urho, uR, us, LaneRef::of(ul), ha);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double t = ha[i];
#pragma unroll
                for (int k = 0; k <= i; ++k) t = __builtin_fma(Lt[tix(i, k)], yt[k], t);
                ha[i] = t;
            }
        } else {
#pragma unroll
            for (int e = 0; e < D; ++e) ha[e] = 0.0;
        }
        chol_inplace<D>(Fa, invd, bad);
        trsv_lower<D>(Fa, invd, ha);
        double Ga[EF];
        ld_node<EF, HAS_CORR>(a.Sg, R, R - 1, LaneRef::of(lane - 1), Ga);
#pragma unroll
        for (int e = 0; e < EF; ++e) Ga[e] *= a.aS;
        trsm_right_lower_t<D>(Fa, invd, Ga);
        syrk_set<D>(Ga, C);
        gemv<D>(Ga, ha, c);
    }
    // keep the first step's loads below the boundary arithmetic (the two together overflow the register file)
    __builtin_amdgcn_sched_barrier(0);

    LogAcc la;
    la.init();
    double quad = 0.0;

    // cross-iteration register prefetch: the raw blocks of step s+1 are in flight while step s is factored;
    // scales / corrections are applied at the point of use so no wait is forced at the load.
    double Fn[ET], Gn[EF], rn[D], Fc[HAS_CORR ? ET : 1], rc[HAS_CORR ? D : 1];
    auto load_step = [&](int s) {
        ld_node<ET, HAS_CORR>(a.Dg, R, s, me, Fn);
        if constexpr (HAS_CORR) ld_node<ET, HAS_CORR>(a.Dcorr, R, s, me, reinterpret_cast<double(&)[ET]>(Fc));
        if (p * R + s + 1 < n) {
            ld_node<EF, HAS_CORR>(a.Sg, R, s, me, Gn);
        } else {
#pragma unroll
            for (int e = 0; e < EF; ++e) Gn[e] = 0.0;
        }
        if (HAS_RHS) {
            ld_node<D, HAS_CORR>(a.rg, R, s, me, rn);
            if constexpr (HAS_CORR) ld_node<D, HAS_CORR>(a.rcorr, R, s, me, reinterpret_cast<double(&)[D]>(rc));
        }
    };
    load_step(0);
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            double F[ET], G[EF], h[D];
#pragma unroll
            for (int e = 0; e < ET; ++e) F[e] = __builtin_fma(a.aD, Fn[e], -C[e]) - (HAS_CORR ? Fc[e] : 0.0);
#pragma unroll
            for (int e = 0; e < EF; ++e) G[e] = Gn[e] * a.aS;
#pragma unroll
            for (int e = 0; e < D; ++e)
                h[e] = HAS_RHS ? (__builtin_fma(a.aR, rn[e], -c[e]) - (HAS_CORR ? rc[e] : 0.0)) : 0.0;
            if (s + 1 < len) load_step(s + 1);
            double invd[D];
            chol_inplace<D>(F, invd, bad);
            if (HAS_RHS) trsv_lower<D>(F, invd, h);
            trsm_right_lower_t<D>(F, invd, G);
            st_node<ET, HAS_CORR>(a.Lg, R, s, me, F);
            if (a.Gg) st_node<EF, HAS_CORR>(a.Gg, R, s, me, G);     // callers that rebuild H from S in the backward pass skip this store
            if (HAS_RHS) st_node<D, HAS_CORR>(a.yg, R, s, me, h);
            syrk_set<D>(G, C);
            if (HAS_RHS) gemv<D>(G, h, c);
#pragma unroll
            for (int j = 0; j < D; ++j) la.mul(F[tix(j, j)]);
            la.renorm();
#pragma unroll
            for (int j = 0; j < D; ++j) quad = __builtin_fma(h[j], h[j], quad);
        }
    }
    if (a.part) {
        a.part[lane] = la.value();
        a.part[Lp + lane] = quad;
    }
    if (bad) flag_not_pd(a.info, a.lv.level, lane);
}
template <int D, bool HAS_RHS, bool HAS_CORR, bool HAS_UP>
static __global__ __launch_bounds__(64) void k_forward(SweepArgs a) {
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    forward_body<D, HAS_RHS, HAS_CORR, HAS_UP>(a, lane, LaneRef{(int)blockIdx.x, (int)threadIdx.x});
}

// ---- backward -----------------------------------------------------------------------------------
// USE_S: the forward pass did not store L_{t+1,t} = S L^{-T}; with P = L^{-T} L^{-1} the same quantities follow from the input
// sub-diagonal block S (scaled by aS):  H = L_{t+1,t} L^{-1} = S P,  L_{t+1,t}^T x = L^{-1} (S^T x).  Same bytes read here (S for
// L_{t+1,t}), d^2 doubles per node fewer written by the forward pass.
template <int D, bool HAS_RHS, bool HAS_UP, bool WANT_SUB, bool WANT_MOM, bool USE_S = false, bool CL = false>
MFGM_DEV void backward_body(const SweepArgs& a, const int lane, const LaneRef me) {
    const double* __restrict__ Gsrc = USE_S ? a.Sg : a.Gg;
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int P = a.lv.P, R = a.lv.R, Lp = a.lv.Lpad, n = a.lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    const int se = len - 1;

    double Sn[ET], xn[D];
    if (HAS_UP) {
        const int uP = a.up.P, uR = a.up.R;
        const int q = p, ul = b * uP + q / uR, us = q % uR;
        ld_node<ET, true>(a.uSig, uR, us, LaneRef::of(ul), Sn);
        if (HAS_RHS) ld_node<D, true>(a.umu, uR, us, LaneRef::of(ul), xn);
    } else {
        double Lt[ET], invd[D], X[ET];
        ld_node<ET, CL>(a.Lg, R, se, me, Lt);
#pragma unroll
        for (int j = 0; j < D; ++j) invd[j] = rcp_nr(Lt[tix(j, j)]);
        tri_inverse<D>(Lt, invd, X);
        tri_t_tri<D>(X, Sn);
        if (HAS_RHS) {
            ld_node<D, CL>(a.yg, R, se, me, xn);
            trsv_lower_t<D>(Lt, invd, xn);
        }
    }
    if (!HAS_RHS) {
#pragma unroll
        for (int e = 0; e < D; ++e) xn[e] = 0.0;
    }
    st_node<ET, CL>(a.Sigg, R, se, me, Sn);
    if (HAS_RHS) st_node<D, CL>(a.mug, R, se, me, xn);
    if (WANT_SUB && (p * R + se == n - 1)) st_node_zero<EF, CL>(a.Subg, R, se, me);
    if (WANT_MOM) {
        // (mu, diag Sigma) of the separator now; diag Sigma_{t+1,t} of the separator is written by the lane on its right
        double mm[3 * D];
#pragma unroll
        for (int i = 0; i < D; ++i) { mm[i] = xn[i]; mm[D + i] = Sn[tix(i, i)]; mm[2 * D + i] = 0.0; }
        double* pm = a.momg + ((size_t)me.tile * R + se) * (size_t)(3 * D * 64);
#pragma unroll
        for (int e = 0; e < 2 * D; ++e) pm[e * 64 + me.l] = mm[e];
        if (p * R + se == n - 1) {
#pragma unroll
            for (int e = 2 * D; e < 3 * D; ++e) pm[e * 64 + me.l] = 0.0;
        }
    }

    double Ln[ET], Gn[EF], yn[D];
    if (len > 1) {
        ld_node<ET, CL>(a.Lg, R, se - 1, me, Ln);
        ld_node<EF, CL>(Gsrc, R, se - 1, me, Gn);
        if (HAS_RHS) ld_node<D, CL>(a.yg, R, se - 1, me, yn);
    }
    for (int s = R - 2; s >= 0; --s) {
        if (s < len - 1) {
            double Lt[ET], G[EF], x[D];
#pragma unroll
            for (int e = 0; e < ET; ++e) Lt[e] = Ln[e];
#pragma unroll
            for (int e = 0; e < EF; ++e) G[e] = Gn[e];
#pragma unroll
            for (int e = 0; e < D; ++e) x[e] = HAS_RHS ? yn[e] : 0.0;
            if (s > 0) {
                ld_node<ET, CL>(a.Lg, R, s - 1, me, Ln);
                ld_node<EF, CL>(Gsrc, R, s - 1, me, Gn);
                if (HAS_RHS) ld_node<D, CL>(a.yg, R, s - 1, me, yn);
            }
            double invd[D], X[ET], H[EF], Ssub[EF], Sig[ET];
#pragma unroll
            for (int j = 0; j < D; ++j) invd[j] = rcp_nr(Lt[tix(j, j)]);
            tri_inverse<D>(Lt, invd, X);
            tri_t_tri<D>(X, Sig);                 // P = L^{-T} L^{-1}
            if (USE_S) {
                // H = aS S P
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) t = __builtin_fma(G[i * D + k], Sig[six(k, j)], t);
                        H[i * D + j] = a.aS * t;
                    }
            } else {
                gemm_full_tri<D>(G, X, H);        // H = L_{t+1,t} L_tt^{-1}
            }
            gemm_sym_full<D>(Sn, H, Ssub);        // S_{t+1,t+1} H
#pragma unroll
            for (int e = 0; e < EF; ++e) Ssub[e] = -Ssub[e];
            gemm_tn_sym_acc<D>(Ssub, H, -1.0, Sig);  // + H^T S_{t+1,t+1} H = - S_{t+1,t}^T H
            if (HAS_RHS) {
                double t[D];
                gemv_t<D>(G, xn, t);              // G^T x_n  (USE_S: S^T x_n, still to be scaled and multiplied by L^{-1})
                if (USE_S) {
                    double u[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k <= i; ++k) acc = __builtin_fma(X[tix(i, k)], t[k], acc);
                        u[i] = a.aS * acc;
                    }
#pragma unroll
                    for (int e = 0; e < D; ++e) t[e] = u[e];
                }
#pragma unroll
                for (int e = 0; e < D; ++e) x[e] -= t[e];
                trsv_lower_t<D>(Lt, invd, x);
                st_node<D, CL>(a.mug, R, s, me, x);
#pragma unroll
                for (int e = 0; e < D; ++e) xn[e] = x[e];
            }
            st_node<ET, CL>(a.Sigg, R, s, me, Sig);
            if (WANT_SUB) st_node<EF, CL>(a.Subg, R, s, me, Ssub);
            if (WANT_MOM) {
                double mm[3 * D];
#pragma unroll
                for (int i = 0; i < D; ++i) { mm[i] = x[i]; mm[D + i] = Sig[tix(i, i)]; mm[2 * D + i] = Ssub[i * D + i]; }
                st_node<3 * D, CL>(a.momg, R, s, me, mm);
            }
#pragma unroll
            for (int e = 0; e < ET; ++e) Sn[e] = Sig[e];
        }
    }
    if ((WANT_SUB || WANT_MOM) && p > 0) {
        // S_{t0, t0-1} for the separator on the left, whose own blocks belong to lane-1
        double Lt[ET], G[EF], invd[D], X[ET], H[EF], Ssub[EF];
        ld_node<ET, CL>(a.Lg, R, R - 1, LaneRef::of(lane - 1), Lt);
        ld_node<EF, CL>(Gsrc, R, R - 1, LaneRef::of(lane - 1), G);
#pragma unroll
        for (int j = 0; j < D; ++j) invd[j] = rcp_nr(Lt[tix(j, j)]);
        tri_inverse<D>(Lt, invd, X);
        if (USE_S) {
            double Pm[ET];
            tri_t_tri<D>(X, Pm);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = __builtin_fma(G[i * D + k], Pm[six(k, j)], t);
                    H[i * D + j] = a.aS * t;
                }
        } else {
            gemm_full_tri<D>(G, X, H);
        }
        gemm_sym_full<D>(Sn, H, Ssub);
#pragma unroll
        for (int e = 0; e < EF; ++e) Ssub[e] = -Ssub[e];
        if (WANT_SUB) st_node<EF, CL>(a.Subg, R, R - 1, LaneRef::of(lane - 1), Ssub);
        if (WANT_MOM) {
            const LaneRef left = LaneRef::of(lane - 1);
            double* pm = a.momg + ((size_t)left.tile * R + (R - 1)) * (size_t)(3 * D * 64);
#pragma unroll
            for (int i = 0; i < D; ++i) pm[(2 * D + i) * 64 + left.l] = Ssub[i * D + i];
        }
    }
}

// CL: the level's own arrays are level >= 1 arrays of the plan workspace (addressed through coarse_off)
template <int D, bool HAS_RHS, bool HAS_UP, bool WANT_SUB, bool WANT_MOM, bool USE_S = false, bool CL = false>
static __global__ __launch_bounds__(64) void k_backward(SweepArgs a) {
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    backward_body<D, HAS_RHS, HAS_UP, WANT_SUB, WANT_MOM, USE_S, CL>(a, lane, LaneRef{(int)blockIdx.x, (int)threadIdx.x});
}

// ---- coarse levels in one launch per pass ----------------------------------------------------------------------------------------
// The levels above the finest one are latency-bound (few lanes, a handful of dependent steps each): launched one kernel per level
// they cost a launch boundary and a cold start apiece (five levels x three passes per factorisation + selected inverse at the
// headline size).  Here ONE workgroup owns one chain and walks the levels l0 .. top itself: lane p of the workgroup is segment p of
// the chain at every level, levels are separated by a workgroup barrier, and the level data stay in the plan workspace (written and
// re-read by the same CU, so they are served from its L1 / the XCD's L2).  The level bodies are the same device functions the
// per-level kernels run, so both routes produce identical numbers.
MFGM_HD void bind_level_inputs(const Plan& P, int l, double* ws, SweepArgs& a) {
    // inputs of level l >= 1 are the reduced system written by reduce(l-1)
    a.Dg = ws + P.off_Dhat[l];
    a.Dcorr = ws + P.off_Rsub[l];
    a.Sg = ws + P.off_S[l];
    a.rg = ws + P.off_rhat[l];
    a.rcorr = ws + P.off_rho[l];
    a.aD = a.aS = a.aR = 1.0;
    a.Lg = ws + P.off_L[l];
    a.Gg = ws + P.off_G[l];
    a.yg = ws + P.off_y[l];
    a.Sigg = ws + P.off_Sig[l];
    a.Subg = nullptr;
    a.mug = ws + P.off_mu[l];
    a.part = nullptr;
}
MFGM_HD void bind_up(const Plan& P, int l, double* ws, SweepArgs& a) {
    // coarser level l+1
    a.up = P.lv[l + 1];
    a.uDhat = ws + P.off_Dhat[l + 1];
    a.uRsub = ws + P.off_Rsub[l + 1];
    a.uS = ws + P.off_S[l + 1];
    a.urhat = ws + P.off_rhat[l + 1];
    a.urho = ws + P.off_rho[l + 1];
    a.uL = ws + P.off_L[l + 1];
    a.uy = ws + P.off_y[l + 1];
    a.uSig = ws + P.off_Sig[l + 1];
    a.umu = ws + P.off_mu[l + 1];
}
MFGM_HD SweepArgs coarse_level_args(const Plan& P, int l, double* ws, int* info) {
    SweepArgs a = {};
    a.lv = P.lv[l];
    a.info = info;
    bind_level_inputs(P, l, ws, a);
    if (l < P.nlevels - 1) bind_up(P, l, ws, a);
    return a;
}

// ---- per-chain sum of the per-lane partials ---------------------------------------------------------
// part: [2*Lpad]; out_logdet[b] = sum_p part[b*P+p]; out_quad[b] likewise.  One wave per chain.
static __global__ __launch_bounds__(256) void k_sum_partials(const double* part, int P, int Lpad, double* out_logdet,
                                                            double* out_quad) {
    // one block per chain; fixed summation order (deterministic)
    __shared__ double sh0[4], sh1[4];
    const int b = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        s0 += part[(size_t)b * P + p];
        s1 += part[(size_t)Lpad + (size_t)b * P + p];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { sh0[threadIdx.x >> 6] = s0; sh1[threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        double t0 = 0.0, t1 = 0.0;
        for (int w = 0; w < nw; ++w) { t0 += sh0[w]; t1 += sh1[w]; }
        if (out_logdet) out_logdet[b] = t0;
        if (out_quad) out_quad[b] = t1;
    }
}

// The same sum in two deterministic stages for long partial arrays (the wavefront-per-node local kernels of the wide path keep one
// partial per node: 200 000 per chain at config 5): kSumSplit blocks per chain sum a contiguous slice each into scratch
// [2][B * kSumSplit], then k_sum_partials adds those.
constexpr int kSumSplit = 128;
static __global__ __launch_bounds__(256) void k_sum_partials_split(const double* part, int P, int stride2, int has2, double* scratch) {
    __shared__ double sh0[4], sh1[4];
    const int g = blockIdx.x, b = blockIdx.y, B = gridDim.y;
    const int slice = (P + kSumSplit - 1) / kSumSplit;
    const int lo = g * slice, hi = min(P, lo + slice);
    double s0 = 0.0, s1 = 0.0;
    for (int p = lo + threadIdx.x; p < hi; p += blockDim.x) {
        s0 += part[(size_t)b * P + p];
        if (has2) s1 += part[(size_t)stride2 + (size_t)b * P + p];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { sh0[threadIdx.x >> 6] = s0; sh1[threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[(size_t)b * kSumSplit + g] = sh0[0] + sh0[1] + sh0[2] + sh0[3];
        scratch[(size_t)B * kSumSplit + (size_t)b * kSumSplit + g] = sh1[0] + sh1[1] + sh1[2] + sh1[3];
    }
}
// host helper: out0[b] = sum_p part[b*P + p], out1[b] = sum_p part[stride2 + b*P + p]
inline int launch_sum_partials(const double* part, int P, int stride2, int B, double* out0, double* out1, double* scratch, hipStream_t st) {
    if (P >= 8192 && scratch) {
        hipLaunchKernelGGL(k_sum_partials_split, dim3(kSumSplit, B), dim3(256), 0, st, part, P, stride2, out1 != nullptr, scratch);
        if (hipGetLastError() != hipSuccess) return 3;
        hipLaunchKernelGGL(k_sum_partials, dim3(B), dim3(256), 0, st, (const double*)scratch, kSumSplit, B * kSumSplit, out0, out1);
    } else {
        hipLaunchKernelGGL(k_sum_partials, dim3(B), dim3(256), 0, st, part, P, stride2, out0, out1);
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

}  // namespace mfgm
