// Level-0 sweeps of the CVI-DP loop on the STRUCTURED posterior naturals ("cq" state).
//
// For an SDE prior whose drift acts per dimension and whose diffusion is diagonal (the family the closed-form kernels of mfgm_sde.h
// cover), everything the Girsanov update ever adds to the posterior naturals theta_q is diagonal inside the d x d blocks:
// theta~ (mfgm_girsanov.h) has diagonal theta_diag / theta_sub blocks, the linearised prior's naturals are diagonal too, and the
// off-diagonal entries of the blocks are the initial Girsanov sites (one uniform value, -1e-10 in the reference,
// variational_cvi_sde.py:141-152) times the running product of (1 - lr).  The only dense d x d contributions are
//   * the data sites, at the observation nodes (n_obs << T), one block per observation -- the same block for every observation
//     under a Gaussian likelihood (its gradient -1/2 R^{-1} does not depend on q), and
//   * P0^{-1} at node 0 of every chain.
// So the resident state per node is 3d doubles instead of d(d+1)/2 + d^2 + d (18 instead of 63 at d = 6):
//   dyn = ( theta_lin [d], diag(theta_diag) [d], diag(theta_sub) [d] )      WITHOUT the data sites,
// plus two scalars (the uniform off-diagonal entries of theta_diag / theta_sub), the node-0 block p0off, and the observation sites
// given sparsely: slot[node] = index of the observation at that node or -1 (packed node order), site_lin [n, d], site_sym [ET].
// The data-site update then touches only the small site arrays (no scatter into packed arrays), the Girsanov update is
//   dyn <- (1 - lr) dyn + lr theta~,   off-diagonals <- (1 - lr) off-diagonals            (no data term at all: the sites are
//   implicit, g = theta_q - theta_prior - data, and g + lr (data - dKL/d eta) keeps theta_prior + g free of the data sites),
// and the sweeps rebuild the dense blocks in registers.  The factor L, the marginals Sigma and everything downstream stay dense.
// Arithmetic after the rebuild is that of k_reduce / k_forward / k_backward_girsanov / k_backward_kl, instruction for instruction.
#pragma once
#include "mfgm_girsanov.h"

namespace mfgm {

struct CqArgs {
    const double* dyn;        // [3D per node], level-0 lane-interleaved layout
    double dOff, sOff;        // uniform off-diagonal entries of theta_diag / theta_sub
    const double* p0off;      // device [ET]: added to theta_diag at node 0 of every chain (zero diagonal); may be null
    const int* slot;          // [R * Lpad] (packed node order [tile][step][64 lanes]): observation index or -1; null: no sites
    const double* site_lin;   // [n, D] natural: data-site nat1
    const double* site_sym;   // device [ET]: the data-site nat2 block every observation adds to theta_diag
    double* dyn_out;          // Girsanov sweep: the updated dyn (another buffer)
    double* obs_mu;           // KL sweep: marginal means / covariances at the observation nodes, [n, D] / [n, D, D]; may be null
    double* obs_cov;
    // k_forward_reduce_cq: the sweep's second wavefront makes the level-0 reduce of the NEXT factorisation -- the same dyn / offsets with
    // the observation sites (site_lin2, site_sym2) -- and writes that separator system to the pre_* arrays (level-1 layout)
    const double* site_lin2;
    const double* site_sym2;
    double* pre_Dhat; double* pre_Rsub; double* pre_S; double* pre_rhat; double* pre_rho;
};

// elements [E0, E0 + N) of a node with E doubles
template <int E, int E0, int N>
MFGM_DEV void ld_part(const double* __restrict__ base, int R, int s, LaneRef w, double (&out)[N]) {
    const double* p = base + ((size_t)w.tile * R + s) * (size_t)(E * 64) + (size_t)E0 * 64;
    if (w.nt >= 2) {
#pragma unroll
        for (int e = 0; e < N; ++e) out[e] = __builtin_nontemporal_load(p + e * 64 + w.l);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) out[e] = p[e * 64 + w.l];
    }
}
template <int E, int E0, int N>
MFGM_DEV void st_part(double* __restrict__ base, int R, int s, LaneRef w, const double (&v)[N]) {
    double* p = base + ((size_t)w.tile * R + s) * (size_t)(E * 64) + (size_t)E0 * 64;
    if (w.nt >= 1) {
#pragma unroll
        for (int e = 0; e < N; ++e) __builtin_nontemporal_store(v[e], p + e * 64 + w.l);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) p[e * 64 + w.l] = v[e];
    }
}
MFGM_DEV int cq_slot(const int* __restrict__ slot, int R, int s, LaneRef w) { return slot[((size_t)w.tile * R + s) * 64 + w.l]; }

// a value that is the same in every lane, pinned to scalar registers
MFGM_DEV double cq_uniform(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// the data-site block every observation adds to theta_diag, read ONCE per kernel into scalar registers: read inside the node loop it
// costs ET dependent loads per node, each behind an s_waitcnt vmcnt(0) that also drains the prefetch of the next record
template <int ET>
MFGM_DEV void cq_site_sym(const CqArgs& q, bool sites, double (&ssym)[ET]) {
#pragma unroll
    for (int e = 0; e < ET; ++e) ssym[e] = sites ? q.site_sym[e] : 0.0;
#pragma unroll
    for (int e = 0; e < ET; ++e) ssym[e] = cq_uniform(ssym[e]);
}

// theta_sub block of a transition from its diagonal
template <int D>
MFGM_DEV void cq_sub(const double (&sd)[D], double sOff, double (&G)[D * D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) G[i * D + j] = (i == j) ? sd[i] : sOff;
}

// ---- reduce --------------------------------------------------------------------------------------------------------------------
// reduce_body<D, true, false> on the cq state.  Record s+1 (and the slot of node s+1) is requested one step ahead; the site's linear
// part is gathered at the top of the step that ends by consuming it.
// LEAN: the kernel is held to 256 registers (two wavefronts per SIMD) so that its wavefronts can share a SIMD with those of
// k_forward_cq (193 registers): launched on a second stream it then rides in the forward sweep's memory waits (that sweep is
// bandwidth-bound with two thirds of its issue slots idle; this one is arithmetic-bound on 18 doubles per node).  What the reduce only
// accumulates -- the Gram correction Racc, rho -- lives in LDS, [element][lane]: one conflict-free ds_read / ds_write_b64 per element and
// node, 13.5 KB per wavefront.
#ifndef CQ_LEAN_ATTR
#define CQ_LEAN_ATTR
#endif
template <int D, bool LEAN>
MFGM_DEV void reduce_cq_body(const SweepArgs& a, const CqArgs& q, double* lds_acc);
template <int D>
static __global__ __launch_bounds__(64) void k_reduce_cq(SweepArgs a, CqArgs q) { reduce_cq_body<D, false>(a, q, nullptr); }
template <int D>
static __global__ __launch_bounds__(64) CQ_LEAN_ATTR void k_reduce_cq_lean(SweepArgs a, CqArgs q) {
    __shared__ double lds_acc[(MFGM_NTRI(D) + D) * 64];
    reduce_cq_body<D, true>(a, q, lds_acc);
}
template <int D, bool LEAN>
MFGM_DEV void reduce_cq_body(const SweepArgs& a, const CqArgs& q, double* lds_acc) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, E3 = 3 * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = a.lv.P, R = a.lv.R;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, a.lv.n - p * R);
    const bool sites = (q.slot != nullptr);
    int bad = 0;
    double ssym[ET];
    cq_site_sym<ET>(q, sites, ssym);

    double F[ET], W[EF], h[D], Racc[ET], rho[D], sdc[D];     // sdc: diag theta_sub of the node being eliminated
    {
        double r0[E3];
        ld_node<E3>(q.dyn, R, 0, me, r0);
        const int s0 = sites ? cq_slot(q.slot, R, 0, me) : -1;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) F[tix(i, j)] = (i == j) ? r0[D + i] : q.dOff;
#pragma unroll
        for (int i = 0; i < D; ++i) { h[i] = r0[i]; sdc[i] = r0[2 * D + i]; }
        if (s0 >= 0) {
#pragma unroll
            for (int e = 0; e < ET; ++e) F[e] += ssym[e];
#pragma unroll
            for (int i = 0; i < D; ++i) h[i] += q.site_lin[(size_t)s0 * D + i];
        }
        if (p == 0 && q.p0off) {
#pragma unroll
            for (int e = 0; e < ET; ++e) F[e] += q.p0off[e];
        }
#pragma unroll
        for (int e = 0; e < ET; ++e) F[e] *= a.aD;
#pragma unroll
        for (int i = 0; i < D; ++i) h[i] *= a.aR;
    }
    if (p > 0) {
        double sl_[D];
        ld_part<E3, 2 * D, D>(q.dyn, R, R - 1, LaneRef::of(lane - 1), sl_);
        cq_sub<D>(sl_, q.sOff, W);
#pragma unroll
        for (int e = 0; e < EF; ++e) W[e] *= a.aS;
    } else {
#pragma unroll
        for (int e = 0; e < EF; ++e) W[e] = 0.0;
    }
#pragma unroll
    for (int e = 0; e < ET; ++e) Racc[e] = 0.0;
#pragma unroll
    for (int e = 0; e < D; ++e) rho[e] = 0.0;
    if constexpr (LEAN) {
#pragma unroll
        for (int e = 0; e < ET + D; ++e) lds_acc[e * 64 + me.l] = 0.0;
    }

    double rn[E3];
    int sn = -1;
    if (!LEAN && len > 1) {
        ld_node<E3>(q.dyn, R, 1, me, rn);
        if (sites) sn = cq_slot(q.slot, R, 1, me);
    }
    for (int s = 0; s < R - 1; ++s) {
        if (s < len - 1) {
            double rc[E3], sl[D];
            int sc;
            if constexpr (LEAN) {
                // the record of node s + 1 is requested at the top of the step that ends by consuming it (one record in flight instead
                // of two: 36 registers; the wavefront that shares the SIMD covers what this leaves of the latency)
                ld_node<E3>(q.dyn, R, s + 1, me, rc);
                sc = sites ? cq_slot(q.slot, R, s + 1, me) : -1;
            } else {
#pragma unroll
                for (int e = 0; e < E3; ++e) rc[e] = rn[e];
                sc = sn;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) sl[i] = 0.0;
            if (!LEAN && sc >= 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) sl[i] = q.site_lin[(size_t)sc * D + i];
            }
            if (!LEAN && s + 1 < len - 1) {
                ld_node<E3>(q.dyn, R, s + 2, me, rn);
                if (sites) sn = cq_slot(q.slot, R, s + 2, me);
            }
            // eliminate interior node s
            double invd[D], G[EF];
            chol_inplace<D>(F, invd, bad);
            trsm_left_lower<D>(F, invd, W);        // W := L^{-1} W   (spike towards the left separator)
            if constexpr (LEAN) {
#pragma unroll
                for (int e = 0; e < ET; ++e) Racc[e] = lds_acc[e * 64 + me.l];
            }
            syrk_t_acc<D>(W, Racc);                // R += W^T W
            if constexpr (LEAN) {
#pragma unroll
                for (int e = 0; e < ET; ++e) lds_acc[e * 64 + me.l] = Racc[e];
#pragma unroll
                for (int e = 0; e < D; ++e) rho[e] = lds_acc[(ET + e) * 64 + me.l];
            }
            {
                trsv_lower<D>(F, invd, h);         // y := L^{-1} h
                double t[D];
                gemv_t<D>(W, h, t);
#pragma unroll
                for (int e = 0; e < D; ++e) rho[e] += t[e];
            }
            if constexpr (LEAN) {
#pragma unroll
                for (int e = 0; e < D; ++e) lds_acc[(ET + e) * 64 + me.l] = rho[e];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) G[i * D + j] = a.aS * ((i == j) ? sdc[i] : q.sOff);
            trsm_right_lower_t<D>(F, invd, G);     // G := S L^{-T}
            if constexpr (LEAN) {
                __builtin_amdgcn_sched_barrier(0);
                if (sc >= 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) sl[i] = q.site_lin[(size_t)sc * D + i];
                }
            }
            // Schur complement onto node s+1
            syrk_set<D>(G, F);
            const double cnt = (sc >= 0) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double dn = __builtin_fma(cnt, ssym[tix(i, j)], (i == j) ? rc[D + i] : q.dOff);
                    F[tix(i, j)] = __builtin_fma(a.aD, dn, -F[tix(i, j)]);
                }
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
            {
                double t[D];
                gemv<D>(G, h, t);
#pragma unroll
                for (int e = 0; e < D; ++e) h[e] = __builtin_fma(a.aR, rc[e] + sl[e], -t[e]);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) sdc[i] = rc[2 * D + i];
        }
    }
    if constexpr (LEAN) {
#pragma unroll
        for (int e = 0; e < ET; ++e) Racc[e] = lds_acc[e * 64 + me.l];
#pragma unroll
        for (int e = 0; e < D; ++e) rho[e] = lds_acc[(ET + e) * 64 + me.l];
    }
    // separator of this segment is node q = p of the coarser level (level >= 1 layout: coarse_off)
    const int uP = a.up.P, uR = a.up.R;
    {
        const int qq = p, ul = b * uP + qq / uR, us = qq % uR;
        st_node<ET, true>(a.uDhat, uR, us, LaneRef::of(ul), F);
        st_node<D, true>(a.urhat, uR, us, LaneRef::of(ul), h);
        if (p == P - 1) {
            st_node_zero<ET, true>(a.uRsub, uR, us, LaneRef::of(ul));
            st_node_zero<D, true>(a.urho, uR, us, LaneRef::of(ul));
            st_node_zero<EF, true>(a.uS, uR, us, LaneRef::of(ul));
        }
    }
    if (p > 0) {
        const int qq = p - 1, ul = b * uP + qq / uR, us = qq % uR;
        st_node<EF, true>(a.uS, uR, us, LaneRef::of(ul), W);      // couples separator p-1 -> p
        st_node<ET, true>(a.uRsub, uR, us, LaneRef::of(ul), Racc);
        st_node<D, true>(a.urho, uR, us, LaneRef::of(ul), rho);
    }
    if (bad) flag_not_pd(a.info, a.lv.level, lane);
}

// ---- the P-form of the level-0 factor --------------------------------------------------------------------------------------------------
// The cq sweeps keep, per node, P_t = F_t^{-1} = L_t^{-T} L_t^{-1} (packed lower triangle, in the array the dense route calls L) and
// z_t = L_t^{-T} y_t (in the array it calls y) instead of the Cholesky factor and the forward-substituted right-hand side: the two
// backward sweeps of a step -- latency-bound at one wavefront per SIMD -- used to rebuild exactly these from L_t (6 reciprocals, a
// triangular inverse, X^T X and a transposed substitution per node, ~200 of their 1 450 / 1 800 instructions), while the forward sweep
// is bandwidth-bound with issue slots to spare; the arithmetic that produces P and z is the same, it moved.  With
// S = diag(theta_sub) + sOff (1 1^T - I) (csrc header above) the products with S are a row scaling plus a rank-one term:
//   forward   C = (aS S) P (aS S)^T,  c = (aS S) z                                   (was: G = aS S L^{-T} by substitution, G G^T, G y)
//   backward  H = aS S P,  Sigma_{t+1,t} = -Sigma_{t+1} H,  Sigma_t = P - Sigma_{t+1,t}^T H,  x_t = z_t - P (aS S x_{t+1}).
template <int D>
MFGM_DEV void cq_carry(const double (&Pm)[MFGM_NTRI(D)], const double (&z)[D], const double (&sd)[D], double aS, double sOff,
                       double (&C)[MFGM_NTRI(D)], double (&c)[D]) {
    if (D == 1) sOff = 0.0;                 // a 1 x 1 block has no off-diagonal entry: whatever the state carries there is not a value
    const double cs = aS * sOff;
    double e[D], u[D], sig = 0.0, sz = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        e[i] = aS * (sd[i] - sOff);
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t += Pm[six(k, i)];
        u[i] = t;
        sig += t;
        sz += z[i];
    }
    const double c2 = cs * cs * sig;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double wi = cs * e[i] * u[i];
#pragma unroll
        for (int j = 0; j <= i; ++j) C[tix(i, j)] = __builtin_fma(e[i] * e[j], Pm[tix(i, j)], wi + __builtin_fma(cs * e[j], u[j], c2));
        c[i] = __builtin_fma(e[i], z[i], cs * sz);
    }
}
// one step of the backward sweeps: Sigma_{t+1} (Sn), x_{t+1} (xn) -> Sigma_t (Sig), Sigma_{t+1,t} (Ssub), x_t (x holds z_t on entry)
template <int D>
MFGM_DEV void backward_p_step(const double (&Pm)[MFGM_NTRI(D)], const double (&sd)[D], double sOff, double aS, const double (&Sn)[MFGM_NTRI(D)],
                              const double (&xn)[D], double (&Sig)[MFGM_NTRI(D)], double (&Ssub)[D * D], double (&x)[D]) {
    if (D == 1) sOff = 0.0;
    const double cs = aS * sOff;
    double e[D], u[D], H[D * D], tg[D], sx = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        e[i] = aS * (sd[i] - sOff);
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t += Pm[six(k, i)];
        u[i] = cs * t;
        sx += xn[i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) H[i * D + j] = __builtin_fma(e[i], Pm[six(i, j)], u[j]);
        tg[i] = __builtin_fma(e[i], xn[i], cs * sx);
    }
    gemm_sym_full<D>(Sn, H, Ssub);
#pragma unroll
    for (int k = 0; k < D * D; ++k) Ssub[k] = -Ssub[k];
#pragma unroll
    for (int k = 0; k < MFGM_NTRI(D); ++k) Sig[k] = Pm[k];
    gemm_tn_sym_acc<D>(Ssub, H, -1.0, Sig);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double t = x[i];
#pragma unroll
        for (int k = 0; k < D; ++k) t = __builtin_fma(-Pm[six(i, k)], tg[k], t);
        x[i] = t;
    }
}

// ---- forward -------------------------------------------------------------------------------------------------------------------
// forward_body<D, true, false, true> on the cq state (L_{t+1,t} is never stored).  A node's record is requested one step ahead; its
// slot two steps ahead, so that the gather of the site's linear part rides with the record.
// SYNC: the body runs as one wavefront of a two-wavefront workgroup (k_forward_reduce_cq) and meets the other one at a workgroup
// barrier once per node, so that the two stay within a node of each other and the record either of them loads first is still in the
// CU's L1 / the XCD's L2 when the other asks for it.
#ifndef MFGM_CQ_SYNC_EVERY
#define MFGM_CQ_SYNC_EVERY 2       // nodes between the barriers of k_forward_reduce_cq's two wavefronts (a power of two)
#endif
template <int D, bool SYNC>
MFGM_DEV void forward_cq_body(const SweepArgs& a, const CqArgs& q, const int lane, const LaneRef me) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, E3 = 3 * D;
    const int P = a.lv.P, R = a.lv.R, Lp = a.lv.Lpad, n = a.lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    const bool sites = (q.slot != nullptr);
    int bad = 0;
    double ssym[ET];
    cq_site_sym<ET>(q, sites, ssym);

    double C[ET], c[D];
#pragma unroll
    for (int e = 0; e < ET; ++e) C[e] = 0.0;
#pragma unroll
    for (int e = 0; e < D; ++e) c[e] = 0.0;

    if (p > 0) {
        // natural-order Cholesky state at the separator to the left:  F_a = Ltil Ltil^T + R_p ,  h_a = Ltil ytil + rho_p
        const int uP = a.up.P, uR = a.up.R;
        const int qq = p - 1, ul = b * uP + qq / uR, us = qq % uR;
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
        {
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
        }
        chol_inplace<D>(Fa, invd, bad);
        trsv_lower<D>(Fa, invd, ha);
        double Ga[EF], sl_[D];
        ld_part<E3, 2 * D, D>(q.dyn, R, R - 1, LaneRef::of(lane - 1), sl_);
        cq_sub<D>(sl_, q.sOff, Ga);
#pragma unroll
        for (int e = 0; e < EF; ++e) Ga[e] *= a.aS;
        trsm_right_lower_t<D>(Fa, invd, Ga);
        syrk_set<D>(Ga, C);
        gemv<D>(Ga, ha, c);
    } else if (q.p0off) {
        // node 0 of the chain: theta_diag carries p0off as well; F = aD (D + p0off) - 0
#pragma unroll
        for (int e = 0; e < ET; ++e) C[e] = -(a.aD * q.p0off[e]);
    }
    __builtin_amdgcn_sched_barrier(0);

    LogAcc la;
    la.init();
    double quad = 0.0;

    double rn[E3], sln[D];
    int sA = -1, sB = -1;            // slot of the node whose record is in rn / of the node after it
    ld_node<E3>(q.dyn, R, 0, me, rn);
    if (sites) {
        sA = cq_slot(q.slot, R, 0, me);
        if (len > 1) sB = cq_slot(q.slot, R, 1, me);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) sln[i] = 0.0;
    if (sA >= 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) sln[i] = q.site_lin[(size_t)sA * D + i];
    }
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            double F[ET], h[D];
            const double cnt = (sA >= 0) ? 1.0 : 0.0;
            const bool has_next = (p * R + s + 1 < n);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double dn = __builtin_fma(cnt, ssym[tix(i, j)], (i == j) ? rn[D + i] : q.dOff);
                    F[tix(i, j)] = __builtin_fma(a.aD, dn, -C[tix(i, j)]);
                }
            double sdn[D];
#pragma unroll
            for (int i = 0; i < D; ++i) sdn[i] = has_next ? rn[2 * D + i] : 0.0;      // (the chain's last node has no transition: its slot is not read)
#pragma unroll
            for (int e = 0; e < D; ++e) h[e] = __builtin_fma(a.aR, rn[e] + sln[e], -c[e]);
            if (s + 1 < len) {
                ld_node<E3>(q.dyn, R, s + 1, me, rn);
                sA = sB;
#pragma unroll
                for (int i = 0; i < D; ++i) sln[i] = 0.0;
                if (sA >= 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) sln[i] = q.site_lin[(size_t)sA * D + i];
                }
                if (sites && s + 2 < len) sB = cq_slot(q.slot, R, s + 2, me);
            }
            double invd[D];
            chol_inplace<D>(F, invd, bad);
            trsv_lower<D>(F, invd, h);
#pragma unroll
            for (int j = 0; j < D; ++j) la.mul(F[tix(j, j)]);
            la.renorm();
#pragma unroll
            for (int j = 0; j < D; ++j) quad = __builtin_fma(h[j], h[j], quad);
            // P-form of the factor (cq_pform below): P = F^{-1} = L^{-T} L^{-1} and z = L^{-T} y go to the factor arrays
            double X[ET], Pm[ET], z[D];
            tri_inverse<D>(F, invd, X);
            tri_t_tri<D>(X, Pm);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double t = 0.0;
#pragma unroll
                for (int k = i; k < D; ++k) t = __builtin_fma(X[tix(k, i)], h[k], t);
                z[i] = t;
            }
            st_node<ET>(a.Lg, R, s, me, Pm);
            st_node<D>(a.yg, R, s, me, z);
            // carried to the next node: C = (aS S) P (aS S)^T, c = (aS S) z with S = diag(theta_sub) + sOff (1 1^T - I)
            cq_carry<D>(Pm, z, sdn, has_next ? a.aS : 0.0, q.sOff, C, c);
        }
        if constexpr (SYNC) {
            if ((s & (MFGM_CQ_SYNC_EVERY - 1)) == MFGM_CQ_SYNC_EVERY - 1) __syncthreads();
        }
    }
    if (a.part) {
        a.part[lane] = la.value();
        a.part[Lp + lane] = quad;
    }
    if (bad) flag_not_pd(a.info, a.lv.level, lane);
}
// NT: cache policy of the level-0 arrays (LevelDesc::nt, ld_node / st_node) as a compile-time constant -- behind a run-time branch
// the compiler merges the two copies of a load and drops the hint.
template <int D, int NT = 0>
static __global__ __launch_bounds__(64) void k_forward_cq(SweepArgs a, CqArgs q) {
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    forward_cq_body<D, false>(a, q, lane, LaneRef{(int)blockIdx.x, (int)threadIdx.x});
}

// ---- forward sweep and the NEXT factorisation's reduce as two wavefronts of one workgroup ------------------------------------------------
// (what k_forward_cq on one stream and k_reduce_cq_lean on a second do, without the second read of the records: the pair on two streams
// moves 3.28 GB at the memory system's ceiling, 148 B per node of it the reduce re-reading what the forward sweep has just streamed.)
// Measured (A/Bs on one box each, tools/ab_pipe.sh, tools/ab_lib.sh): 0.544-0.566 ms per launch whatever the box, against 0.40-0.46 ms
// for the forward sweep alone and 0.53-0.57 for the two-stream pair; step -0.02 ... -0.03 ms against the two-stream form, -0.17 ... -0.20
// against no pipelining.  SQ counters: each wavefront 45 % active, i.e. the SIMD's VALU issue ~90 % taken by the two together (1 570
// instructions per node, 1 255 of them fp64): the kernel is bound by instruction issue now, not by the records' traffic.  Without the
// per-node barriers the wavefronts drift apart and the second read goes to HBM again (0.592 ms); a second record in flight for the
// forward wavefront changes nothing (0.566 against 0.563).
// Wavefront 0 of a workgroup is the forward sweep of a tile, wavefront 1 the reduce of the same tile for the state (q.site_lin2,
// q.site_sym2) -> q.pre_*.  Both must fit 256 registers (two wavefronts per SIMD): the forward body does (193); the reduce keeps what it
// cannot -- the spike W, the Gram correction Racc, rho -- in LDS, [element][lane] (conflict-free 8-byte accesses, 32 KB per workgroup,
// four workgroups per CU), and streams them through registers a column / an element at a time.
template <int D>
MFGM_DEV void reduce_cq_lds_body(const SweepArgs& a, const CqArgs& q, const int lane, const LaneRef me, double* __restrict__ lds,
                                 const int l /* physical lane: the LDS slot */) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, E3 = 3 * D;
    const int P = a.lv.P, R = a.lv.R;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, a.lv.n - p * R);
    const bool sites = (q.slot != nullptr);
    double* ldsW = lds;                     // [EF][64]
    double* ldsR = lds + EF * 64;           // [ET][64]
    double* ldsr = ldsR + ET * 64;          // [D][64]
    int bad = 0;
    double F[ET], h[D], sdc[D];
    {
        double r0[E3];
        ld_node<E3>(q.dyn, R, 0, me, r0);
        const int s0 = sites ? cq_slot(q.slot, R, 0, me) : -1;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) F[tix(i, j)] = (i == j) ? r0[D + i] : q.dOff;
#pragma unroll
        for (int i = 0; i < D; ++i) { h[i] = r0[i]; sdc[i] = r0[2 * D + i]; }
        if (s0 >= 0) {
#pragma unroll
            for (int e = 0; e < ET; ++e) F[e] += q.site_sym2[e];
#pragma unroll
            for (int i = 0; i < D; ++i) h[i] += q.site_lin2[(size_t)s0 * D + i];
        }
        if (p == 0 && q.p0off) {
#pragma unroll
            for (int e = 0; e < ET; ++e) F[e] += q.p0off[e];
        }
#pragma unroll
        for (int e = 0; e < ET; ++e) F[e] *= a.aD;
#pragma unroll
        for (int i = 0; i < D; ++i) h[i] *= a.aR;
        double sl_[D];
#pragma unroll
        for (int i = 0; i < D; ++i) sl_[i] = 0.0;
        if (p > 0) ld_part<E3, 2 * D, D>(q.dyn, R, R - 1, LaneRef::of(lane - 1), sl_);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) ldsW[(i * D + j) * 64 + l] = (p > 0) ? a.aS * ((i == j) ? sl_[i] : q.sOff) : 0.0;
#pragma unroll
        for (int e = 0; e < ET; ++e) ldsR[e * 64 + l] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) ldsr[i * 64 + l] = 0.0;
    }
    for (int s = 0; s < R; ++s) {
        if (s < len - 1) {
            // the record of node s + 1: the forward wavefront of the workgroup asks for the same lines in its iteration s + 1
            double rc[E3], sl[D];
            ld_node<E3>(q.dyn, R, s + 1, me, rc);
            const int sc = sites ? cq_slot(q.slot, R, s + 1, me) : -1;
            double invd[D];
            chol_inplace<D>(F, invd, bad);
            trsv_lower<D>(F, invd, h);                     // y := L^{-1} h
            {
                // W := L^{-1} W (all of it through registers once), Racc += W^T W and rho += W^T y element by element from / to LDS
                double W[EF];
#pragma unroll
                for (int e = 0; e < EF; ++e) W[e] = ldsW[e * 64 + l];
                trsm_left_lower<D>(F, invd, W);
#pragma unroll
                for (int e = 0; e < EF; ++e) ldsW[e * 64 + l] = W[e];
                // (the accumulators come in as ONE batch of LDS reads: element by element each read-modify-write is a round trip of its
                //  own, 27 dependent ones per node)
                double Racc[ET], rho[D];
#pragma unroll
                for (int e = 0; e < ET; ++e) Racc[e] = ldsR[e * 64 + l];
#pragma unroll
                for (int i = 0; i < D; ++i) rho[i] = ldsr[i * 64 + l];
                syrk_t_acc<D>(W, Racc);
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double t = rho[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) t = __builtin_fma(W[k * D + i], h[k], t);
                    rho[i] = t;
                }
#pragma unroll
                for (int e = 0; e < ET; ++e) ldsR[e * 64 + l] = Racc[e];
#pragma unroll
                for (int i = 0; i < D; ++i) ldsr[i * 64 + l] = rho[i];
            }
            __builtin_amdgcn_sched_barrier(0);
            double G[EF];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) G[i * D + j] = a.aS * ((i == j) ? sdc[i] : q.sOff);
            trsm_right_lower_t<D>(F, invd, G);             // G := S L^{-T}
            syrk_set<D>(G, F);
#pragma unroll
            for (int i = 0; i < D; ++i) sl[i] = 0.0;
            if (sc >= 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) sl[i] = q.site_lin2[(size_t)sc * D + i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double dn = (i == j) ? rc[D + i] : q.dOff;
                    if (sc >= 0) dn += q.site_sym2[tix(i, j)];
                    F[tix(i, j)] = __builtin_fma(a.aD, dn, -F[tix(i, j)]);
                }
            {                                              // W := -G W: W in as one batch, out column by column
                double W[EF];
#pragma unroll
                for (int e = 0; e < EF; ++e) W[e] = ldsW[e * 64 + l];
#pragma unroll
                for (int c = 0; c < D; ++c)
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) t = __builtin_fma(G[i * D + k], W[k * D + c], t);
                        ldsW[(i * D + c) * 64 + l] = -t;
                    }
            }
            {
                double t[D];
                gemv<D>(G, h, t);
#pragma unroll
                for (int e = 0; e < D; ++e) h[e] = __builtin_fma(a.aR, rc[e] + sl[e], -t[e]);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) sdc[i] = rc[2 * D + i];
        }
        if ((s & (MFGM_CQ_SYNC_EVERY - 1)) == MFGM_CQ_SYNC_EVERY - 1) __syncthreads();
    }
    double W[EF], Racc[ET], rho[D];
#pragma unroll
    for (int e = 0; e < EF; ++e) W[e] = ldsW[e * 64 + l];
#pragma unroll
    for (int e = 0; e < ET; ++e) Racc[e] = ldsR[e * 64 + l];
#pragma unroll
    for (int i = 0; i < D; ++i) rho[i] = ldsr[i * 64 + l];
    const int uP = a.up.P, uR = a.up.R;
    {
        const int qq = p, ul = b * uP + qq / uR, us = qq % uR;
        st_node<ET, true>(q.pre_Dhat, uR, us, LaneRef::of(ul), F);
        st_node<D, true>(q.pre_rhat, uR, us, LaneRef::of(ul), h);
        if (p == P - 1) {
            st_node_zero<ET, true>(q.pre_Rsub, uR, us, LaneRef::of(ul));
            st_node_zero<D, true>(q.pre_rho, uR, us, LaneRef::of(ul));
            st_node_zero<EF, true>(q.pre_S, uR, us, LaneRef::of(ul));
        }
    }
    if (p > 0) {
        const int qq = p - 1, ul = b * uP + qq / uR, us = qq % uR;
        st_node<EF, true>(q.pre_S, uR, us, LaneRef::of(ul), W);
        st_node<ET, true>(q.pre_Rsub, uR, us, LaneRef::of(ul), Racc);
        st_node<D, true>(q.pre_rho, uR, us, LaneRef::of(ul), rho);
    }
    if (bad) flag_not_pd(a.info, a.lv.level, lane);
}

template <int D, int NT = 0>
static __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_forward_reduce_cq(SweepArgs a, CqArgs q) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    __shared__ double lds[(EF + ET + D) * 64];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    // Both wavefronts meet at R workgroup barriers, so no lane may leave early and no barrier may sit in divergent control flow: the
    // padding lanes of the last tile repeat the work of the last live lane (same loads, same results, same values stored to the same
    // places) instead of returning.
    const int lane = min((int)blockIdx.x * 64 + l, a.lv.L - 1);
    const LaneRef me{(int)blockIdx.x, lane & 63, NT >= 1 ? 1 : 0};     // streamed stores only: the second reader of a record is served from the cache
    if (wave == 0) forward_cq_body<D, true>(a, q, lane, me);
    else reduce_cq_lds_body<D>(a, q, lane, me, lds, l);
}

// ---- backward helpers ------------------------------------------------------------------------------------------------------------
// backward_s_left with the left separator's theta_sub rebuilt from the cq state
template <int D>
MFGM_DEV void backward_s_left_cq(const SweepArgs& a, const CqArgs& q, int R, LaneRef left, const double (&Sn)[MFGM_NTRI(D)],
                                 double (&Gd)[D], double (&SnH)[D * D]) {
    constexpr int ET = MFGM_NTRI(D);
    double H[D * D], Pm[ET], e[D], u[D];
    ld_node<ET>(a.Lg, R, R - 1, left, Pm);                   // P-form: the array holds P = F^{-1}
    ld_part<3 * D, 2 * D, D>(q.dyn, R, R - 1, left, Gd);
    const double so = (D == 1) ? 0.0 : q.sOff;
    const double cs = a.aS * so;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        e[i] = a.aS * (Gd[i] - so);
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t += Pm[six(k, i)];
        u[i] = cs * t;
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) H[i * D + j] = __builtin_fma(e[i], Pm[six(i, j)], u[j]);
    gemm_sym_full<D>(Sn, H, SnH);
}

// new diag(theta_diag) and diag(theta_sub) of a node
template <int D>
MFGM_DEV void cq_store_blocks(const SdeParams& pr, const CqArgs& q, int R, int s, LaneRef w, bool has_next, const double (&dg)[D],
                              const double (&sb)[D], const double (&Sdiag)[D]) {
    const double lr = pr.lr, kp = 1.0 - pr.lr;
    double ad[D], sd[D];
    ld_part<3 * D, D, D>(q.dyn, R, s, w, ad);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        ad[i] = kp * ad[i] + lr * dg[i];
        sd[i] = has_next ? kp * Sdiag[i] + lr * sb[i] : Sdiag[i];
    }
    st_part<3 * D, D, D>(q.dyn_out, R, s, w, ad);
    st_part<3 * D, 2 * D, D>(q.dyn_out, R, s, w, sd);
}
// the same with the old values already in registers (requested a step ahead by the sweep)
template <int D>
MFGM_DEV void cq_store_blocks_pre(const SdeParams& pr, const CqArgs& q, int R, int s, LaneRef w, const double (&dg)[D],
                                  const double (&sb)[D], const double (&Sdiag)[D], const double (&ad_old)[D]) {
    const double lr = pr.lr, kp = 1.0 - pr.lr;
    double ad[D], sd[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        ad[i] = kp * ad_old[i] + lr * dg[i];
        sd[i] = kp * Sdiag[i] + lr * sb[i];
    }
    st_part<3 * D, D, D>(q.dyn_out, R, s, w, ad);
    st_part<3 * D, 2 * D, D>(q.dyn_out, R, s, w, sd);
}
template <int D>
MFGM_DEV void cq_store_lin_pre(const SdeParams& pr, const CqArgs& q, int R, int s, LaneRef w, const double (&own)[D],
                               const double (&prev)[D], const double (&lin_old)[D]) {
    const double lr = pr.lr, kp = 1.0 - pr.lr;
    double a1[D];
#pragma unroll
    for (int i = 0; i < D; ++i) a1[i] = kp * lin_old[i] + lr * (own[i] + prev[i]);
    st_part<3 * D, 0, D>(q.dyn_out, R, s, w, a1);
}
template <int D>
MFGM_DEV void cq_store_lin(const SdeParams& pr, const CqArgs& q, int R, int s, LaneRef w, const double (&own)[D],
                           const double (&prev)[D]) {
    const double lr = pr.lr, kp = 1.0 - pr.lr;
    double a1[D];
    ld_part<3 * D, 0, D>(q.dyn, R, s, w, a1);
#pragma unroll
    for (int i = 0; i < D; ++i) a1[i] = kp * a1[i] + lr * (own[i] + prev[i]);
    st_part<3 * D, 0, D>(q.dyn_out, R, s, w, a1);
}

// ---- backward sweep fused with the Girsanov-site update (k_backward_girsanov on the cq state) --------------------------------------
// fix: [D][Lpad] hand-over of lr (We - W J m) from a segment's last interior node to its separator (k_girsanov_fixup_cq adds it)
template <int D, int NT = 0>
static __global__ __launch_bounds__(64) void k_backward_girsanov_cq(SweepArgs a, SdeParams pr, CqArgs q, double* fix) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, E3 = 3 * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x, NT};
    const int P = a.lv.P, R = a.lv.R, Lp = a.lv.Lpad, n = a.lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    const int se = len - 1;
    const bool last = (p == P - 1);
    const int uP = a.up.P, uR = a.up.R;

    double Sn[ET], xn[D];
    {
        const int ul = b * uP + p / uR, us = p % uR;
        ld_node<ET, true>(a.uSig, uR, us, LaneRef::of(ul), Sn);
        ld_node<D, true>(a.umu, uR, us, LaneRef::of(ul), xn);
    }
    double pend[D];     // theta~_lin of the node one step ahead, still waiting for the pair of the node about to be visited
#pragma unroll
    for (int i = 0; i < D; ++i) pend[i] = 0.0;
    if (last) {
        // the chain's final node has no transition of its own
        double v[D], c[D], mn[D], lin[D], dg[D], sb[D], wd[D], Sd[D];
#pragma unroll
        for (int i = 0; i < D; ++i) { v[i] = Sn[tix(i, i)]; c[i] = 0.0; mn[i] = 0.0; }
        girsanov_node<D>(pr, false, xn, v, c, mn, lin, dg, sb, wd);
        ld_part<E3, 2 * D, D>(q.dyn, R, se, me, Sd);
        cq_store_blocks<D>(pr, q, R, se, me, false, dg, sb, Sd);
#pragma unroll
        for (int i = 0; i < D; ++i) pend[i] = lin[i];
    }

    // the record parts a step reads -- theta_sub and theta_diag of the node it visits, theta_lin of the node above it -- are all
    // requested one step ahead (the old theta_diag / theta_lin used to be read at the end of the step that blends them: two more
    // memory round trips per node on the critical path)
    double Ln[ET], Gdn[D], yn[D], adn[D], l1n[D];
    if (len > 1) {
        ld_node<ET>(a.Lg, R, se - 1, me, Ln);
        ld_part<E3, 2 * D, D>(q.dyn, R, se - 1, me, Gdn);
        ld_part<E3, D, D>(q.dyn, R, se - 1, me, adn);
        ld_part<E3, 0, D>(q.dyn, R, se, me, l1n);
        ld_node<D>(a.yg, R, se - 1, me, yn);
    }
    for (int s = R - 2; s >= 0; --s) {
        if (s < len - 1) {
            double Lt[ET], x[D], Gd[D], adc[D], l1c[D];
#pragma unroll
            for (int e = 0; e < ET; ++e) Lt[e] = Ln[e];
#pragma unroll
            for (int e = 0; e < D; ++e) { Gd[e] = Gdn[e]; x[e] = yn[e]; adc[e] = adn[e]; l1c[e] = l1n[e]; }
            if (s > 0) {
                ld_node<ET>(a.Lg, R, s - 1, me, Ln);
                ld_part<E3, 2 * D, D>(q.dyn, R, s - 1, me, Gdn);
                ld_part<E3, D, D>(q.dyn, R, s - 1, me, adn);
                ld_part<E3, 0, D>(q.dyn, R, s, me, l1n);
                ld_node<D>(a.yg, R, s - 1, me, yn);
            }
            double Ssub[EF], Sig[ET];
            backward_p_step<D>(Lt, Gd, q.sOff, a.aS, Sn, xn, Sig, Ssub, x);
            double v[D], c[D], lin[D], dg[D], sb[D], wd[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { v[i] = Sig[tix(i, i)]; c[i] = Ssub[i * D + i]; }
            girsanov_node<D>(pr, true, x, v, c, xn, lin, dg, sb, wd);
            cq_store_blocks_pre<D>(pr, q, R, s, me, dg, sb, Gd, adc);
            if (s + 1 == se && !last) {
                // the separator's theta_lin is assembled by the lane on the right
#pragma unroll
                for (int i = 0; i < D; ++i) fix[(size_t)i * Lp + lane] = pr.lr * wd[i];
            } else {
                cq_store_lin_pre<D>(pr, q, R, s + 1, me, pend, wd, l1c);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) { pend[i] = lin[i]; xn[i] = x[i]; }
#pragma unroll
            for (int e = 0; e < ET; ++e) Sn[e] = Sig[e];
        }
    }
    double wprev[D];
#pragma unroll
    for (int i = 0; i < D; ++i) wprev[i] = 0.0;
    if (p > 0) {
        // the separator on the left: Sigma_{t0, t0-1} as in k_backward, then its theta~ (its moments come from the coarser level)
        const LaneRef left = LaneRef::of(lane - 1);
        double Gd[D], Ssub[EF];
        backward_s_left_cq<D>(a, q, R, left, Sn, Gd, Ssub);
        double m[D], v[D], c[D], lin[D], dg[D], sb[D], zero[D];
        up_moments<D>(a, b, p - 1, m, v);
#pragma unroll
        for (int i = 0; i < D; ++i) { c[i] = -Ssub[i * D + i]; zero[i] = 0.0; }
        girsanov_node<D>(pr, true, m, v, c, xn, lin, dg, sb, wprev);
        cq_store_blocks<D>(pr, q, R, R - 1, left, true, dg, sb, Gd);
        cq_store_lin<D>(pr, q, R, R - 1, left, lin, zero);      // k_girsanov_fixup_cq adds the pair of node t-1
    }
    if (len == 1 && !last) {
        // a one-node segment has no interior node to hand the pair over: the separator's own lane writes a zero hand-over
#pragma unroll
        for (int i = 0; i < D; ++i) fix[(size_t)i * Lp + lane] = 0.0;
    }
    if (p == 0) {
        // node 0 of the chain: no transition enters it (undo the -1/2 W of theta~_diag and its -2 theta~_diag m share) and the
        // prior of x0 does: theta~_diag -= 1/2 P0^{-1} (its off-diagonal part is the constant p0off, which the blend leaves fixed);
        // -P0^{-1}(m - mu0) of F_m and +P0^{-1} m of -2 theta~_diag m leave P0^{-1} mu0
        double ad[D];
        ld_part<E3, D, D>(q.dyn_out, R, 0, me, ad);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ad[i] += 0.5 * pr.lr * (pr.W[i] - pr.P0inv[tix(i, i)]);
            double l = pend[i] - pr.W[i] * xn[i];
#pragma unroll
            for (int j = 0; j < D; ++j) l = __builtin_fma(pr.P0inv[six(i, j)], pr.mu0[j], l);
            pend[i] = l;
        }
        st_part<E3, D, D>(q.dyn_out, R, 0, me, ad);
    }
    cq_store_lin<D>(pr, q, R, 0, me, pend, wprev);
}

// theta_lin of every separator += the hand-over of the segment's last interior node
template <int D>
static __global__ __launch_bounds__(64) void k_girsanov_fixup_cq(LevelDesc lv, double* dyn_out, const double* fix) {
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const int p = lane % lv.P;
    if (p == lv.P - 1) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    double* qp = dyn_out + ((size_t)me.tile * lv.R + (lv.R - 1)) * (size_t)(3 * D * 64);
#pragma unroll
    for (int i = 0; i < D; ++i) qp[i * 64 + me.l] += fix[(size_t)i * lv.Lpad + lane];
}

// ---- backward sweep with the KL sum (k_backward_kl on the cq state) + marginals at the observation nodes --------------------------
template <int D>
MFGM_DEV void cq_store_obs(const CqArgs& q, int slot, const double (&x)[D], const double (&Sig)[MFGM_NTRI(D)]) {
    if (slot >= 0) {
        double* pm = q.obs_mu + (size_t)slot * D;
        double* pc = q.obs_cov + (size_t)slot * D * D;
#pragma unroll
        for (int i = 0; i < D; ++i) pm[i] = x[i];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) pc[i * D + j] = Sig[six(i, j)];
    }
}

template <int D, int NT = 0>
static __global__ __launch_bounds__(64) void k_backward_kl_cq(SweepArgs a, SdeParams pr, CqArgs q) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, E3 = 3 * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x, NT};
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    const int se = len - 1;
    const int uP = a.up.P, uR = a.up.R;
    const bool obs = (q.slot != nullptr && q.obs_mu != nullptr);
    double acc = 0.0;

    double Sn[ET], xn[D];
    {
        const int ul = b * uP + p / uR, us = p % uR;
        ld_node<ET, true>(a.uSig, uR, us, LaneRef::of(ul), Sn);
        ld_node<D, true>(a.umu, uR, us, LaneRef::of(ul), xn);
    }
    const bool keep = (a.Sigg != nullptr);       // the marginal arrays are optional: the ELBO needs the sums and the observation nodes only
    if (keep) {
        st_node<ET>(a.Sigg, R, se, me, Sn);
        st_node<D>(a.mug, R, se, me, xn);
    }
    if (obs) cq_store_obs<D>(q, cq_slot(q.slot, R, se, me), xn, Sn);

    double Ln[ET], Gdn[D], yn[D];
    int sn = -1;
    if (len > 1) {
        ld_node<ET>(a.Lg, R, se - 1, me, Ln);
        ld_part<E3, 2 * D, D>(q.dyn, R, se - 1, me, Gdn);
        ld_node<D>(a.yg, R, se - 1, me, yn);
        if (obs) sn = cq_slot(q.slot, R, se - 1, me);
    }
    for (int s = R - 2; s >= 0; --s) {
        if (s < len - 1) {
            double Lt[ET], x[D], Gd[D];
#pragma unroll
            for (int e = 0; e < ET; ++e) Lt[e] = Ln[e];
#pragma unroll
            for (int e = 0; e < D; ++e) { Gd[e] = Gdn[e]; x[e] = yn[e]; }
            const int sc = sn;
            if (s > 0) {
                ld_node<ET>(a.Lg, R, s - 1, me, Ln);
                ld_part<E3, 2 * D, D>(q.dyn, R, s - 1, me, Gdn);
                ld_node<D>(a.yg, R, s - 1, me, yn);
                if (obs) sn = cq_slot(q.slot, R, s - 1, me);
            }
            double Ssub[EF], Sig[ET];
            backward_p_step<D>(Lt, Gd, q.sOff, a.aS, Sn, xn, Sig, Ssub, x);
            if (keep) {
                st_node<D>(a.mug, R, s, me, x);
                st_node<ET>(a.Sigg, R, s, me, Sig);
            }
            if (obs) cq_store_obs<D>(q, sc, x, Sig);
            double v[D], c[D], vn[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { v[i] = Sig[tix(i, i)]; c[i] = Ssub[i * D + i]; vn[i] = Sn[tix(i, i)]; }
            acc += kl_transition<D>(pr, x, v, c, xn, vn);
#pragma unroll
            for (int i = 0; i < D; ++i) xn[i] = x[i];
#pragma unroll
            for (int e = 0; e < ET; ++e) Sn[e] = Sig[e];
        }
    }
    if (p > 0) {
        // the transition out of the separator on the left (its own moments come from the coarser level)
        const LaneRef left = LaneRef::of(lane - 1);
        double Gd[D], Ssub[EF];
        backward_s_left_cq<D>(a, q, R, left, Sn, Gd, Ssub);
        double m[D], v[D], c[D], vn[D];
        up_moments<D>(a, b, p - 1, m, v);
#pragma unroll
        for (int i = 0; i < D; ++i) { c[i] = -Ssub[i * D + i]; vn[i] = Sn[tix(i, i)]; }
        acc += kl_transition<D>(pr, m, v, c, xn, vn);
    } else {
        // node 0: 1/2 [ tr(P0^{-1} Sigma_0) + (m0 - mu0)^T P0^{-1} (m0 - mu0) + logdet P0 ]
        double tr = 0.0, mh = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double pm = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                pm = __builtin_fma(pr.P0inv[six(i, j)], xn[j] - pr.mu0[j], pm);
                tr = __builtin_fma(pr.P0inv[six(i, j)], Sn[six(i, j)], tr);
            }
            mh = __builtin_fma(pm, xn[i] - pr.mu0[i], mh);
        }
        acc += 0.5 * (tr + mh + pr.logdetP0);
    }
    a.part[lane] = acc;
}

// ---- conversions between the dense packed naturals and the cq state -------------------------------------------------------------------
// dense (lin VEC, diag SYM, sub FULL; lane-interleaved) -> dyn: the diagonals.  (The caller checks that the off-diagonals are
// uniform -- mfgm_cq_pack returns their range.)
template <int D>
static __global__ __launch_bounds__(64) void k_cq_pack(LevelDesc lv, const double* __restrict__ lin, const double* __restrict__ diag,
                                                       const double* __restrict__ sub, double* __restrict__ dyn,
                                                       double* __restrict__ range /* [4][Lpad]: min/max off-diag of diag, of sub */) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, E3 = 3 * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = (lane < lv.L) ? min(R, n - p * R) : 0;
    (void)b;
    double dmin = 1e300, dmax = -1e300, smin = 1e300, smax = -1e300;
    for (int s = 0; s < R; ++s) {
        double rec[E3];
        if (s < len) {
            double l[D], dd[ET], ss[EF];
            ld_node<D>(lin, R, s, me, l);
            ld_node<ET>(diag, R, s, me, dd);
            const bool has_next = (p * R + s + 1 < n);
            if (has_next) ld_node<EF>(sub, R, s, me, ss);
#pragma unroll
            for (int i = 0; i < D; ++i) { rec[i] = l[i]; rec[D + i] = dd[tix(i, i)]; rec[2 * D + i] = has_next ? ss[i * D + i] : 0.0; }
            const bool node0 = (p == 0 && s == 0);        // node 0 carries P0^{-1}: its off-diagonals are not part of the uniform value
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    if (i > j && !node0) { dmin = fmin(dmin, dd[tix(i, j)]); dmax = fmax(dmax, dd[tix(i, j)]); }
                    if (i != j && has_next) { smin = fmin(smin, ss[i * D + j]); smax = fmax(smax, ss[i * D + j]); }
                }
        } else {
#pragma unroll
            for (int e = 0; e < E3; ++e) rec[e] = 0.0;
        }
        st_node<E3>(dyn, R, s, me, rec);
    }
    range[lane] = dmin; range[lv.Lpad + lane] = dmax; range[2 * lv.Lpad + lane] = smin; range[3 * lv.Lpad + lane] = smax;
}

// dyn + constants -> dense packed naturals (WITHOUT the observation sites; p0off included)
template <int D>
static __global__ __launch_bounds__(64) void k_cq_unpack(LevelDesc lv, CqArgs q, double* __restrict__ lin, double* __restrict__ diag,
                                                         double* __restrict__ sub) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, E3 = 3 * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int p = lane % P;
    const int len = min(R, n - p * R);
    for (int s = 0; s < len; ++s) {
        double rec[E3], l[D], dd[ET], ss[EF];
        ld_node<E3>(q.dyn, R, s, me, rec);
        const bool has_next = (p * R + s + 1 < n);
#pragma unroll
        for (int i = 0; i < D; ++i) l[i] = rec[i];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double v = (i == j) ? rec[D + i] : q.dOff;
                if (p == 0 && s == 0 && q.p0off) v += q.p0off[tix(i, j)];
                dd[tix(i, j)] = v;
            }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) ss[i * D + j] = has_next ? ((i == j) ? rec[2 * D + i] : q.sOff) : 0.0;
        st_node<D>(lin, R, s, me, l);
        st_node<ET>(diag, R, s, me, dd);
        st_node<EF>(sub, R, s, me, ss);
    }
}

// slot[packed node index of node_ids[i]] = i  (slot pre-filled with -1 by the caller)
static __global__ void k_cq_slots(LevelDesc lv, int T, const long long* __restrict__ node_ids, int nn, int* __restrict__ slot,
                                  int* __restrict__ dup) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    const unsigned id = (unsigned)node_ids[i];
    const unsigned cb = id / (unsigned)T, t = id - cb * (unsigned)T;
    const unsigned p = t / lv.R, s = t - p * lv.R, lane = cb * lv.P + p;
    const size_t at = ((size_t)(lane >> 6) * lv.R + s) * 64 + (lane & 63);
    if (atomicExch(&slot[at], i) != -1) atomicMax(dup, 1);      // two observations at one node: not representable
}

// per-trajectory variational expectations of a multivariate Gaussian likelihood from COMPACT marginals (obs_mu [B n_per, D],
// obs_cov [B n_per, D, D] as written by k_backward_kl_cq); same partial-sum shape as k_mvn_obs_ve
template <int D>
static __global__ __launch_bounds__(256) void k_mvn_ve_compact(int n_per, const double* __restrict__ mu, const double* __restrict__ cov,
                                                              const double* __restrict__ y, const double* __restrict__ Sinv, double cst,
                                                              double* __restrict__ ve) {
    __shared__ double sS[D * D];
    __shared__ double red[256];
    for (int e = threadIdx.x; e < D * D; e += blockDim.x) sS[e] = Sinv[e];
    __syncthreads();
    const int b = blockIdx.y;
    double acc = 0.0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_per; j += gridDim.x * blockDim.x) {
        const size_t i = (size_t)b * n_per + j;
        double diff[D], v = 0.0;
#pragma unroll
        for (int r = 0; r < D; ++r) diff[r] = y[i * D + r] - mu[i * D + r];
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int c = 0; c < D; ++c) v += sS[r * D + c] * (cov[(i * D + r) * D + c] + diff[r] * diff[c]);
        acc += -0.5 * v + cst;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ve[(size_t)b * gridDim.x + blockIdx.x] = red[0];
}

}  // namespace mfgm
