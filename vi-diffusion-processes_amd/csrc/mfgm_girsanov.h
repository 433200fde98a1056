// Backward sweep of the CVI-DP refresh fused with the Girsanov-site update (variational_cvi_sde.py:279-299).
//
// The update theta_q <- (1 - lr) theta_q + lr theta~ needs, per node, only (mu_t, diag Sigma_t, diag Sigma_{t+1,t}) and mu_{t+1}
// (see k_sde_lean in mfgm_sde.h): exactly what the level-0 backward sweep holds in registers while it walks a segment.  Doing
// the update there removes the moment / marginal stores of that sweep and the separate pass that re-reads them together with
// theta_q: per node 2 d^2 + 3 d + ... doubles less traffic (d = 6: 153 doubles instead of 252 for the two kernels).
//
// The new theta_q goes to a second buffer: a lane reads theta_sub of its left neighbour's last node, which that neighbour may
// already have replaced.  One quantity crosses lanes the other way -- the (We, W J m) pair of a segment's last interior node
// enters theta~_lin of its separator, whose other terms belong to the lane on the right -- and is handed over through `fix`,
// added by k_girsanov_fixup.
#pragma once
#include "mfgm_sde.h"

namespace mfgm {

struct GirsanovArgs {
    const double* q1;   // theta_lin  (current)
    const double* qd;   // theta_diag (current); theta_sub is SweepArgs::Sg
    double* n1;         // new theta_lin / theta_diag / theta_sub
    double* nd;
    double* ns;
    double* fix;        // [D][Lpad] hand-over of lr (We - W J m) from a segment's last interior node to its separator
};

// theta~ terms of one node t >= 1 (node 0 is patched by girsanov_patch_node0) for the per-dimension cubic drift: lin = theta~_lin without the (We, W J m) pair of node t-1,
// dg / sb = diagonals of theta~_diag / theta~_sub, wd = We_t - W J_t m_t (what node t+1 still needs from this node).
template <int D>
MFGM_DEV void girsanov_node(const SdeParams& pr, bool has_next, const double (&m)[D], const double (&v)[D],
                            const double (&c)[D], const double (&mn)[D], double (&lin)[D], double (&dg)[D], double (&sb)[D],
                            double (&wd)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        // the drift constants stay in scalar registers: products of them hoisted out of the sweep loop would each occupy a vector
        // register pair for the whole sweep, which has none to spare
        double W = pr.W[i], al = pr.alpha[i], be = pr.beta[i];
        asm volatile("" : "+s"(W), "+s"(al), "+s"(be));
        double l = 0.0, g = -0.5 * W, s_ = 0.0, w_ = 0.0;
        if (has_next) {
            // Gaussian moments of u(x) = al x - be x^3 (drift_mom<0>), without the variance itself
            const double mi = m[i], vi = v[i], m2 = mi * mi, a2 = m2 + vi;
            const double ub = al * mi - be * mi * (m2 + 3.0 * vi);
            const double J = al - 3.0 * be * a2;
            const double ub_v = -3.0 * be * mi, J_m = -6.0 * be * mi, J_v = -3.0 * be;
            const double V_m = -12.0 * al * be * mi * vi + be * be * mi * vi * (36.0 * m2 + 72.0 * vi);
            const double V_v = al * al - 6.0 * al * be * (m2 + 2.0 * vi) + be * be * (9.0 * m2 * m2 + 72.0 * m2 * vi + 45.0 * vi * vi);
            const double k = W * c[i], we = W * (ub - mn[i]);
            l = k * J_m - we * J - 0.5 * W * V_m - W * J * mn[i];
            g += k * J_v - we * ub_v - 0.5 * W * V_v;
            s_ = W * J;
            w_ = we - W * J * m[i];
        }
        lin[i] = l - 2.0 * g * m[i];
        dg[i] = g;
        sb[i] = s_;
        wd[i] = w_;
    }
}

// new theta_sub off the diagonal: (1 - lr) theta_sub, known as soon as the raw block is loaded (frees its registers early)
template <int D>
MFGM_DEV void girsanov_store_sub_offdiag(const SdeParams& pr, const GirsanovArgs& g, int R, int s, LaneRef w, bool has_next,
                                         const double (&Sraw)[D * D]) {
    const double kp = has_next ? 1.0 - pr.lr : 1.0;
    double* q = g.ns + ((size_t)w.tile * R + s) * (size_t)(D * D * 64);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (i != j) q[(i * D + j) * 64 + w.l] = kp * Sraw[i * D + j];
}

// new theta_diag and the diagonal of the new theta_sub
template <int D>
MFGM_DEV void girsanov_store_blocks(const SdeParams& pr, const GirsanovArgs& g, int R, int s, LaneRef w, bool has_next,
                                    const double (&dg)[D], const double (&sb)[D], const double (&Sdiag)[D]) {
    constexpr int ET = MFGM_NTRI(D);
    const double lr = pr.lr, kp = 1.0 - pr.lr;
    double ad[ET];
    ld_node<ET>(g.qd, R, s, w, ad);
#pragma unroll
    for (int e = 0; e < ET; ++e) ad[e] *= kp;
#pragma unroll
    for (int i = 0; i < D; ++i) ad[tix(i, i)] += lr * dg[i];
    st_node<ET>(g.nd, R, s, w, ad);
    double* q = g.ns + ((size_t)w.tile * R + s) * (size_t)(D * D * 64);
#pragma unroll
    for (int i = 0; i < D; ++i) q[(i * D + i) * 64 + w.l] = has_next ? kp * Sdiag[i] + lr * sb[i] : Sdiag[i];
}

template <int D>
MFGM_DEV void girsanov_store_lin(const SdeParams& pr, const GirsanovArgs& g, int R, int s, LaneRef w, const double (&own)[D],
                                 const double (&prev)[D]) {
    const double lr = pr.lr, kp = 1.0 - pr.lr;
    double a1[D];
    ld_node<D>(g.q1, R, s, w, a1);
#pragma unroll
    for (int i = 0; i < D; ++i) a1[i] = kp * a1[i] + lr * (own[i] + prev[i]);
    st_node<D>(g.n1, R, s, w, a1);
}

// ---- the step of the level-0 backward sweep that both kernels below share (k_backward's USE_S variant: L_{t+1,t} rebuilt from theta_sub) ----
// first half: X = L^{-1}, P = L^{-T} L^{-1} (left in Sig), H = L_{t+1,t} L^{-1} = aS S P, tg = S^T x_next
template <int D>
MFGM_DEV void backward_s_head(const double (&Lt)[MFGM_NTRI(D)], const double (&G)[D * D], double aS, const double (&xn)[D],
                              double (&invd)[D], double (&X)[MFGM_NTRI(D)], double (&Sig)[MFGM_NTRI(D)], double (&H)[D * D],
                              double (&tg)[D]) {
#pragma unroll
    for (int j = 0; j < D; ++j) invd[j] = rcp_nr(Lt[tix(j, j)]);
    tri_inverse<D>(Lt, invd, X);
    tri_t_tri<D>(X, Sig);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(G[i * D + k], Sig[six(k, j)], t);
            H[i * D + j] = aS * t;
        }
    gemv_t<D>(G, xn, tg);
}
// second half: Sigma_{t+1,t} = -Sigma_{t+1} H (Ssub), Sigma_t = P - Sigma_{t+1,t}^T H (Sig), mean x = L^{-T} (y - L^{-1} aS tg) (x holds y on entry)
template <int D>
MFGM_DEV void backward_s_tail(const double (&Lt)[MFGM_NTRI(D)], const double (&invd)[D], const double (&X)[MFGM_NTRI(D)], double aS,
                              const double (&Sn)[MFGM_NTRI(D)], const double (&H)[D * D], const double (&tg)[D],
                              double (&Sig)[MFGM_NTRI(D)], double (&Ssub)[D * D], double (&x)[D]) {
    gemm_sym_full<D>(Sn, H, Ssub);
#pragma unroll
    for (int e = 0; e < D * D; ++e) Ssub[e] = -Ssub[e];
    gemm_tn_sym_acc<D>(Ssub, H, -1.0, Sig);
    double u[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k <= i; ++k) acc = __builtin_fma(X[tix(i, k)], tg[k], acc);
        u[i] = aS * acc;
    }
#pragma unroll
    for (int e = 0; e < D; ++e) x[e] -= u[e];
    trsv_lower_t<D>(Lt, invd, x);
}
// the separator on the left of a segment: +Sigma_{t0} H of its transition (the caller negates), from its L and theta_sub blocks
template <int D>
MFGM_DEV void backward_s_left(const SweepArgs& a, int R, LaneRef left, const double (&Sn)[MFGM_NTRI(D)], double (&G)[D * D],
                              double (&SnH)[D * D]) {
    constexpr int ET = MFGM_NTRI(D);
    double Lt[ET], invd[D], X[ET], H[D * D], Pm[ET];
    ld_node<ET>(a.Lg, R, R - 1, left, Lt);
    ld_node<D * D>(a.Sg, R, R - 1, left, G);
#pragma unroll
    for (int j = 0; j < D; ++j) invd[j] = rcp_nr(Lt[tix(j, j)]);
    tri_inverse<D>(Lt, invd, X);
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
    gemm_sym_full<D>(Sn, H, SnH);
}
// (mu, diag Sigma) of node q of the coarser level
template <int D>
MFGM_DEV void up_moments(const SweepArgs& a, int b, int q, double (&m)[D], double (&v)[D]) {
    constexpr int ET = MFGM_NTRI(D);
    const int uP = a.up.P, uR = a.up.R, ul = b * uP + q / uR, us = q % uR;
    const LaneRef uw = LaneRef::of(ul);
    ld_node<D, true>(a.umu, uR, us, uw, m);
    const double* ps = a.uSig + coarse_off<ET>(ul, uR, us);
#pragma unroll
    for (int i = 0; i < D; ++i) v[i] = ps[tix(i, i) * 64];
}

// Level-0 backward sweep (always below a coarser level, means wanted, L_{t+1,t} rebuilt from theta_sub: the USE_S variant of
// k_backward) that writes the updated theta_q instead of the marginals.
template <int D>
static __global__ __launch_bounds__(64) void k_backward_girsanov(SweepArgs a, SdeParams pr, GirsanovArgs g) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
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
        double v[D], c[D], mn[D], lin[D], dg[D], sb[D], wd[D], Sraw[EF];
#pragma unroll
        for (int i = 0; i < D; ++i) { v[i] = Sn[tix(i, i)]; c[i] = 0.0; mn[i] = 0.0; }
        girsanov_node<D>(pr, false, xn, v, c, mn, lin, dg, sb, wd);
        ld_node<EF>(a.Sg, R, se, me, Sraw);
        double Sd[D];
#pragma unroll
        for (int i = 0; i < D; ++i) Sd[i] = Sraw[i * D + i];
        girsanov_store_sub_offdiag<D>(pr, g, R, se, me, false, Sraw);
        girsanov_store_blocks<D>(pr, g, R, se, me, false, dg, sb, Sd);
#pragma unroll
        for (int i = 0; i < D; ++i) pend[i] = lin[i];
    }

    double Ln[ET], Gn[EF], yn[D];
    if (len > 1) {
        ld_node<ET>(a.Lg, R, se - 1, me, Ln);
        ld_node<EF>(a.Sg, R, se - 1, me, Gn);
        ld_node<D>(a.yg, R, se - 1, me, yn);
    }
    for (int s = R - 2; s >= 0; --s) {
        if (s < len - 1) {
            double Lt[ET], G[EF], x[D];
#pragma unroll
            for (int e = 0; e < ET; ++e) Lt[e] = Ln[e];
#pragma unroll
            for (int e = 0; e < EF; ++e) G[e] = Gn[e];
#pragma unroll
            for (int e = 0; e < D; ++e) x[e] = yn[e];
            if (s > 0) {
                ld_node<ET>(a.Lg, R, s - 1, me, Ln);
                ld_node<EF>(a.Sg, R, s - 1, me, Gn);
                ld_node<D>(a.yg, R, s - 1, me, yn);
            }
            double invd[D], X[ET], H[EF], Ssub[EF], Sig[ET], tg[D], Gd[D];
            backward_s_head<D>(Lt, G, a.aS, xn, invd, X, Sig, H, tg);
            girsanov_store_sub_offdiag<D>(pr, g, R, s, me, true, G);      // the raw theta_sub block is not needed any further
#pragma unroll
            for (int i = 0; i < D; ++i) Gd[i] = G[i * D + i];
            backward_s_tail<D>(Lt, invd, X, a.aS, Sn, H, tg, Sig, Ssub, x);
            double v[D], c[D], lin[D], dg[D], sb[D], wd[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { v[i] = Sig[tix(i, i)]; c[i] = Ssub[i * D + i]; }
            girsanov_node<D>(pr, true, x, v, c, xn, lin, dg, sb, wd);
            girsanov_store_blocks<D>(pr, g, R, s, me, true, dg, sb, Gd);
            if (s + 1 == se && !last) {
                // the separator's theta_lin is assembled by the lane on the right
#pragma unroll
                for (int i = 0; i < D; ++i) g.fix[(size_t)i * Lp + lane] = pr.lr * wd[i];
            } else {
                girsanov_store_lin<D>(pr, g, R, s + 1, me, pend, wd);
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
        double G[EF], Ssub[EF];
        backward_s_left<D>(a, R, left, Sn, G, Ssub);
        double m[D], v[D], c[D], lin[D], dg[D], sb[D], zero[D];
        up_moments<D>(a, b, p - 1, m, v);
#pragma unroll
        for (int i = 0; i < D; ++i) { c[i] = -Ssub[i * D + i]; zero[i] = 0.0; }
        girsanov_node<D>(pr, true, m, v, c, xn, lin, dg, sb, wprev);
        double Gd[D];
#pragma unroll
        for (int i = 0; i < D; ++i) Gd[i] = G[i * D + i];
        girsanov_store_sub_offdiag<D>(pr, g, R, R - 1, left, true, G);
        girsanov_store_blocks<D>(pr, g, R, R - 1, left, true, dg, sb, Gd);
        girsanov_store_lin<D>(pr, g, R, R - 1, left, lin, zero);      // k_girsanov_fixup adds the pair of node t-1
    }
    if (p == 0) {
        // node 0 of the chain: no transition enters it (undo the -1/2 W of theta~_diag and its -2 theta~_diag m share) and the
        // prior of x0 does: theta~_diag -= 1/2 P0^{-1}; -P0^{-1}(m - mu0) of F_m and +P0^{-1} m of -2 theta~_diag m leave P0^{-1} mu0
        double ad[ET];
        ld_node<ET>(g.nd, R, 0, me, ad);
#pragma unroll
        for (int e = 0; e < ET; ++e) ad[e] -= 0.5 * pr.lr * pr.P0inv[e];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ad[tix(i, i)] += 0.5 * pr.lr * pr.W[i];
            double l = pend[i] - pr.W[i] * xn[i];
#pragma unroll
            for (int j = 0; j < D; ++j) l = __builtin_fma(pr.P0inv[six(i, j)], pr.mu0[j], l);
            pend[i] = l;
        }
        st_node<ET>(g.nd, R, 0, me, ad);
    }
    girsanov_store_lin<D>(pr, g, R, 0, me, pend, wprev);
}

// ---- backward sweep that also accumulates E_q[log p] of the SDE prior ----------------------------------------------------------
// One transition's share of the moment-array KL (k_sde_lean MODE 0): 1/2 sum_i W_i [v' + m'^2 - 2 (J c + ubar m') + V + ubar^2]
// + 1/2 logdet Qp, from (m, v, c) of the node and (m', v') of its successor.
template <int D>
MFGM_DEV double kl_transition(const SdeParams& pr, const double (&m)[D], const double (&v)[D], const double (&c)[D],
                              const double (&mn)[D], const double (&vn)[D]) {
    double acc = 0.5 * pr.logdetQp;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double W = pr.W[i], al = pr.alpha[i], be = pr.beta[i];
        asm volatile("" : "+s"(W), "+s"(al), "+s"(be));      // see girsanov_node
        const double mi = m[i], vi = v[i], m2 = mi * mi, a2 = m2 + vi;
        const double ub = al * mi - be * mi * (m2 + 3.0 * vi);
        const double J = al - 3.0 * be * a2;
        const double V = al * al * vi - 6.0 * al * be * vi * a2 + be * be * vi * (9.0 * m2 * m2 + 36.0 * m2 * vi + 15.0 * vi * vi);
        acc += 0.5 * W * (vn[i] + mn[i] * mn[i] - 2.0 * (J * c[i] + ub * mn[i]) + V + ub * ub);
    }
    return acc;
}

// Level-0 backward sweep of the refresh before the ELBO: marginals (Sigma_tt, mu) as k_backward's USE_S variant, and, instead of
// the moment array that k_sde_lean MODE 0 would re-read, the per-lane partial of that kernel's sum directly (part[lane]).
template <int D>
static __global__ __launch_bounds__(64) void k_backward_kl(SweepArgs a, SdeParams pr) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= a.lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    const int se = len - 1;
    const int uP = a.up.P, uR = a.up.R;
    double acc = 0.0;

    double Sn[ET], xn[D];
    {
        const int ul = b * uP + p / uR, us = p % uR;
        ld_node<ET, true>(a.uSig, uR, us, LaneRef::of(ul), Sn);
        ld_node<D, true>(a.umu, uR, us, LaneRef::of(ul), xn);
    }
    st_node<ET>(a.Sigg, R, se, me, Sn);
    st_node<D>(a.mug, R, se, me, xn);

    double Ln[ET], Gn[EF], yn[D];
    if (len > 1) {
        ld_node<ET>(a.Lg, R, se - 1, me, Ln);
        ld_node<EF>(a.Sg, R, se - 1, me, Gn);
        ld_node<D>(a.yg, R, se - 1, me, yn);
    }
    for (int s = R - 2; s >= 0; --s) {
        if (s < len - 1) {
            double Lt[ET], G[EF], x[D];
#pragma unroll
            for (int e = 0; e < ET; ++e) Lt[e] = Ln[e];
#pragma unroll
            for (int e = 0; e < EF; ++e) G[e] = Gn[e];
#pragma unroll
            for (int e = 0; e < D; ++e) x[e] = yn[e];
            if (s > 0) {
                ld_node<ET>(a.Lg, R, s - 1, me, Ln);
                ld_node<EF>(a.Sg, R, s - 1, me, Gn);
                ld_node<D>(a.yg, R, s - 1, me, yn);
            }
            double invd[D], X[ET], H[EF], Ssub[EF], Sig[ET], tg[D];
            backward_s_head<D>(Lt, G, a.aS, xn, invd, X, Sig, H, tg);
            backward_s_tail<D>(Lt, invd, X, a.aS, Sn, H, tg, Sig, Ssub, x);
            st_node<D>(a.mug, R, s, me, x);
            st_node<ET>(a.Sigg, R, s, me, Sig);
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
        double G[EF], Ssub[EF];
        backward_s_left<D>(a, R, left, Sn, G, Ssub);
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

// theta_lin of every separator += the hand-over of the segment's last interior node
template <int D>
static __global__ __launch_bounds__(64) void k_girsanov_fixup(LevelDesc lv, GirsanovArgs g) {
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const int p = lane % lv.P;
    if (p == lv.P - 1) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    double* q = g.n1 + ((size_t)me.tile * lv.R + (lv.R - 1)) * (size_t)(D * 64);
#pragma unroll
    for (int i = 0; i < D; ++i) q[i * 64 + me.l] += g.fix[(size_t)i * lv.Lpad + lane];
}

}  // namespace mfgm
