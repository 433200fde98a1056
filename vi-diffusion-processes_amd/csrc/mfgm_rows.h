// Row-per-lane bodies for the SMALL coarse levels of the lane-per-segment plans (d <= 8).
//
// Above level 0 a chain has few segments (headline: 256 / 64 / 16 / 4 / 1) and every level is a short chain of dependent d x d block steps.
// With one lane per segment (mfgm_sweeps.h) a step is ~1 500 dependent instructions of a single lane while the other lanes of the
// workgroup idle: 5.5 us (factor) / 4.2 us (selected inverse) per step on MI355X, 19 % of the CVI-DP step at the headline size.  Here a
// segment is worked by ONE 16-LANE DPP ROW: lane i (< d) of the row holds row i of every block in registers, and the rows of other
// matrices reach it through `v_mov_b64_dpp row_newbcast:j` (a one-instruction broadcast of lane j's double to its row: no LDS, no
// barrier, no readlane -- four segments share a wavefront and may diverge freely, DPP never leaves the row).  A step becomes ~400-600
// instructions per lane.  The algorithms are those of the wavefront-per-segment kernels (mfgm_wide.h: kw_reduce / kw_forward /
// kw_backward); the arrays are the narrow plans' triangle-packed level arrays (mfgm_sweeps.h, ld_node<E, true>), read and
// written in place, so the level-0 kernels and the lane-per-segment bodies of the larger levels see exactly what they saw before.
//
// In the reduce body the two halves of a row do different work on the same broadcasts: lanes 0..7 carry the pivot block F and the
// coupling G = S L^-T, lanes 8..15 the spike Z = W^T towards the left separator, so that G G^T (lower) and Z G^T (upper) are one loop.
//
// Replaces, like the bodies it stands in for, banded_matrices' cholesky_band / solve_triang_mat / inverse_from_cholesky_band behind
// block_tri_diag.py:330-331,350,440 and ssm_gaussian_transformations.py:443-444.
#pragma once
#include "mfgm_sweeps.h"

namespace mfgm {

template <int J>
MFGM_DEV double row_bcast_c(double x) {
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + J, 0xF, 0xF, true);     // row_newbcast:J
}
// lane j of the caller's 16-lane row (j must fold to a constant: every use sits in a fully unrolled loop)
MFGM_DEV double rb(double x, int j) {
    switch (j) {
        case 0: return row_bcast_c<0>(x);
        case 1: return row_bcast_c<1>(x);
        case 2: return row_bcast_c<2>(x);
        case 3: return row_bcast_c<3>(x);
        case 4: return row_bcast_c<4>(x);
        case 5: return row_bcast_c<5>(x);
        case 6: return row_bcast_c<6>(x);
        case 7: return row_bcast_c<7>(x);
        case 8: return row_bcast_c<8>(x);
        case 9: return row_bcast_c<9>(x);
        case 10: return row_bcast_c<10>(x);
        case 11: return row_bcast_c<11>(x);
        case 12: return row_bcast_c<12>(x);
        case 13: return row_bcast_c<13>(x);
        case 14: return row_bcast_c<14>(x);
        default: return row_bcast_c<15>(x);
    }
}

// ---- one row of a block of the level arrays (i < D; loads are branch-free, stores are the caller's to guard) --------------------------
// A node is named by (lane, R, s) of its level, as in ld_node<E, true>; At{lane - 1, R, R - 1} is the left separator of segment `lane`.
struct At { int lane, R, s; };
template <int E>
MFGM_DEV const double* nm_node(const double* __restrict__ base, At n) { return base + coarse_off<E>(n.lane, n.R, n.s); }
template <int E>
MFGM_DEV double* nm_node(double* __restrict__ base, At n) { return base + coarse_off<E>(n.lane, n.R, n.s); }

template <int D>
MFGM_DEV void ld_sym_row(const double* __restrict__ base, At node, int i, double (&o)[D]) {
    const double* p = nm_node<MFGM_NTRI(D)>(base, node);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const int hi = max(i, k), lo = min(i, k);
        o[k] = p[(hi * (hi + 1) / 2 + lo) * 64];
    }
}
// row i of a lower-triangular block (zeros above the diagonal)
template <int D>
MFGM_DEV void ld_low_row(const double* __restrict__ base, At node, int i, double (&o)[D]) {
    const double* p = nm_node<MFGM_NTRI(D)>(base, node);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double v = p[(k <= i ? i * (i + 1) / 2 + k : 0) * 64];
        o[k] = (k <= i) ? v : 0.0;
    }
}
// row i of the TRANSPOSE of a lower-triangular block: o[k] = L[k][i] (zeros for k < i)
template <int D>
MFGM_DEV void ld_lowT_row(const double* __restrict__ base, At node, int i, double (&o)[D]) {
    const double* p = nm_node<MFGM_NTRI(D)>(base, node);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double v = p[(k >= i ? k * (k + 1) / 2 + i : 0) * 64];
        o[k] = (k >= i) ? v : 0.0;
    }
}
template <int D>
MFGM_DEV void ld_full_row(const double* __restrict__ base, At node, int i, double (&o)[D]) {
    const double* p = nm_node<D * D>(base, node);
#pragma unroll
    for (int k = 0; k < D; ++k) o[k] = p[(i * D + k) * 64];
}
template <int D>
MFGM_DEV void ld_full_col(const double* __restrict__ base, At node, int i, double (&o)[D]) {
    const double* p = nm_node<D * D>(base, node);
#pragma unroll
    for (int k = 0; k < D; ++k) o[k] = p[(k * D + i) * 64];
}
template <int D>
MFGM_DEV double ld_vec_elem(const double* __restrict__ base, At node, int i) { return nm_node<D>(base, node)[i * 64]; }

template <int D>
MFGM_DEV void st_low_row(double* __restrict__ base, At node, int i, const double (&v)[D]) {
    double* p = nm_node<MFGM_NTRI(D)>(base, node);
#pragma unroll
    for (int k = 0; k < D; ++k)
        if (k <= i) p[(i * (i + 1) / 2 + k) * 64] = v[k];
}
template <int D>
MFGM_DEV void st_full_row(double* __restrict__ base, At node, int i, const double (&v)[D]) {
    double* p = nm_node<D * D>(base, node);
#pragma unroll
    for (int k = 0; k < D; ++k) p[(i * D + k) * 64] = v[k];
}
template <int D>
MFGM_DEV void st_full_col(double* __restrict__ base, At node, int i, const double (&v)[D]) {
    double* p = nm_node<D * D>(base, node);
#pragma unroll
    for (int k = 0; k < D; ++k) p[(k * D + i) * 64] = v[k];
}

// ---- row arithmetic ------------------------------------------------------------------------------------------------------------------
// Cholesky of the block whose row i (lower part) lane i of the row's LOWER half holds in F, in place; X <- X L^-T for the row X of
// another matrix in every lane (both halves), on the same broadcasts of L.  invd[j] = 1 / L_jj.
template <int D>
MFGM_DEV void rows_chol_rsolve(double (&F)[D], double (&invd)[D], double (&X)[D], int i, int& bad) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double acc = F[j], x = X[j];
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const double ljk = rb(F[k], j);
            acc = __builtin_fma(-F[k], ljk, acc);
            x = __builtin_fma(-X[k], ljk, x);
        }
        double piv = rb(acc, j);
        if (!(piv > 0.0)) { bad = 1; piv = 1.0; }
        const double inv = rsqrt_nr(piv);
        invd[j] = inv;
        F[j] = (i > j) ? acc * inv : ((i == j) ? piv * inv : 0.0);
        X[j] = x * inv;
    }
}
// y = L^-1 h, one element per lane of the lower half
template <int D>
MFGM_DEV double rows_fsolve(const double (&L)[D], const double (&invd)[D], double h, int i) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const double yj = rb(h, j) * invd[j];
        h = (i == j) ? yj : ((i > j) ? __builtin_fma(-L[j], yj, h) : h);
    }
    return h;
}
// sum_k A_i[k] v_k, v one element per lane of the lower half
template <int D>
MFGM_DEV double rows_mv(const double (&A)[D], double v) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) t = __builtin_fma(A[k], rb(v, k), t);
    return t;
}

// node of the coarser level that separator q of chain b is
MFGM_DEV At up_node(const SweepArgs& a, int b, int q) {
    const int uP = a.up.P, uR = a.up.R;
    return At{b * uP + q / uR, uR, q % uR};
}

// ---- reduce (levels >= 1: the inputs are the reduced system of the level below, all scales 1, corrections present) ---------------
template <int D, bool HAS_RHS>
MFGM_DEV void rows_reduce(const SweepArgs& a, const int lane, const int r) {
    const int P = a.lv.P, R = a.lv.R;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, a.lv.n - p * R);
    const int i = min(r & 7, D - 1);
    const bool up = r >= 8, mine = (r & 7) < D;
    const At left{lane - 1, R, R - 1};          // the left separator is the last node of segment p-1
    int bad = 0;

    // lower half: F (row i of the pivot block), h;  upper half: X = Z (row i of W^T), Racc, rho
    double F[D], X[D], Racc[D], h = 0.0, rho = 0.0;
    {
        double c[D];
        ld_sym_row<D>(a.Dg, At{lane, R, 0}, i, F);
        ld_sym_row<D>(a.Dcorr, At{lane, R, 0}, i, c);
#pragma unroll
        for (int k = 0; k < D; ++k) F[k] -= c[k];
    }
    if (p > 0) {
        ld_full_col<D>(a.Sg, left, i, X);       // the left separator is the last node of segment p-1
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) X[k] = 0.0;
    }
    if (HAS_RHS) h = ld_vec_elem<D>(a.rg, At{lane, R, 0}, i) - ld_vec_elem<D>(a.rcorr, At{lane, R, 0}, i);
#pragma unroll
    for (int k = 0; k < D; ++k) Racc[k] = 0.0;

    // the raw blocks of step s+1 are requested before step s is worked (a step is one memory round trip otherwise)
    double Gq[D], Fq[D], cq[D], hq = 0.0, hcq = 0.0;
    auto load_step = [&](int s) {
        ld_full_row<D>(a.Sg, At{lane, R, s}, i, Gq);
        ld_sym_row<D>(a.Dg, At{lane, R, s + 1}, i, Fq);
        ld_sym_row<D>(a.Dcorr, At{lane, R, s + 1}, i, cq);
        if (HAS_RHS) { hq = ld_vec_elem<D>(a.rg, At{lane, R, s + 1}, i); hcq = ld_vec_elem<D>(a.rcorr, At{lane, R, s + 1}, i); }
    };
    if (len > 1) load_step(0);
    for (int s = 0; s < len - 1; ++s) {
        double Fn[D], invd[D], hn = HAS_RHS ? hq - hcq : 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            X[k] = up ? X[k] : Gq[k];
            Fn[k] = up ? 0.0 : Fq[k] - cq[k];
        }
        if (s + 1 < len - 1) load_step(s + 1);
        rows_chol_rsolve<D>(F, invd, X, i, bad);      // lower: G <- S L^-T,  upper: Z <- Z L^-T  (Z^T = L^-1 W)
        if (HAS_RHS) {
            const double y = rows_fsolve<D>(F, invd, h, i);
            const double t = rows_mv<D>(X, y);        // lower: G y,  upper: Z y = W^T y
            rho += t;
            hn -= t;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0, t2 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                t = __builtin_fma(X[k], rb(X[k], j), t);          // lower: (G G^T)_ij,  upper: (Z G^T)_ij
                t2 = __builtin_fma(X[k], rb(X[k], 8 + j), t2);    // upper: (Z Z^T)_ij
            }
            Fn[j] -= t;                                           // lower: F' = D' - G G^T,  upper: Z' = -Z G^T  (W' = -G W)
            Racc[j] += t2;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) { F[k] = Fn[k]; X[k] = Fn[k]; }
        h = hn;
    }
    if (!up && mine) {
        const At nq = up_node(a, b, p);
        st_low_row<D>(a.uDhat, nq, i, F);
        nm_node<D>(a.urhat, nq)[i * 64] = h;
        if (p == P - 1) {
            double z[D];
#pragma unroll
            for (int k = 0; k < D; ++k) z[k] = 0.0;
            st_low_row<D>(a.uRsub, nq, i, z);
            st_full_row<D>(a.uS, nq, i, z);
            nm_node<D>(a.urho, nq)[i * 64] = 0.0;
        }
    }
    if (up && mine && p > 0) {
        const At nq = up_node(a, b, p - 1);
        st_full_col<D>(a.uS, nq, i, X);               // couples separator p-1 -> p:  S~ = W = Z^T
        st_low_row<D>(a.uRsub, nq, i, Racc);
        nm_node<D>(a.urho, nq)[i * 64] = rho;
    }
    if (bad && r == 0) flag_not_pd(a.info, a.lv.level, lane);
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------------
template <int D, bool HAS_RHS, bool HAS_UP>
MFGM_DEV void rows_forward(const SweepArgs& a, const int lane, const int r) {
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    const int i = min(r & 7, D - 1);
    const bool mine = r < D;
    const At left{lane - 1, R, R - 1};          // the left separator is the last node of segment p-1
    int bad = 0;
    double C[D], c = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) C[k] = 0.0;
    if (HAS_UP && p > 0) {
        // natural-order state at the separator on the left:  F_a = Ltil Ltil^T + R_p,  h_a = Ltil ytil + rho_p
        const At nq = up_node(a, b, p - 1);
        double Lt[D], Fa[D], Ga[D], invd[D];
        ld_low_row<D>(a.uL, nq, i, Lt);
        ld_sym_row<D>(a.uRsub, nq, i, Fa);
        ld_full_row<D>(a.Sg, left, i, Ga);
        double ha = 0.0;
        if (HAS_RHS) {
            const double yt = ld_vec_elem<D>(a.uy, nq, i);
            ha = ld_vec_elem<D>(a.urho, nq, i) + rows_mv<D>(Lt, yt);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = Fa[j];
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(Lt[k], rb(Lt[k], j), t);
            Fa[j] = t;
        }
        rows_chol_rsolve<D>(Fa, invd, Ga, i, bad);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(Ga[k], rb(Ga[k], j), t);
            C[j] = t;
        }
        if (HAS_RHS) c = rows_mv<D>(Ga, rows_fsolve<D>(Fa, invd, ha, i));
    }
    double Fq[D], cq[D], Gq[D], hq = 0.0, hcq = 0.0;
    auto load_step = [&](int s) {
        ld_sym_row<D>(a.Dg, At{lane, R, s}, i, Fq);
        ld_sym_row<D>(a.Dcorr, At{lane, R, s}, i, cq);
        if (p * R + s + 1 < n) {
            ld_full_row<D>(a.Sg, At{lane, R, s}, i, Gq);
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) Gq[k] = 0.0;
        }
        if (HAS_RHS) { hq = ld_vec_elem<D>(a.rg, At{lane, R, s}, i); hcq = ld_vec_elem<D>(a.rcorr, At{lane, R, s}, i); }
    };
    load_step(0);
    for (int s = 0; s < len; ++s) {
        double F[D], G[D], invd[D], h = HAS_RHS ? hq - hcq - c : 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { F[k] = Fq[k] - cq[k] - C[k]; G[k] = Gq[k]; }
        if (s + 1 < len) load_step(s + 1);
        rows_chol_rsolve<D>(F, invd, G, i, bad);
        double y = 0.0;
        if (HAS_RHS) y = rows_fsolve<D>(F, invd, h, i);
        if (mine) {
            st_low_row<D>(a.Lg, At{lane, R, s}, i, F);
            st_full_row<D>(a.Gg, At{lane, R, s}, i, G);
            if (HAS_RHS) nm_node<D>(a.yg, At{lane, R, s})[i * 64] = y;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(G[k], rb(G[k], j), t);
            C[j] = t;
        }
        if (HAS_RHS) c = rows_mv<D>(G, y);
    }
    if (bad && r == 0) flag_not_pd(a.info, a.lv.level, lane);
}

// ---- backward --------------------------------------------------------------------------------------------------------------------------
// Transposed factors held by rows, as in kw_backward: Xt = L^-T, Gt = G^T, Ht = Xt Gt, so that
//   Sigma_t = Xt Xt^T + Ht Sigma_n Ht^T,   x_t = Xt (y - Gt x_n).
// Xt comes from a column-oriented back substitution on the rows of U = L^T: no transpose, one reciprocal per lane.
template <int D>
MFGM_DEV void rows_inv_t(const double (&U)[D], const double uii, int i, double (&Xt)[D]) {
    // U: row i of L^T, uii its diagonal element
    const double dinv = rcp_nr(uii);
#pragma unroll
    for (int c = 0; c < D; ++c) Xt[c] = (c == i) ? 1.0 : 0.0;
#pragma unroll
    for (int j = D - 1; j >= 0; --j) {
#pragma unroll
        for (int c = j; c < D; ++c) {
            Xt[c] = (i == j) ? Xt[c] * dinv : Xt[c];
            if (j > 0) {
                const double xb = rb(Xt[c], j);
                Xt[c] = (i < j) ? __builtin_fma(-U[j], xb, Xt[c]) : Xt[c];
            }
        }
    }
}

template <int D, bool HAS_RHS, bool HAS_UP>
MFGM_DEV void rows_backward(const SweepArgs& a, const int lane, const int r) {
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R), se = len - 1;
    const int i = min(r & 7, D - 1);
    const bool mine = r < D;

    double Sn[D], xn = 0.0;
    if (HAS_UP) {
        const At nq = up_node(a, b, p);
        ld_sym_row<D>(a.uSig, nq, i, Sn);
        if (HAS_RHS) xn = ld_vec_elem<D>(a.umu, nq, i);
    } else {
        double Xt[D], U[D];
        ld_lowT_row<D>(a.Lg, At{lane, R, se}, i, U);
        rows_inv_t<D>(U, nm_node<MFGM_NTRI(D)>(a.Lg, At{lane, R, se})[(i * (i + 1) / 2 + i) * 64], i, Xt);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(Xt[k], rb(Xt[k], j), t);
            Sn[j] = t;
        }
        if (HAS_RHS) xn = rows_mv<D>(Xt, ld_vec_elem<D>(a.yg, At{lane, R, se}, i));
    }
    if (mine) {
        st_low_row<D>(a.Sigg, At{lane, R, se}, i, Sn);
        if (HAS_RHS) nm_node<D>(a.mug, At{lane, R, se})[i * 64] = xn;
    }
    double Uq[D], Gq[D], uq = 1.0, yq = 0.0;
    auto load_step = [&](int s) {
        ld_lowT_row<D>(a.Lg, At{lane, R, s}, i, Uq);
        uq = nm_node<MFGM_NTRI(D)>(a.Lg, At{lane, R, s})[(i * (i + 1) / 2 + i) * 64];
        ld_full_col<D>(a.Gg, At{lane, R, s}, i, Gq);
        if (HAS_RHS) yq = ld_vec_elem<D>(a.yg, At{lane, R, s}, i);
    };
    if (len > 1) load_step(len - 2);
    for (int s = len - 2; s >= 0; --s) {
        double Xt[D], Gt[D], U[D], Ht[D], T1[D], Sig[D];
        const double y = yq, uii = uq;
#pragma unroll
        for (int k = 0; k < D; ++k) { U[k] = Uq[k]; Gt[k] = Gq[k]; }
        if (s > 0) load_step(s - 1);
        rows_inv_t<D>(U, uii, i, Xt);
#pragma unroll
        for (int k = 0; k < D; ++k) { Ht[k] = 0.0; T1[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < D; ++k) {                 // Ht = Xt Gt
            const double x = Xt[k];
#pragma unroll
            for (int j = 0; j < D; ++j) Ht[j] = __builtin_fma(x, rb(Gt[j], k), Ht[j]);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {                 // T1 = Ht Sigma_n
            const double x = Ht[k];
#pragma unroll
            for (int j = 0; j < D; ++j) T1[j] = __builtin_fma(x, rb(Sn[j], k), T1[j]);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {                 // Sigma_t = Xt Xt^T + T1 Ht^T
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                t = __builtin_fma(Xt[k], rb(Xt[k], j), t);
                t = __builtin_fma(T1[k], rb(Ht[k], j), t);
            }
            Sig[j] = t;
        }
        if (HAS_RHS) {
            const double v = y - rows_mv<D>(Gt, xn);
            xn = rows_mv<D>(Xt, v);
        }
        if (mine) {
            st_low_row<D>(a.Sigg, At{lane, R, s}, i, Sig);
            if (HAS_RHS) nm_node<D>(a.mug, At{lane, R, s})[i * 64] = xn;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) Sn[k] = Sig[k];
    }
}

// ---- the fused coarse-level kernels (see the "coarse levels in one launch per pass" note in mfgm_sweeps.h) ----------------------------
// One workgroup per chain walks the levels l0 .. top.  A level whose chains have at most `rows_p` segments runs on the row bodies above
// (16 lanes per segment), the larger ones on the lane-per-segment bodies of mfgm_sweeps.h; both read and write the same arrays.
constexpr int kCoarseBlock = 256;     // one wavefront per SIMD: the lane-per-segment bodies need the whole register file
constexpr int kRowsMaxP = kCoarseBlock / 16;

// reduce l0 .. top-1, then forward top .. l0 (l0 >= 1); grid = B workgroups
template <int D, bool HAS_RHS>
static __global__ __launch_bounds__(kCoarseBlock) void k_coarse_factor(Plan P, int l0, double* ws, int* info, int rows_p) {
    const int K = P.nlevels - 1, b = blockIdx.x;
    const int g = threadIdx.x >> 4, r = threadIdx.x & 15;
    for (int l = l0; l < K; ++l) {
        const SweepArgs a = coarse_level_args(P, l, ws, info);
        if (a.lv.P <= rows_p) {
            if (g < a.lv.P) rows_reduce<D, HAS_RHS>(a, b * a.lv.P + g, r);
        } else {
            for (int p = threadIdx.x; p < a.lv.P; p += kCoarseBlock) {
                const int lane = b * a.lv.P + p;
                reduce_body<D, HAS_RHS, true>(a, lane, LaneRef::of(lane));
            }
        }
        __syncthreads();
    }
    {
        const SweepArgs a = coarse_level_args(P, K, ws, info);
        if (a.lv.P <= rows_p) {
            if (g < a.lv.P) rows_forward<D, HAS_RHS, false>(a, b * a.lv.P + g, r);
        } else {
            for (int p = threadIdx.x; p < a.lv.P; p += kCoarseBlock) {
                const int lane = b * a.lv.P + p;
                forward_body<D, HAS_RHS, true, false>(a, lane, LaneRef::of(lane));
            }
        }
        __syncthreads();
    }
    for (int l = K - 1; l >= l0; --l) {
        const SweepArgs a = coarse_level_args(P, l, ws, info);
        if (a.lv.P <= rows_p) {
            if (g < a.lv.P) rows_forward<D, HAS_RHS, true>(a, b * a.lv.P + g, r);
        } else {
            for (int p = threadIdx.x; p < a.lv.P; p += kCoarseBlock) {
                const int lane = b * a.lv.P + p;
                forward_body<D, HAS_RHS, true, true>(a, lane, LaneRef::of(lane));
            }
        }
        __syncthreads();
    }
}

// backward top .. l0
template <int D, bool HAS_RHS>
static __global__ __launch_bounds__(kCoarseBlock) void k_coarse_backward(Plan P, int l0, double* ws, int rows_p) {
    const int K = P.nlevels - 1, b = blockIdx.x;
    const int g = threadIdx.x >> 4, r = threadIdx.x & 15;
    {
        const SweepArgs a = coarse_level_args(P, K, ws, nullptr);
        if (a.lv.P <= rows_p) {
            if (g < a.lv.P) rows_backward<D, HAS_RHS, false>(a, b * a.lv.P + g, r);
        } else {
            for (int p = threadIdx.x; p < a.lv.P; p += kCoarseBlock) {
                const int lane = b * a.lv.P + p;
                backward_body<D, HAS_RHS, false, false, false, false, true>(a, lane, LaneRef::of(lane));
            }
        }
        __syncthreads();
    }
    for (int l = K - 1; l >= l0; --l) {
        const SweepArgs a = coarse_level_args(P, l, ws, nullptr);
        if (a.lv.P <= rows_p) {
            if (g < a.lv.P) rows_backward<D, HAS_RHS, true>(a, b * a.lv.P + g, r);
        } else {
            for (int p = threadIdx.x; p < a.lv.P; p += kCoarseBlock) {
                const int lane = b * a.lv.P + p;
                backward_body<D, HAS_RHS, true, false, false, false, true>(a, lane, LaneRef::of(lane));
            }
        }
        __syncthreads();
    }
}

}  // namespace mfgm
