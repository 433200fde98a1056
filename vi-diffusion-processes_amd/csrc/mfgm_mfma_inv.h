// MFMA sweeps in INVERSE FORM for block sizes 8 < d <= 32 (same tiling, level recursion and natural-layout arrays as mfgm_mfma.h).
//
// The Cholesky-form sweeps (mfgm_mfma.h) spend most of their time in a pivot-by-pivot Gauss-Jordan elimination whose row / column
// broadcasts go through ds_bpermute.  When the caller only wants what the selected inverse delivers (marginal blocks, means,
// log-determinant, quadratic form) the factor itself is not needed, and every step of the three passes can be written with the
// inverse of the pivot block F_t instead of its Cholesky factor:
//
//   reduce   :  Fi = F^{-1};  T_S = Fi S^T,  T_W = Fi W,  t = Fi h;   F' -= S T_S,  W' = -S T_W,  R += W^T T_W,  h' -= S t,  rho += W^T t
//   forward  :  Fi = F^{-1};  J = Fi S^T (stored transposed: J^T = S Fi),  z = Fi h;   C = S J,  c = S z;   log|F|,  h^T z
//   backward :  U = Sigma_n J^T;   Sigma_{t+1,t} = -U;   Sigma_t = Fi + J U;   x_t = z - J x_n          (no factorisation at all)
//
// and the "factor" arrays of a plan hold (Fi, J^T, z) in place of (L, L_{t+1,t}, y).  All products are Gram products gram(X, Y) = X^T Y
// of accumulator-layout tiles (Fi and Sigma are symmetric), as in mfgm_mfma.h.
//
// F^{-1} is taken in place by symmetric block sweeps with 4 x 4 pivot blocks: the four pivot rows 4k .. 4k+3 of a tile are register k
// of every lane (lane (g, c) holds row 4k + g, column c), so with P the 4 x 16 pivot panel and D its 4 x 4 pivot block
//      T = D^{-1} P            one MFMA  (A operand: D^{-1} in the lanes (g, c < 4); B operand: the panel, i.e. the lane's own register)
//      A <- A - P^T T          one MFMA  (both operands are the lane's own registers)
//      rows of the block <- T, pivot block <- -D^{-1}
// and after the four sweeps A = -F^{-1}.  The only cross-lane traffic is the read of the 10 distinct entries of D (v_readlane);
// D^{-1} is formed redundantly by every lane from two 2 x 2 determinants, whose product also accumulates log|F|.
#pragma once
#include "mfgm_mfma.h"

namespace mfgm {

// acc + sum_{a < 4} X[a][m] Y[a][n], one double per lane for each operand: lane (g, c) supplies X[g][c] and Y[g][c]
MFGM_DEV Tile mfma1(double x, double y, const Tile& acc) {
    v4d a = {acc.r[0], acc.r[1], acc.r[2], acc.r[3]};
    a = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a, 0, 0, 0);
    return Tile{{a[0], a[1], a[2], a[3]}};
}

// lower triangle of a symmetric 4 x 4 block, held uniformly by all lanes
struct Sym4 {
    double a00, a10, a11, a20, a21, a22, a30, a31, a32, a33;
};

// inverse of a symmetric positive definite 4 x 4 block through its 2 x 2 blocks [[P, Q^T], [Q, R]]:  S = R - Q P^{-1} Q^T,
// D^{-1} = [[P^{-1} + U^T S^{-1} U, -U^T S^{-1}], [-S^{-1} U, S^{-1}]],  U = Q P^{-1};  det D = det P det S
MFGM_DEV Sym4 inv4(const Sym4& m, double& detP, double& detS, int& bad) {
    detP = __builtin_fma(m.a00, m.a11, -m.a10 * m.a10);
    const bool negP = !(m.a00 > 0.0) || !(detP > 0.0);
    detP = negP ? 1.0 : detP;
    const double ip = rcp_nr(detP);
    const double p00 = m.a11 * ip, p10 = -m.a10 * ip, p11 = m.a00 * ip;
    const double u00 = __builtin_fma(m.a20, p00, m.a21 * p10), u01 = __builtin_fma(m.a20, p10, m.a21 * p11);
    const double u10 = __builtin_fma(m.a30, p00, m.a31 * p10), u11 = __builtin_fma(m.a30, p10, m.a31 * p11);
    const double s00 = m.a22 - __builtin_fma(u00, m.a20, u01 * m.a21);
    const double s10 = m.a32 - __builtin_fma(u10, m.a20, u11 * m.a21);
    const double s11 = m.a33 - __builtin_fma(u10, m.a30, u11 * m.a31);
    detS = __builtin_fma(s00, s11, -s10 * s10);
    const bool negS = !(s00 > 0.0) || !(detS > 0.0);
    detS = negS ? 1.0 : detS;
    bad |= (negP || negS) ? 1 : 0;
    const double is = rcp_nr(detS);
    Sym4 o;
    o.a22 = s11 * is; o.a32 = -s10 * is; o.a33 = s00 * is;
    o.a20 = -__builtin_fma(o.a22, u00, o.a32 * u10); o.a21 = -__builtin_fma(o.a22, u01, o.a32 * u11);
    o.a30 = -__builtin_fma(o.a32, u00, o.a33 * u10); o.a31 = -__builtin_fma(o.a32, u01, o.a33 * u11);
    o.a00 = p00 - __builtin_fma(u00, o.a20, u10 * o.a30);
    o.a10 = p10 - __builtin_fma(u01, o.a20, u11 * o.a30);
    o.a11 = p11 - __builtin_fma(u01, o.a21, u11 * o.a31);
    return o;
}

// A <- A^{-1} for a symmetric positive definite matrix (identity padded), la *= det A
template <int NT>
MFGM_DEV void sweep_inv(Mat<NT>& A, const LaneId& L, LogAcc& la, int& bad) {
    const int cl = L.c & 3;
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cb = 4 * k;
            const double pr = A.t[J][J].r[k];          // lane (g, c): row 16 J + 4 k + g, column 16 J + c
            Sym4 D;
            D.a00 = bcast(pr, cb);
            D.a10 = bcast(pr, 16 | cb); D.a11 = bcast(pr, 16 | (cb + 1));
            D.a20 = bcast(pr, 32 | cb); D.a21 = bcast(pr, 32 | (cb + 1)); D.a22 = bcast(pr, 32 | (cb + 2));
            D.a30 = bcast(pr, 48 | cb); D.a31 = bcast(pr, 48 | (cb + 1)); D.a32 = bcast(pr, 48 | (cb + 2)); D.a33 = bcast(pr, 48 | (cb + 3));
            double detP, detS;
            const Sym4 Di = inv4(D, detP, detS, bad);
            la.mul(detP);
            la.mul(detS);
            la.renorm();
            // lane (g, c < 4): Di[g][c]
            const double r0 = (cl == 0) ? Di.a00 : (cl == 1) ? Di.a10 : (cl == 2) ? Di.a20 : Di.a30;
            const double r1 = (cl == 0) ? Di.a10 : (cl == 1) ? Di.a11 : (cl == 2) ? Di.a21 : Di.a31;
            const double r2 = (cl == 0) ? Di.a20 : (cl == 1) ? Di.a21 : (cl == 2) ? Di.a22 : Di.a32;
            const double r3 = (cl == 0) ? Di.a30 : (cl == 1) ? Di.a31 : (cl == 2) ? Di.a32 : Di.a33;
            double e = (L.g == 0) ? r0 : (L.g == 1) ? r1 : (L.g == 2) ? r2 : r3;
            e = (L.c < 4) ? e : 0.0;
            const bool inK = (L.c >> 2) == k;          // a pivot column (of tile column J)
            double Tpp[NT];
#pragma unroll
            for (int Jc = 0; Jc < NT; ++Jc) {
                double bop = A.t[J][Jc].r[k];
                if (Jc == J) bop = inK ? ((cl == L.g) ? 1.0 : 0.0) : bop;      // identity in the pivot columns: T there is D^{-1}
                const Tile t = mfma1(e, bop, tile_zero());
                Tpp[Jc] = (Jc == J && inK) ? -t.r[0] : t.r[0];
            }
#pragma unroll
            for (int I = 0; I < NT; ++I) {
                double aop = -A.t[J][I].r[k];
                if (I == J) aop = inK ? 0.0 : aop;
#pragma unroll
                for (int Jc = 0; Jc < NT; ++Jc) {
                    Tile acc = A.t[I][Jc];
                    if (Jc == J) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc.r[i] = inK ? 0.0 : acc.r[i];
                    }
                    A.t[I][Jc] = mfma1(aop, Tpp[Jc], acc);
                }
            }
#pragma unroll
            for (int Jc = 0; Jc < NT; ++Jc) A.t[J][Jc].r[k] = Tpp[Jc];
        }
    A = mat_neg<NT>(A);
}

// sum_r a_r b_r of two vectors (uniform result)
template <int NT>
MFGM_DEV double vec_dot(const Vec<NT>& x, const Vec<NT>& y, const LaneId& L) {
    double q = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) q = __builtin_fma(x.t[I].r[i], y.t[I].r[i], q);
    q = (L.c == 0) ? q : 0.0;
    return bcast(q, 0) + bcast(q, 16) + bcast(q, 32) + bcast(q, 48);
}

template <int NT>
MFGM_DEV Mat<NT> mat_add(const Mat<NT>& a, const Mat<NT>& b) {
    Mat<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) o.t[I][J].r[i] = a.t[I][J].r[i] + b.t[I][J].r[i];
    return o;
}
template <int NT>
MFGM_DEV Vec<NT> vec_add(const Vec<NT>& a, const Vec<NT>& b) {
    Vec<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.t[I].r[i] = a.t[I].r[i] + b.t[I].r[i];
    return o;
}
template <int NT>
MFGM_DEV Vec<NT> vec_neg(const Vec<NT>& a) {
    Vec<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.t[I].r[i] = -a.t[I].r[i];
    return o;
}

// ---- reduce ------------------------------------------------------------------------------------------------------------------------
template <int NT, bool HAS_RHS, bool HAS_CORR>
static __global__ __launch_bounds__(64) void kmi_reduce(WideArgs a) {
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);
    const int t0 = p * R, len = min(R, n - t0);
    int bad = 0;
    LogAcc la;
    la.init();
    auto ld_F = [&](int t) {
        Mat<NT> F = ld_mat<NT, false, true>(wblk(a.Dg, b, n, t, EF), d, L, a.aD);
        if (HAS_CORR) F = mat_sub<NT>(F, ld_mat<NT, false, false>(wblk(a.Dcorr, b, n, t, EF), d, L, 1.0));
        return F;
    };
    auto ld_h = [&](int t) {
        Vec<NT> h = ld_vec<NT>(wblk(a.rg, b, n, t, d), d, L, a.aR);
        if (HAS_CORR) h = vec_sub<NT>(h, ld_vec<NT>(wblk(a.rcorr, b, n, t, d), d, L, 1.0));
        return h;
    };
    Mat<NT> F = ld_F(t0);
    Mat<NT> W = (p > 0) ? ld_mat<NT, false, false>(wblk(a.Sg, b, n, t0 - 1, EF), d, L, a.aS) : mat_zero<NT>();
    Vec<NT> h = HAS_RHS ? ld_h(t0) : vec_zero<NT>();
    Mat<NT> Racc = mat_zero<NT>();
    Vec<NT> rho = vec_zero<NT>();
    for (int s = 0; s < len - 1; ++s) {
        const int t = t0 + s;
        const Mat<NT> St = ld_mat<NT, true, false>(wblk(a.Sg, b, n, t, EF), d, L, a.aS);      // S^T
        Mat<NT> Fn = ld_F(t + 1);
        Vec<NT> hn = HAS_RHS ? ld_h(t + 1) : vec_zero<NT>();
        sweep_inv<NT>(F, L, la, bad);                          // (the determinant is not an output of this pass)
        const Mat<NT> nSt = mat_neg<NT>(St);
        const Mat<NT> TW = gram<NT>(F, W);                     // F^{-1} W
        {
            const Mat<NT> TS = gram<NT>(F, St);                // F^{-1} S^T
            Fn = gram<NT>(nSt, TS, Fn);                        // F' = D' - S F^{-1} S^T
        }
        Racc = gram<NT>(W, TW, Racc);                          // R += W^T F^{-1} W
        if (HAS_RHS) {
            const Vec<NT> th = gram<NT>(F, h);                 // F^{-1} h
            rho = gram<NT>(W, th, rho);
            hn = gram<NT>(nSt, th, hn);
        }
        W = gram<NT>(nSt, TW);                                 // W' = -S F^{-1} W
        F = Fn;
        h = hn;
    }
    const int un = a.up.n;
    st_mat<NT, false>(wblk(a.uDhat, b, un, p, EF), d, L, F);
    st_vec<NT>(wblk(a.urhat, b, un, p, d), d, L, h);
    if (p == P - 1) {
        st_mat<NT, false>(wblk(a.uRsub, b, un, p, EF), d, L, mat_zero<NT>());
        st_mat<NT, false>(wblk(a.uS, b, un, p, EF), d, L, mat_zero<NT>());
        st_vec<NT>(wblk(a.urho, b, un, p, d), d, L, vec_zero<NT>());
    }
    if (p > 0) {
        st_mat<NT, false>(wblk(a.uS, b, un, p - 1, EF), d, L, W);
        st_mat<NT, false>(wblk(a.uRsub, b, un, p - 1, EF), d, L, Racc);
        st_vec<NT>(wblk(a.urho, b, un, p - 1, d), d, L, rho);
    }
    if (bad && L.lane == 0) atomicMax(a.info, 1);
}

// ---- forward -----------------------------------------------------------------------------------------------------------------------
// Levels above the finest also keep their pivot blocks F_t and right-hand sides h_t (in the level's Sigma / mu arrays, which the
// backward pass fills later): the level below rebuilds the state on its separators from them,  F_a = F~ + R_p,  h_a = h~ + rho_p.
template <int NT, bool HAS_RHS, bool HAS_CORR, bool HAS_UP>
static __global__ __launch_bounds__(64) void kmi_forward(WideArgs a) {
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);
    const int t0 = p * R, len = min(R, n - t0);
    int bad = 0;
    LogAcc la;
    la.init();
    Mat<NT> C = mat_zero<NT>();
    Vec<NT> cv = vec_zero<NT>();
    if (HAS_UP && p > 0) {
        const int un = a.up.n;
        Mat<NT> Fa = ld_mat<NT, false, true>(wblk(a.uSig, b, un, p - 1, EF), d, L, 1.0);
        Fa = mat_add<NT>(Fa, ld_mat<NT, false, false>(wblk(a.uRsub, b, un, p - 1, EF), d, L, 1.0));
        const Mat<NT> Sat = ld_mat<NT, true, false>(wblk(a.Sg, b, n, t0 - 1, EF), d, L, a.aS);
        Vec<NT> ha = vec_zero<NT>();
        if (HAS_RHS) ha = vec_add<NT>(ld_vec<NT>(wblk(a.umu, b, un, p - 1, d), d, L, 1.0), ld_vec<NT>(wblk(a.urho, b, un, p - 1, d), d, L, 1.0));
        sweep_inv<NT>(Fa, L, la, bad);
        la.init();                                             // that node's determinant is counted where it is owned
        const Mat<NT> Ja = gram<NT>(Fa, Sat);
        C = gram<NT>(Sat, Ja);
        const bool keep = (a.store_left && p == a.seg_lo);     // sharded chain: see WideArgs::store_left
        if (keep) {
            st_mat<NT, false>(wblk(a.Lg, b, n, t0 - 1, EF), d, L, Fa);
            st_mat<NT, true>(wblk(a.Gg, b, n, t0 - 1, EF), d, L, Ja);
        }
        if (HAS_RHS) {
            const Vec<NT> za = gram<NT>(Fa, ha);
            if (keep) st_vec<NT>(wblk(a.yg, b, n, t0 - 1, d), d, L, za);
            cv = gram<NT>(Sat, za);
        }
    }
    double quad = 0.0;
    for (int s = 0; s < len; ++s) {
        const int t = t0 + s;
        Mat<NT> F = ld_mat<NT, false, true>(wblk(a.Dg, b, n, t, EF), d, L, a.aD);
        if (HAS_CORR) F = mat_sub<NT>(F, ld_mat<NT, false, false>(wblk(a.Dcorr, b, n, t, EF), d, L, 1.0));
        F = mat_sub<NT>(F, C);
        Vec<NT> h = vec_zero<NT>();
        if (HAS_RHS) {
            h = ld_vec<NT>(wblk(a.rg, b, n, t, d), d, L, a.aR);
            if (HAS_CORR) h = vec_sub<NT>(h, ld_vec<NT>(wblk(a.rcorr, b, n, t, d), d, L, 1.0));
            h = vec_sub<NT>(h, cv);
        }
        const bool has_next = (t + 1 < n);
        const Mat<NT> St = has_next ? ld_mat<NT, true, false>(wblk(a.Sg, b, n, t, EF), d, L, a.aS) : mat_zero<NT>();
        if (a.Sigg) {
            st_mat<NT, false>(wblk(a.Sigg, b, n, t, EF), d, L, F);
            if (HAS_RHS) st_vec<NT>(wblk(a.mug, b, n, t, d), d, L, h);
        }
        sweep_inv<NT>(F, L, la, bad);
        st_mat<NT, false>(wblk(a.Lg, b, n, t, EF), d, L, F);
        const Mat<NT> J = gram<NT>(F, St);                     // F^{-1} S^T
        if (has_next) st_mat<NT, true>(wblk(a.Gg, b, n, t, EF), d, L, J);
        C = gram<NT>(St, J);
        if (HAS_RHS) {
            const Vec<NT> z = gram<NT>(F, h);
            st_vec<NT>(wblk(a.yg, b, n, t, d), d, L, z);
            cv = gram<NT>(St, z);
            quad += vec_dot<NT>(h, z, L);
        }
    }
    if (a.part && L.lane == 0) {
        a.part[b * P + p] = 0.5 * la.value();
        a.part[a.lv.Lpad + b * P + p] = quad;
    }
    if (bad && L.lane == 0) atomicMax(a.info, 1);
}

// ---- backward ----------------------------------------------------------------------------------------------------------------------
template <int NT, bool HAS_RHS, bool HAS_UP, bool WANT_SUB>
static __global__ __launch_bounds__(64) void kmi_backward(WideArgs a) {
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);
    const int t0 = p * R, len = min(R, n - t0), te = t0 + len - 1;
    Mat<NT> Sn;
    Vec<NT> xn = vec_zero<NT>();
    if (HAS_UP) {
        Sn = ld_mat<NT, false, false>(wblk(a.uSig, b, a.up.n, p, EF), d, L, 1.0);
        if (HAS_RHS) xn = ld_vec<NT>(wblk(a.umu, b, a.up.n, p, d), d, L, 1.0);
    } else {
        Sn = ld_mat<NT, false, false>(wblk(a.Lg, b, n, te, EF), d, L, 1.0);
        if (HAS_RHS) xn = ld_vec<NT>(wblk(a.yg, b, n, te, d), d, L, 1.0);
    }
    st_mat<NT, false>(wblk(a.Sigg, b, n, te, EF), d, L, Sn);
    if (HAS_RHS) st_vec<NT>(wblk(a.mug, b, n, te, d), d, L, xn);
    auto step = [&](int t, bool write_node) {
        const Mat<NT> Jt = ld_mat<NT, false, false>(wblk(a.Gg, b, n, t, EF), d, L, 1.0);      // S F^{-1}
        Mat<NT> Fi = mat_zero<NT>();
        Vec<NT> z = vec_zero<NT>();
        if (write_node) {
            Fi = ld_mat<NT, false, false>(wblk(a.Lg, b, n, t, EF), d, L, 1.0);
            if (HAS_RHS) z = ld_vec<NT>(wblk(a.yg, b, n, t, d), d, L, 1.0);
        }
        const Mat<NT> U = gram<NT>(Sn, Jt);                    // Sigma_n S F^{-1}
        if (WANT_SUB) st_mat<NT, false>(wblk(a.Subg, b, n, t, EF), d, L, mat_neg<NT>(U));
        if (!write_node) return;
        const Mat<NT> Sig = gram<NT>(Jt, U, Fi);
        st_mat<NT, false>(wblk(a.Sigg, b, n, t, EF), d, L, Sig);
        if (HAS_RHS) {
            xn = gram<NT>(mat_neg<NT>(Jt), xn, z);
            st_vec<NT>(wblk(a.mug, b, n, t, d), d, L, xn);
        }
        Sn = Sig;
    };
    for (int s = len - 2; s >= 0; --s) step(t0 + s, true);
    if (WANT_SUB && p > 0) step(t0 - 1, false);
}

}  // namespace mfgm
