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
// and after the four sweeps A = -F^{-1}.  The only cross-lane traffic is the read of the 10 distinct entries of D (v_readlane) and a
// 16-double LDS table that puts D^{-1} into the lanes (g, c < 4); D^{-1} is formed redundantly by every lane from two 2 x 2
// determinants, whose product also accumulates log|F|.
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
MFGM_DEV void sweep_inv(Mat<NT>& A, const LaneId& L, LogAcc& la, int& bad, double* lds) {
    const int cl = L.c & 3;
    const int ehi = max(L.g, cl), elo = min(L.g, cl), eidx = ehi * (ehi + 1) / 2 + elo;      // packed lower triangle
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
            // lane (g, c < 4): Di[g][c] -- through a 16-double table in LDS (one lane writes it, every lane reads its entry), which
            // costs a third of the instructions of selecting among the 10 uniform values lane by lane
            if (L.lane == 0) {
                lds[0] = Di.a00; lds[1] = Di.a10; lds[2] = Di.a11; lds[3] = Di.a20; lds[4] = Di.a21;
                lds[5] = Di.a22; lds[6] = Di.a30; lds[7] = Di.a31; lds[8] = Di.a32; lds[9] = Di.a33;
            }
            __builtin_amdgcn_wave_barrier();
            double e = lds[eidx];
            __builtin_amdgcn_wave_barrier();
            e = (L.c < 4) ? e : 0.0;
            const bool inK = (L.c >> 2) == k;          // a pivot column (of tile column J)
            const double eye = (inK && cl == L.g) ? 1.0 : 0.0;
            double Tpp[NT];
#pragma unroll
            for (int Jc = 0; Jc < NT; ++Jc) {
                double bop = A.t[J][Jc].r[k];
                if (Jc == J) bop = inK ? eye : bop;                            // identity in the pivot columns: T there is D^{-1}
                const Tile t = mfma1(e, bop, tile_zero());
                Tpp[Jc] = (Jc == J && inK) ? -t.r[0] : t.r[0];
            }
            // A <- A - P^T T''; in the pivot columns of the other rows the old entries are REPLACED by P^T D^{-1}, so the accumulator is
            // zeroed there.  (Subtracting P^T (T'' + I) instead, which is the same thing for an exactly symmetric A, leaves the rounding
            // asymmetry of A behind at the scale of A, and T is smaller than A by the size of the pivot block: not done.)
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

// transpose through LDS (one 16 x 16 tile at a time, 17-double row stride); one wavefront, LDS accesses in program order
template <int NT>
MFGM_DEV Mat<NT> mat_transpose_w(const Mat<NT>& m, double* lds, const LaneId& L) {
    Mat<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[(L.g + 4 * i) * 17 + L.c] = m.t[I][J].r[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) o.t[J][I].r[i] = lds[L.c * 17 + L.g + 4 * i];
            __builtin_amdgcn_wave_barrier();
        }
    return o;
}

// ---- vectors on the vector ALU -------------------------------------------------------------------------------------------------------
// A d-vector next to accumulator-layout tiles comes in two shapes: by COLUMN (lane (g, c) holds v[16 J + c], the same in the four
// 16-lane rows) and by ROW (lane (g, c) holds v[16 I + g + 4 i], i = 0..3, the same in the 16 lanes of a row).  M^T x of a tile matrix
// with x by row is four FMAs per tile and lane plus a sum over the four 16-lane rows (v_permlane32_swap / v_permlane16_swap), and
// comes out by column; the pivot inverses and Sigma are symmetric, so every matrix-vector product of the sweeps has this shape.
// Column -> row goes through 16 NT doubles of LDS.  (As tile columns of an MFMA the same products cost four MFMAs each.)
template <int NT>
struct RowVec {
    double r[NT][4];
};
template <int NT>
struct ColVec {
    double c[NT];
};

MFGM_DEV double sum_rows(double x) {
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double u = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(u);
    hi = __double2hiint(u);
    const auto c = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto e = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(e[0], c[0]) + __hiloint2double(e[1], c[1]);
}

// acc + sign * M^T x
template <int NT>
MFGM_DEV ColVec<NT> tmatvec(const Mat<NT>& M, const RowVec<NT>& x, const ColVec<NT>& acc, double sign) {
    ColVec<NT> y;
#pragma unroll
    for (int J = 0; J < NT; ++J) {
        double p = 0.0;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int i = 0; i < 4; ++i) p = __builtin_fma(M.t[I][J].r[i], x.r[I][i], p);
        y.c[J] = __builtin_fma(sign, sum_rows(p), acc.c[J]);
    }
    return y;
}
template <int NT>
MFGM_DEV ColVec<NT> col_zero() {
    ColVec<NT> v;
#pragma unroll
    for (int J = 0; J < NT; ++J) v.c[J] = 0.0;
    return v;
}
template <int NT>
MFGM_DEV ColVec<NT> col_sub(const ColVec<NT>& a, const ColVec<NT>& b) {
    ColVec<NT> v;
#pragma unroll
    for (int J = 0; J < NT; ++J) v.c[J] = a.c[J] - b.c[J];
    return v;
}
template <int NT>
MFGM_DEV ColVec<NT> col_add(const ColVec<NT>& a, const ColVec<NT>& b) {
    ColVec<NT> v;
#pragma unroll
    for (int J = 0; J < NT; ++J) v.c[J] = a.c[J] + b.c[J];
    return v;
}
// lds: 16 NT doubles; the workgroup is one wavefront, whose LDS accesses execute in program order
template <int NT>
MFGM_DEV RowVec<NT> to_row(const ColVec<NT>& v, double* lds, const LaneId& L) {
    if (L.g == 0) {
#pragma unroll
        for (int J = 0; J < NT; ++J) lds[16 * J + L.c] = v.c[J];
    }
    __builtin_amdgcn_wave_barrier();
    RowVec<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.r[I][i] = lds[16 * I + L.g + 4 * i];
    __builtin_amdgcn_wave_barrier();
    return o;
}
template <int NT>
MFGM_DEV ColVec<NT> ld_col(const double* __restrict__ v, int d, const LaneId& L, double scale) {
    ColVec<NT> o;
    double x[NT];
#pragma unroll
    for (int J = 0; J < NT; ++J) x[J] = v[(16 * J + L.c < d) ? 16 * J + L.c : 0];
#pragma unroll
    for (int J = 0; J < NT; ++J) o.c[J] = x[J] * ((16 * J + L.c < d) ? scale : 0.0);
    return o;
}
template <int NT>
MFGM_DEV void st_col(double* __restrict__ v, int d, const LaneId& L, const ColVec<NT>& a) {
#pragma unroll
    for (int J = 0; J < NT; ++J)
        if (L.g == 0 && 16 * J + L.c < d) v[16 * J + L.c] = a.c[J];
}

// d x d sub-block (row0, col0) of a row-major matrix with leading dimension ld, times scale, as tiles (zero padded; TRANSPOSED: its
// transpose)
template <int NT, bool TRANSPOSED>
MFGM_DEV Mat<NT> ld_sub(const double* __restrict__ base, int ld, int row0, int col0, int d, const LaneId& L, double scale) {
    Mat<NT> m;
    double x[NT][NT][4];
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * I + L.g + 4 * i, c = 16 * J + L.c;
                const bool ok = r < d && c < d;
                x[I][J][i] = base[ok ? (TRANSPOSED ? (row0 + c) * ld + col0 + r : (row0 + r) * ld + col0 + c) : 0];
            }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * I + L.g + 4 * i, c = 16 * J + L.c;
                m.t[I][J].r[i] = x[I][J][i] * ((r < d && c < d) ? scale : 0.0);
            }
    return m;
}

// Loads whose values are consumed later: the raw doubles of a (sub-)block / vector now, scale and padding when they are needed
// (fix_*), so that a pass can request the next node's inputs before the pivot inverse of the current one and touch them after it --
// scaling inside the load makes the compiler wait for the data where the load was issued.
template <int NT>
struct RawMat {
    double x[NT][NT][4];
};
template <int NT>
struct RawCol {
    double x[NT];
};
template <int NT, bool TRANSPOSED>
MFGM_DEV RawMat<NT> ld_raw(const double* __restrict__ base, int ld, int row0, int col0, int d, const LaneId& L) {
    RawMat<NT> m;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * I + L.g + 4 * i, c = 16 * J + L.c;
                const bool ok = r < d && c < d;
                m.x[I][J][i] = base[ok ? (TRANSPOSED ? (row0 + c) * ld + col0 + r : (row0 + r) * ld + col0 + c) : 0];
            }
    return m;
}
// quadrant `quad` (0: upper left = first state of the pair, 1: lower left = second x first, 2: lower right = second state) of site m,
// from either layout of the site tensor (WideArgs::site_packed)
template <int NT, bool TRANSPOSED>
MFGM_DEV RawMat<NT> ld_site_raw(const WideArgs& a, int m, int quad, const LaneId& L) {
    const int d = a.d;
    if (!a.site_packed) return ld_raw<NT, TRANSPOSED>(a.site2 + (size_t)m * (4 * d * d), 2 * d, quad == 0 ? 0 : d, quad == 2 ? d : 0, d, L);
    const int ET = d * (d + 1) / 2, EF = d * d;
    const double* base = a.site2 + (size_t)m * (2 * ET + EF) + (quad == 0 ? 0 : (quad == 1 ? ET : ET + EF));
    RawMat<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * I + L.g + 4 * i, c = 16 * J + L.c;
                const bool ok = r < d && c < d;
                const int hi = r > c ? r : c, lo = r > c ? c : r;
                const int off = (quad == 1) ? (TRANSPOSED ? c * d + r : r * d + c) : hi * (hi + 1) / 2 + lo;
                o.x[I][J][i] = base[ok ? off : 0];
            }
    return o;
}
// acc + scale * raw (zero outside d x d; PAD_EYE: plus an identity on the padded diagonal)
template <int NT, bool PAD_EYE>
MFGM_DEV Mat<NT> fix_mat(const RawMat<NT>& raw, int d, const LaneId& L, double scale, const Mat<NT>& acc) {
    Mat<NT> m;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * I + L.g + 4 * i, c = 16 * J + L.c;
                const bool ok = r < d && c < d;
                m.t[I][J].r[i] = __builtin_fma(raw.x[I][J][i], ok ? scale : 0.0, acc.t[I][J].r[i] + ((!ok && PAD_EYE && r == c) ? 1.0 : 0.0));
            }
    return m;
}
template <int NT>
MFGM_DEV RawCol<NT> ld_col_raw(const double* __restrict__ v, int d, const LaneId& L) {
    RawCol<NT> o;
#pragma unroll
    for (int J = 0; J < NT; ++J) o.x[J] = v[(16 * J + L.c < d) ? 16 * J + L.c : 0];
    return o;
}
template <int NT>
MFGM_DEV ColVec<NT> fix_col(const RawCol<NT>& raw, int d, const LaneId& L, double scale, const ColVec<NT>& acc) {
    ColVec<NT> o;
#pragma unroll
    for (int J = 0; J < NT; ++J) o.c[J] = __builtin_fma(raw.x[J], (16 * J + L.c < d) ? scale : 0.0, acc.c[J]);
    return o;
}

// The inputs of one node of a pass (diagonal block, its Gram correction, right-hand side and its correction, and -- sparse-CVI
// inputs -- the site quadrants / halves that are overlap-added), raw.  HAS_S: also the sub-diagonal block S_t (TRANSPOSED_S: as S_t^T).
template <int NT>
struct RawNode {
    RawMat<NT> D, Dc, Shi, Slo, S, Ss;
    RawCol<NT> r, rc, s1, s2;
};
template <int NT, bool HAS_RHS, bool HAS_CORR, bool SITES, bool TRANSPOSED_S>
MFGM_DEV void ld_node_raw(const WideArgs& a, int b, int t, bool want_S, const LaneId& L, RawNode<NT>& o) {
    const int d = a.d, EF = d * d, d2 = 2 * d, n = a.lv.n;
    o.D = ld_raw<NT, false>(wblk(a.Dg, b, n, t, EF), d, 0, 0, d, L);
    if (HAS_CORR) o.Dc = ld_raw<NT, false>(wblk(a.Dcorr, b, n, t, EF), d, 0, 0, d, L);
    if (SITES) {
        o.Shi = ld_site_raw<NT, false>(a, t + 1, 0, L);
        o.Slo = ld_site_raw<NT, false>(a, t, 2, L);
    }
    if (want_S) {
        o.S = ld_raw<NT, TRANSPOSED_S>(wblk(a.Sg, b, n, t, EF), d, 0, 0, d, L);
        if (SITES) o.Ss = ld_site_raw<NT, TRANSPOSED_S>(a, t + 1, 1, L);
    }
    if (HAS_RHS) {
        if (a.rg) o.r = ld_col_raw<NT>(wblk(a.rg, b, n, t, d), d, L);
        if (HAS_CORR) o.rc = ld_col_raw<NT>(wblk(a.rcorr, b, n, t, d), d, L);
        if (SITES) {
            o.s1 = ld_col_raw<NT>(a.site1 + (size_t)(t + 1) * d2, d, L);
            o.s2 = ld_col_raw<NT>(a.site1 + (size_t)t * d2 + d, d, L);
        }
    }
}
template <int NT, bool HAS_CORR, bool SITES>
MFGM_DEV Mat<NT> node_F(const WideArgs& a, const RawNode<NT>& o, const LaneId& L) {
    Mat<NT> F = fix_mat<NT, true>(o.D, a.d, L, a.aD, mat_zero<NT>());
    if (HAS_CORR) F = fix_mat<NT, false>(o.Dc, a.d, L, -1.0, F);
    if (SITES) F = fix_mat<NT, false>(o.Slo, a.d, L, a.aD, fix_mat<NT, false>(o.Shi, a.d, L, a.aD, F));
    return F;
}
template <int NT, bool SITES>
MFGM_DEV Mat<NT> node_S(const WideArgs& a, const RawNode<NT>& o, const LaneId& L) {
    Mat<NT> S = fix_mat<NT, false>(o.S, a.d, L, a.aS, mat_zero<NT>());
    if (SITES) S = fix_mat<NT, false>(o.Ss, a.d, L, 2.0 * a.aS, S);
    return S;
}
template <int NT, bool HAS_RHS, bool HAS_CORR, bool SITES>
MFGM_DEV ColVec<NT> node_h(const WideArgs& a, const RawNode<NT>& o, const LaneId& L) {
    ColVec<NT> h = col_zero<NT>();
    if (HAS_RHS) {
        if (a.rg) h = fix_col<NT>(o.r, a.d, L, a.aR, h);
        if (HAS_CORR) h = fix_col<NT>(o.rc, a.d, L, -1.0, h);
        if (SITES) h = fix_col<NT>(o.s2, a.d, L, a.aR, fix_col<NT>(o.s1, a.d, L, a.aR, h));
    }
    return h;
}

// Posterior naturals of the sparse-CVI model at node t, formed on load (WideArgs::site1 / site2): site t + 1 has state t as the first
// of its pair, site t as the second
template <int NT>
MFGM_DEV Mat<NT> site_diag(const WideArgs& a, int t, const LaneId& L) {
    const Mat<NT> hi = fix_mat<NT, false>(ld_site_raw<NT, false>(a, t + 1, 0, L), a.d, L, a.aD, mat_zero<NT>());
    return fix_mat<NT, false>(ld_site_raw<NT, false>(a, t, 2, L), a.d, L, a.aD, hi);
}
template <int NT, bool TRANSPOSED>
MFGM_DEV Mat<NT> site_sub(const WideArgs& a, int t, const LaneId& L) {
    return fix_mat<NT, false>(ld_site_raw<NT, TRANSPOSED>(a, t + 1, 1, L), a.d, L, 2.0 * a.aS, mat_zero<NT>());
}
template <int NT>
MFGM_DEV ColVec<NT> site_lin(const WideArgs& a, int t, const LaneId& L) {
    const int d = a.d, d2 = 2 * d;
    return col_add<NT>(ld_col<NT>(a.site1 + (size_t)(t + 1) * d2, d, L, a.aR), ld_col<NT>(a.site1 + (size_t)t * d2 + d, d, L, a.aR));
}

// ---- reduce ------------------------------------------------------------------------------------------------------------------------
template <int NT, bool HAS_RHS, bool HAS_CORR, bool SITES = false>
static __global__ __launch_bounds__(64) void kmi_reduce(WideArgs a) {
    __shared__ double lds[16 * NT];
    __shared__ double ldsE[16];
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
        if (SITES) F = mat_add<NT>(F, site_diag<NT>(a, t, L));
        return F;
    };
    auto ld_h = [&](int t) {
        ColVec<NT> h = a.rg ? ld_col<NT>(wblk(a.rg, b, n, t, d), d, L, a.aR) : col_zero<NT>();
        if (HAS_CORR) h = col_sub<NT>(h, ld_col<NT>(wblk(a.rcorr, b, n, t, d), d, L, 1.0));
        if (SITES) h = col_add<NT>(h, site_lin<NT>(a, t, L));
        return h;
    };
    auto ld_S = [&](int t) {            // S_t
        Mat<NT> S = ld_mat<NT, false, false>(wblk(a.Sg, b, n, t, EF), d, L, a.aS);
        if (SITES) S = mat_add<NT>(S, site_sub<NT, false>(a, t, L));
        return S;
    };
    Mat<NT> F = ld_F(t0);
    Mat<NT> W = (p > 0) ? ld_S(t0 - 1) : mat_zero<NT>();
    ColVec<NT> hc = HAS_RHS ? ld_h(t0) : col_zero<NT>();
    Mat<NT> Racc = mat_zero<NT>();
    ColVec<NT> rho = col_zero<NT>();
    for (int s = 0; s < len - 1; ++s) {
        const int t = t0 + s;
        // the next node's inputs and S_t^T are requested now and touched after the pivot inverse, which covers their latency
        RawNode<NT> nx;
        ld_node_raw<NT, HAS_RHS, HAS_CORR, SITES, true>(a, b, t + 1, false, L, nx);
        nx.S = ld_raw<NT, true>(wblk(a.Sg, b, n, t, EF), d, 0, 0, d, L);
        if (SITES) nx.Ss = ld_site_raw<NT, true>(a, t + 1, 1, L);
        sweep_inv<NT>(F, L, la, bad, ldsE);                          // (the determinant is not an output of this pass)
        __builtin_amdgcn_sched_barrier(0);
        const Mat<NT> St = node_S<NT, SITES>(a, nx, L);
        Mat<NT> Fn = node_F<NT, HAS_CORR, SITES>(a, nx, L);
        ColVec<NT> hn = node_h<NT, HAS_RHS, HAS_CORR, SITES>(a, nx, L);
        const Mat<NT> nSt = mat_neg<NT>(St);
        const Mat<NT> TW = gram<NT>(F, W);                     // F^{-1} W
        {
            const Mat<NT> TS = gram<NT>(F, St);                // F^{-1} S^T
            Fn = gram<NT>(nSt, TS, Fn);                        // F' = D' - S F^{-1} S^T
        }
        Racc = gram<NT>(W, TW, Racc);                          // R += W^T F^{-1} W
        if (HAS_RHS) {
            const RowVec<NT> th = to_row<NT>(tmatvec<NT>(F, to_row<NT>(hc, lds, L), col_zero<NT>(), 1.0), lds, L);      // F^{-1} h
            rho = tmatvec<NT>(W, th, rho, 1.0);
            hn = tmatvec<NT>(St, th, hn, -1.0);
        }
        W = gram<NT>(nSt, TW);                                 // W' = -S F^{-1} W
        F = Fn;
        hc = hn;
    }
    const int un = a.up.n;
    st_mat<NT, false>(wblk(a.uDhat, b, un, p, EF), d, L, F);
    st_col<NT>(wblk(a.urhat, b, un, p, d), d, L, hc);
    if (p == P - 1) {
        st_mat<NT, false>(wblk(a.uRsub, b, un, p, EF), d, L, mat_zero<NT>());
        st_mat<NT, false>(wblk(a.uS, b, un, p, EF), d, L, mat_zero<NT>());
        st_col<NT>(wblk(a.urho, b, un, p, d), d, L, col_zero<NT>());
    }
    if (p > 0) {
        st_mat<NT, false>(wblk(a.uS, b, un, p - 1, EF), d, L, W);
        st_mat<NT, false>(wblk(a.uRsub, b, un, p - 1, EF), d, L, Racc);
        st_col<NT>(wblk(a.urho, b, un, p - 1, d), d, L, rho);
    }
    if (bad && L.lane == 0) flag_not_pd(a.info, a.lv.level, b * a.lv.P + p);
}

// ---- forward -----------------------------------------------------------------------------------------------------------------------
// Levels above the finest also keep their pivot blocks F_t and right-hand sides h_t (in the level's Sigma / mu arrays, which the
// backward pass fills later): the level below rebuilds the state on its separators from them,  F_a = F~ + R_p,  h_a = h~ + rho_p.
template <int NT, bool HAS_RHS, bool HAS_CORR, bool HAS_UP, bool SITES = false>
static __global__ __launch_bounds__(64) void kmi_forward(WideArgs a) {
    __shared__ double lds[16 * NT];
    __shared__ double ldsT[16 * 17];
    __shared__ double ldsE[16];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);
    const int t0 = p * R, len = min(R, n - t0);
    int bad = 0;
    LogAcc la;
    la.init();
    Mat<NT> C = mat_zero<NT>();
    ColVec<NT> cv = col_zero<NT>();
    if (HAS_UP && p > 0) {
        const int un = a.up.n;
        Mat<NT> Fa = ld_mat<NT, false, true>(wblk(a.uSig, b, un, p - 1, EF), d, L, 1.0);
        Fa = mat_add<NT>(Fa, ld_mat<NT, false, false>(wblk(a.uRsub, b, un, p - 1, EF), d, L, 1.0));
        Mat<NT> Sat = ld_mat<NT, true, false>(wblk(a.Sg, b, n, t0 - 1, EF), d, L, a.aS);
        if (SITES) Sat = mat_add<NT>(Sat, site_sub<NT, true>(a, t0 - 1, L));
        ColVec<NT> ha = col_zero<NT>();
        if (HAS_RHS) ha = col_add<NT>(ld_col<NT>(wblk(a.umu, b, un, p - 1, d), d, L, 1.0), ld_col<NT>(wblk(a.urho, b, un, p - 1, d), d, L, 1.0));
        const bool keep = (a.store_left && p == a.seg_lo);     // sharded chain: see WideArgs::store_left
        if (keep && a.Sigg) {                                  // ... and the level below rebuilds its own boundary from this node
            st_mat<NT, false>(wblk(a.Sigg, b, n, t0 - 1, EF), d, L, Fa);
            if (HAS_RHS) st_col<NT>(wblk(a.mug, b, n, t0 - 1, d), d, L, ha);
        }
        sweep_inv<NT>(Fa, L, la, bad, ldsE);
        la.init();                                             // that node's determinant is counted where it is owned
        const Mat<NT> Ja = gram<NT>(Fa, Sat);
        C = gram<NT>(Sat, Ja);
        if (keep) {
            st_mat<NT, false>(wblk(a.Lg, b, n, t0 - 1, EF), d, L, Fa);
            st_mat<NT, false>(wblk(a.Gg, b, n, t0 - 1, EF), d, L, mat_transpose_w<NT>(Ja, ldsT, L));
        }
        if (HAS_RHS) {
            const ColVec<NT> za = tmatvec<NT>(Fa, to_row<NT>(ha, lds, L), col_zero<NT>(), 1.0);
            if (keep) st_col<NT>(wblk(a.yg, b, n, t0 - 1, d), d, L, za);
            cv = tmatvec<NT>(Sat, to_row<NT>(za, lds, L), col_zero<NT>(), 1.0);
        }
    }
    double quad = 0.0;                       // per lane: sum over the nodes of h_c z_c for the lane's column
    // the inputs of a node are requested one node ahead (raw) and touched at the top of its own step: the pivot inverse and the products
    // of the current node cover their latency
    RawNode<NT> nx;
    ld_node_raw<NT, HAS_RHS, HAS_CORR, SITES, true>(a, b, t0, t0 + 1 < n, L, nx);
    for (int s = 0; s < len; ++s) {
        const int t = t0 + s;
        const bool has_next = (t + 1 < n);
        Mat<NT> F = mat_sub<NT>(node_F<NT, HAS_CORR, SITES>(a, nx, L), C);
        const ColVec<NT> h = col_sub<NT>(node_h<NT, HAS_RHS, HAS_CORR, SITES>(a, nx, L), cv);
        const Mat<NT> St = has_next ? node_S<NT, SITES>(a, nx, L) : mat_zero<NT>();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < len) ld_node_raw<NT, HAS_RHS, HAS_CORR, SITES, true>(a, b, t + 1, t + 2 < n, L, nx);
        if (a.Sigg) {
            st_mat<NT, false>(wblk(a.Sigg, b, n, t, EF), d, L, F);
            if (HAS_RHS) st_col<NT>(wblk(a.mug, b, n, t, d), d, L, h);
        }
        sweep_inv<NT>(F, L, la, bad, ldsE);
        st_mat<NT, false>(wblk(a.Lg, b, n, t, EF), d, L, F);
        const Mat<NT> J = gram<NT>(F, St);                     // F^{-1} S^T
        if (has_next) st_mat<NT, false>(wblk(a.Gg, b, n, t, EF), d, L, mat_transpose_w<NT>(J, ldsT, L));     // S F^{-1}, rows contiguous
        C = gram<NT>(St, J);
        if (HAS_RHS) {
            const ColVec<NT> z = tmatvec<NT>(F, to_row<NT>(h, lds, L), col_zero<NT>(), 1.0);
            st_col<NT>(wblk(a.yg, b, n, t, d), d, L, z);
            cv = tmatvec<NT>(St, to_row<NT>(z, lds, L), col_zero<NT>(), 1.0);
#pragma unroll
            for (int J = 0; J < NT; ++J) quad = __builtin_fma(h.c[J], z.c[J], quad);
        }
    }
    if (a.part) {
        quad = (L.g == 0) ? quad : 0.0;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) quad += __shfl_xor(quad, off, 64);
        if (L.lane == 0) {
            a.part[b * P + p] = 0.5 * la.value();
            a.part[a.lv.Lpad + b * P + p] = quad;
        }
    }
    if (bad && L.lane == 0) flag_not_pd(a.info, a.lv.level, b * a.lv.P + p);
}

// ---- backward ----------------------------------------------------------------------------------------------------------------------
template <int NT, bool HAS_RHS, bool HAS_UP, bool WANT_SUB>
static __global__ __launch_bounds__(64) void kmi_backward(WideArgs a) {
    __shared__ double lds[16 * NT];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);
    const int t0 = p * R, len = min(R, n - t0), te = t0 + len - 1;
    Mat<NT> Sn;
    ColVec<NT> xn = col_zero<NT>();
    if (HAS_UP) {
        Sn = ld_mat<NT, false, false>(wblk(a.uSig, b, a.up.n, p, EF), d, L, 1.0);
        if (HAS_RHS) xn = ld_col<NT>(wblk(a.umu, b, a.up.n, p, d), d, L, 1.0);
    } else {
        Sn = ld_mat<NT, false, false>(wblk(a.Lg, b, n, te, EF), d, L, 1.0);
        if (HAS_RHS) xn = ld_col<NT>(wblk(a.yg, b, n, te, d), d, L, 1.0);
    }
    st_mat<NT, false>(wblk(a.Sigg, b, n, te, EF), d, L, Sn);
    if (HAS_RHS) st_col<NT>(wblk(a.mug, b, n, te, d), d, L, xn);
    auto step = [&](int t, bool write_node) {
        const Mat<NT> Jt = ld_mat<NT, false, false>(wblk(a.Gg, b, n, t, EF), d, L, 1.0);      // S F^{-1}
        Mat<NT> Fi = mat_zero<NT>();
        ColVec<NT> z = col_zero<NT>();
        if (write_node) {
            Fi = ld_mat<NT, false, false>(wblk(a.Lg, b, n, t, EF), d, L, 1.0);
            if (HAS_RHS) z = ld_col<NT>(wblk(a.yg, b, n, t, d), d, L, 1.0);
        }
        const Mat<NT> U = gram<NT>(Sn, Jt);                    // Sigma_n S F^{-1}
        if (WANT_SUB) st_mat<NT, false>(wblk(a.Subg, b, n, t, EF), d, L, mat_neg<NT>(U));
        if (!write_node) return;
        const Mat<NT> Sig = gram<NT>(Jt, U, Fi);
        st_mat<NT, false>(wblk(a.Sigg, b, n, t, EF), d, L, Sig);
        if (HAS_RHS) {
            xn = tmatvec<NT>(Jt, to_row<NT>(xn, lds, L), z, -1.0);
            st_col<NT>(wblk(a.mug, b, n, t, d), d, L, xn);
        }
        Sn = Sig;
    };
    for (int s = len - 2; s >= 0; --s) step(t0 + s, true);
    if (WANT_SUB && p > 0) step(t0 - 1, false);
}

}  // namespace mfgm
