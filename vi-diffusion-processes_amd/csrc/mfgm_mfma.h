// MFMA sweeps for block sizes 8 < d <= 32: one wavefront per chain segment, every d x d block held as NT x NT tiles of 16 x 16
// (NT = 1 for d <= 16, NT = 2 for d <= 32) in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane (g = lane>>4, c = lane&15),
// register i  <->  element (row g + 4i, col c) of a tile; blocks are padded in registers with an identity diagonal).
//
// The f64 MFMA takes A[m = lane&15][k = lane>>4] and B[k = lane>>4][n = lane&15]; feeding register i of two tiles X and Y that
// are both in accumulator layout as the A and B operands of four chained MFMAs (i = 0..3) sums over k = g + 4i and yields
//
//        gram(X, Y) = X^T Y        in accumulator layout again, with no cross-lane movement at all
//
// (for tiled matrices (X^T Y)_{IJ} = sum_K gram(X_{KI}, Y_{KJ})).  The three passes of the partitioned solver (mfgm_sweeps.h /
// mfgm_wide.h: reduce, forward, backward; same level recursion, same natural-layout arrays and outputs as the wide path) are
// therefore written with Gram products only:
//   forward / reduce :  Xt = L^{-T};  G^T = gram(Xt, S^T),  y = gram(Xt, h),  G G^T = gram(G^T, G^T),  G y = gram(G^T, y),
//                       spike  V = gram(Xt, W),  R += gram(V, V),  rho += gram(V, y),  W' = -gram(G^T, V)
//   backward         :  X = L^{-1};   H = G X = gram(G^T, X),  Sn H = gram(Sn, H),  Sigma = gram(X, X) + gram(H, Sn H),
//                       x = gram(X, y - gram(G, x_n))
// S^T / G^T are obtained by loading the block transposed from memory; vectors ride in column 0 of a tile column.  The only part
// that is not a matrix product is the pivot-by-pivot Cholesky / triangular inversion, done as Gauss-Jordan row operations on
// [F | I] with ds_bpermute row / column broadcasts (gj), and one LDS transpose of L^{-1} per node in reduce / forward.
#pragma once
#include "mfgm_layout.h"
#include "mfgm_math.h"
#include "mfgm_wide.h"   // WideArgs, wblk, bcast

namespace mfgm {

typedef double v4d __attribute__((ext_vector_type(4)));

struct Tile {
    double r[4];
};

struct LaneId {
    int lane, g, c;
};

template <int NT>
struct Mat {            // NT x NT tiles: block (I, J) holds rows 16 I .., columns 16 J ..
    Tile t[NT][NT];
};
template <int NT>
struct Vec {            // a vector: column 0 of the tile column, block row I
    Tile t[NT];
};

MFGM_DEV Tile tile_zero() { return Tile{{0.0, 0.0, 0.0, 0.0}}; }

// acc + X^T Y for single tiles
MFGM_DEV Tile gram(const Tile& X, const Tile& Y, const Tile& acc) {
    v4d a = {acc.r[0], acc.r[1], acc.r[2], acc.r[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) a = __builtin_amdgcn_mfma_f64_16x16x4f64(X.r[i], Y.r[i], a, 0, 0, 0);
    return Tile{{a[0], a[1], a[2], a[3]}};
}

template <int NT>
MFGM_DEV Mat<NT> mat_zero() {
    Mat<NT> m;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) m.t[I][J] = tile_zero();
    return m;
}
template <int NT>
MFGM_DEV Vec<NT> vec_zero() {
    Vec<NT> v;
#pragma unroll
    for (int I = 0; I < NT; ++I) v.t[I] = tile_zero();
    return v;
}
template <int NT>
MFGM_DEV Mat<NT> mat_eye(const LaneId& L) {
    Mat<NT> m = mat_zero<NT>();
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) m.t[I][I].r[i] = (L.g + 4 * i == L.c) ? 1.0 : 0.0;
    return m;
}
template <int NT>
MFGM_DEV Mat<NT> mat_neg(const Mat<NT>& a) {
    Mat<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) o.t[I][J].r[i] = -a.t[I][J].r[i];
    return o;
}
template <int NT>
MFGM_DEV Mat<NT> mat_sub(const Mat<NT>& a, const Mat<NT>& b) {
    Mat<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) o.t[I][J].r[i] = a.t[I][J].r[i] - b.t[I][J].r[i];
    return o;
}
template <int NT>
MFGM_DEV Vec<NT> vec_sub(const Vec<NT>& a, const Vec<NT>& b) {
    Vec<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.t[I].r[i] = a.t[I].r[i] - b.t[I].r[i];
    return o;
}

// acc + X^T Y  (matrix x matrix, matrix x vector)
template <int NT>
MFGM_DEV Mat<NT> gram(const Mat<NT>& X, const Mat<NT>& Y, const Mat<NT>& acc) {
    Mat<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            Tile a = acc.t[I][J];
#pragma unroll
            for (int K = 0; K < NT; ++K) a = gram(X.t[K][I], Y.t[K][J], a);
            o.t[I][J] = a;
        }
    return o;
}
template <int NT>
MFGM_DEV Mat<NT> gram(const Mat<NT>& X, const Mat<NT>& Y) { return gram<NT>(X, Y, mat_zero<NT>()); }
template <int NT>
MFGM_DEV Vec<NT> gram(const Mat<NT>& X, const Vec<NT>& y, const Vec<NT>& acc) {
    Vec<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I) {
        Tile a = acc.t[I];
#pragma unroll
        for (int K = 0; K < NT; ++K) a = gram(X.t[K][I], y.t[K], a);
        o.t[I] = a;
    }
    return o;
}
template <int NT>
MFGM_DEV Vec<NT> gram(const Mat<NT>& X, const Vec<NT>& y) { return gram<NT>(X, y, vec_zero<NT>()); }

// ---- loads / stores between natural row-major d x d blocks (or d-vectors) and the accumulator layout -----------------------------
// direct: M (identity padded when PAD_EYE);  transposed: M^T.  Branch-free: out-of-range lanes read element 0 and the value is
// consumed unconditionally (times 0), so all loads of a block are in flight together and none is sunk under a branch.
template <int NT, bool TRANSPOSED, bool PAD_EYE>
MFGM_DEV Mat<NT> ld_mat(const double* __restrict__ blk, int d, const LaneId& L, double scale) {
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
                x[I][J][i] = blk[ok ? (TRANSPOSED ? c * d + r : r * d + c) : 0];
            }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * I + L.g + 4 * i, c = 16 * J + L.c;
                const bool ok = r < d && c < d;
                m.t[I][J].r[i] = __builtin_fma(x[I][J][i], ok ? scale : 0.0, (!ok && PAD_EYE && r == c) ? 1.0 : 0.0);
            }
    return m;
}
template <int NT, bool TRANSPOSED>
MFGM_DEV void st_mat(double* __restrict__ blk, int d, const LaneId& L, const Mat<NT>& m) {
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * I + L.g + 4 * i, c = 16 * J + L.c;
                if (r < d && c < d) {
                    if (TRANSPOSED) blk[c * d + r] = m.t[I][J].r[i];
                    else blk[r * d + c] = m.t[I][J].r[i];
                }
            }
}
template <int NT>
MFGM_DEV Vec<NT> ld_vec(const double* __restrict__ v, int d, const LaneId& L, double scale) {
    Vec<NT> o;
    double x[NT][4];
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * I + L.g + 4 * i;
            x[I][i] = v[r < d ? r : 0];
        }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * I + L.g + 4 * i;
            o.t[I].r[i] = x[I][i] * ((L.c == 0 && r < d) ? scale : 0.0);
        }
    return o;
}
template <int NT>
MFGM_DEV void st_vec(double* __restrict__ v, int d, const LaneId& L, const Vec<NT>& a) {
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * I + L.g + 4 * i;
            if (L.c == 0 && r < d) v[r] = a.t[I].r[i];
        }
}

// transpose through LDS (one 16 x 16 tile at a time, 17-double row stride); the workgroup is one wavefront
template <int NT>
MFGM_DEV Mat<NT> mat_transpose(const Mat<NT>& m, double* lds, const LaneId& L) {
    Mat<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[(L.g + 4 * i) * 17 + L.c] = m.t[I][J].r[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) o.t[J][I].r[i] = lds[L.c * 17 + L.g + 4 * i];
            __syncthreads();
        }
    return o;
}

// ds_bpermute of a double with a precomputed byte address (source lane * 4)
MFGM_DEV double bperm(double x, int addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

// Gauss-Jordan row elimination on [A | Bm] (Bm = I on entry).
//   CHOL: A symmetric positive definite  ->  A = L^T in its upper triangle (the strict lower triangle is left with rounding
//         residue and must be masked by the caller), Bm = L^{-1}
//   else: A = L lower triangular         ->  Bm = L^{-1}  (A is consumed)
// In both modes prod *= prod_j 1 / L_jj.
// Row j = 16 J + jj lives in tile row J, lane row jr = jj & 3, register ji = jj >> 2.  Registers holding only rows below the
// pivot need no per-lane selects; the register of the pivot row masks its multiplier.  Finished (pivot) rows stay unscaled in the
// registers during the loop; their scale factors are collected per row in `srow` and applied once at the end.
template <int NT, bool CHOL>
MFGM_DEV void gj(Mat<NT>& A, Mat<NT>& Bm, const LaneId& L, double& prod, int& bad) {
    const int rowaddr[4] = {L.c << 2, (16 | L.c) << 2, (32 | L.c) << 2, (48 | L.c) << 2};   // lane (jr, c), as byte addresses
    const int colbase = (L.lane & 0x30) << 2;                                               // lane (g, 0)
    Vec<NT> srow;
#pragma unroll
    for (int I = 0; I < NT; ++I) srow.t[I] = Tile{{1.0, 1.0, 1.0, 1.0}};
#pragma unroll
    for (int j = 0; j < 16 * NT; ++j) {
        const int J = j >> 4, jj = j & 15, jr = jj & 3, ji = jj >> 2;
        double rA[NT], rB[NT];
#pragma unroll
        for (int Jc = 0; Jc < NT; ++Jc) {
            rA[Jc] = bperm(A.t[J][Jc].r[ji], rowaddr[jr]);       // row j of both matrices (unscaled), per column
            rB[Jc] = bperm(Bm.t[J][Jc].r[ji], rowaddr[jr]);
        }
        double p = bcast(A.t[J][J].r[ji], (jr << 4) | jj);        // pivot A[j][j], read at its home lane
        double s, s2;
        if (CHOL) {
            const bool neg = !(p > 0.0);
            bad |= neg ? 1 : 0;
            p = neg ? 1.0 : p;
            s = rsqrt_nr(p);
            s2 = s;
            prod *= s;
        } else {
            s = rcp_nr(p);
            s2 = 1.0;
            prod *= s;
        }
#pragma unroll
        for (int Jc = 0; Jc < NT; ++Jc) { rA[Jc] *= s; rB[Jc] *= s; }
        srow.t[J].r[ji] = (L.g == jr) ? s : srow.t[J].r[ji];
        const int colj = colbase + 4 * jj;
#pragma unroll
        for (int I = J; I < NT; ++I)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (I == J && i < ji) continue;
                double m = bperm(A.t[I][J].r[i], colj) * s2;      // A[row][j] * s2
                if (I == J && i == ji) m = (L.g > jr) ? m : 0.0;  // rows of this register at or above the pivot are left alone
#pragma unroll
                for (int Jc = 0; Jc < NT; ++Jc) {
                    A.t[I][Jc].r[i] = __builtin_fma(-m, rA[Jc], A.t[I][Jc].r[i]);
                    Bm.t[I][Jc].r[i] = __builtin_fma(-m, rB[Jc], Bm.t[I][Jc].r[i]);
                }
            }
    }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int Jc = 0; Jc < NT; ++Jc)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                A.t[I][Jc].r[i] *= srow.t[I].r[i];
                Bm.t[I][Jc].r[i] *= srow.t[I].r[i];
            }
}

// keep the upper triangle (col >= row)
template <int NT>
MFGM_DEV Mat<NT> mat_upper(const Mat<NT>& m, const LaneId& L) {
    Mat<NT> o;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) o.t[I][J].r[i] = (16 * J + L.c >= 16 * I + L.g + 4 * i) ? m.t[I][J].r[i] : 0.0;
    return o;
}

// sum of squares of a vector (uniform result)
template <int NT>
MFGM_DEV double vec_sumsq(const Vec<NT>& y, const LaneId& L) {
    double q = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int i = 0; i < 4; ++i) q = __builtin_fma(y.t[I].r[i], y.t[I].r[i], q);
    q = (L.c == 0) ? q : 0.0;
    return bcast(q, 0) + bcast(q, 16) + bcast(q, 32) + bcast(q, 48);
}

// ---- reduce ------------------------------------------------------------------------------------------------------------------------
template <int NT, bool HAS_RHS, bool HAS_CORR>
static __global__ __launch_bounds__(64) void km_reduce(WideArgs a) {
    __shared__ double lds[16 * 17];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);   // this launch covers segments [seg_lo, seg_lo + nseg)
    const int t0 = p * R, len = min(R, n - t0);
    int bad = 0;
    double prod = 1.0;
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
        const Mat<NT> St = ld_mat<NT, true, false>(wblk(a.Sg, b, n, t, EF), d, L, a.aS);
        Mat<NT> Fn = ld_F(t + 1);
        Vec<NT> hn = HAS_RHS ? ld_h(t + 1) : vec_zero<NT>();
        Mat<NT> X = mat_eye<NT>(L);
        gj<NT, true>(F, X, L, prod, bad);
        const Mat<NT> Xt = mat_transpose<NT>(X, lds, L);
        const Mat<NT> V = gram<NT>(Xt, W);
        const Mat<NT> Gt = gram<NT>(Xt, St);
        const Mat<NT> nGt = mat_neg<NT>(Gt);
        Racc = gram<NT>(V, V, Racc);
        Fn = gram<NT>(nGt, Gt, Fn);
        W = gram<NT>(nGt, V);
        if (HAS_RHS) {
            const Vec<NT> y = gram<NT>(Xt, h);
            rho = gram<NT>(V, y, rho);
            hn = gram<NT>(nGt, y, hn);
        }
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
    if (bad && L.lane == 0) flag_not_pd(a.info, a.lv.level, b * a.lv.P + p);
}

// ---- forward -----------------------------------------------------------------------------------------------------------------------
template <int NT, bool HAS_RHS, bool HAS_CORR, bool HAS_UP>
static __global__ __launch_bounds__(64) void km_forward(WideArgs a) {
    __shared__ double lds[16 * 17];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);   // this launch covers segments [seg_lo, seg_lo + nseg)
    const int t0 = p * R, len = min(R, n - t0);
    int bad = 0;
    Mat<NT> C = mat_zero<NT>();
    Vec<NT> cv = vec_zero<NT>();
    if (HAS_UP && p > 0) {
        // boundary state on the separator to the left:  F_a = Ltil Ltil^T + R_p,  h_a = Ltil ytil + rho_p
        const int un = a.up.n;
        const Mat<NT> Ltu = ld_mat<NT, true, false>(wblk(a.uL, b, un, p - 1, EF), d, L, 1.0);
        Mat<NT> Fa = ld_mat<NT, false, true>(wblk(a.uRsub, b, un, p - 1, EF), d, L, 1.0);
        Fa = gram<NT>(Ltu, Ltu, Fa);
        const Mat<NT> Sat = ld_mat<NT, true, false>(wblk(a.Sg, b, n, t0 - 1, EF), d, L, a.aS);
        Mat<NT> X = mat_eye<NT>(L);
        double dummy = 1.0;
        gj<NT, true>(Fa, X, L, dummy, bad);
        const Mat<NT> Xt = mat_transpose<NT>(X, lds, L);
        const Mat<NT> Gat = gram<NT>(Xt, Sat);
        C = gram<NT>(Gat, Gat);
        const bool keep = (a.store_left && p == a.seg_lo);       // sharded chain: see WideArgs::store_left
        if (keep) {
            st_mat<NT, true>(wblk(a.Lg, b, n, t0 - 1, EF), d, L, mat_upper<NT>(Fa, L));      // Fa holds L^T
            st_mat<NT, true>(wblk(a.Gg, b, n, t0 - 1, EF), d, L, Gat);
        }
        if (HAS_RHS) {
            Vec<NT> ha = ld_vec<NT>(wblk(a.urho, b, un, p - 1, d), d, L, 1.0);
            ha = gram<NT>(Ltu, ld_vec<NT>(wblk(a.uy, b, un, p - 1, d), d, L, 1.0), ha);
            const Vec<NT> ya = gram<NT>(Xt, ha);
            if (keep) st_vec<NT>(wblk(a.yg, b, n, t0 - 1, d), d, L, ya);
            cv = gram<NT>(Gat, ya);
        }
    }
    double quad = 0.0;
    LogAcc la;                                   // prod_j 1 / L_jj over this segment, mantissa / exponent form
    la.init();
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
        Mat<NT> X = mat_eye<NT>(L);
        double prod = 1.0;
        gj<NT, true>(F, X, L, prod, bad);
        la.mul(prod);
        la.renorm();
        st_mat<NT, true>(wblk(a.Lg, b, n, t, EF), d, L, mat_upper<NT>(F, L));            // F holds L^T
        const Mat<NT> Xt = mat_transpose<NT>(X, lds, L);
        const Mat<NT> Gt = gram<NT>(Xt, St);
        if (has_next) st_mat<NT, true>(wblk(a.Gg, b, n, t, EF), d, L, Gt);
        C = gram<NT>(Gt, Gt);
        if (HAS_RHS) {
            const Vec<NT> y = gram<NT>(Xt, h);
            st_vec<NT>(wblk(a.yg, b, n, t, d), d, L, y);
            cv = gram<NT>(Gt, y);
            quad += vec_sumsq<NT>(y, L);
        }
    }
    if (a.part && L.lane == 0) {
        a.part[b * P + p] = -la.value();
        a.part[a.lv.Lpad + b * P + p] = quad;
    }
    if (bad && L.lane == 0) flag_not_pd(a.info, a.lv.level, b * a.lv.P + p);
}

// ---- backward ----------------------------------------------------------------------------------------------------------------------
template <int NT, bool HAS_RHS, bool HAS_UP, bool WANT_SUB>
static __global__ __launch_bounds__(64) void km_backward(WideArgs a) {
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);   // this launch covers segments [seg_lo, seg_lo + nseg)
    const int t0 = p * R, len = min(R, n - t0), te = t0 + len - 1;
    (void)P;
    int bad = 0;
    auto inv_L = [&](Mat<NT> Lm) {
        Mat<NT> X = mat_eye<NT>(L);
        double dummy = 1.0;
        gj<NT, false>(Lm, X, L, dummy, bad);
        return X;
    };
    Mat<NT> Sn;
    Vec<NT> xn = vec_zero<NT>();
    if (HAS_UP) {
        Sn = ld_mat<NT, false, false>(wblk(a.uSig, b, a.up.n, p, EF), d, L, 1.0);
        if (HAS_RHS) xn = ld_vec<NT>(wblk(a.umu, b, a.up.n, p, d), d, L, 1.0);
    } else {
        const Vec<NT> yv = HAS_RHS ? ld_vec<NT>(wblk(a.yg, b, n, te, d), d, L, 1.0) : vec_zero<NT>();
        const Mat<NT> X = inv_L(ld_mat<NT, false, true>(wblk(a.Lg, b, n, te, EF), d, L, 1.0));
        Sn = gram<NT>(X, X);
        if (HAS_RHS) xn = gram<NT>(X, yv);
    }
    st_mat<NT, false>(wblk(a.Sigg, b, n, te, EF), d, L, Sn);
    if (HAS_RHS) st_vec<NT>(wblk(a.mug, b, n, te, d), d, L, xn);
    auto step = [&](int t, bool write_node) {
        // every load of the step is issued before the elimination, whose latency then covers them
        const Mat<NT> Lm = ld_mat<NT, false, true>(wblk(a.Lg, b, n, t, EF), d, L, 1.0);
        const Mat<NT> Gt = ld_mat<NT, true, false>(wblk(a.Gg, b, n, t, EF), d, L, 1.0);
        Mat<NT> Gm = mat_zero<NT>();
        Vec<NT> yv = vec_zero<NT>();
        if (HAS_RHS) {
            Gm = ld_mat<NT, false, false>(wblk(a.Gg, b, n, t, EF), d, L, 1.0);
            yv = ld_vec<NT>(wblk(a.yg, b, n, t, d), d, L, 1.0);
        }
        const Mat<NT> X = inv_L(Lm);
        const Mat<NT> H = gram<NT>(Gt, X);                 // G L^{-1}
        const Mat<NT> SnH = gram<NT>(Sn, H);               // Sigma_n H
        if (WANT_SUB) st_mat<NT, false>(wblk(a.Subg, b, n, t, EF), d, L, mat_neg<NT>(SnH));
        if (!write_node) return;
        const Mat<NT> Sig = gram<NT>(H, SnH, gram<NT>(X, X));
        st_mat<NT, false>(wblk(a.Sigg, b, n, t, EF), d, L, Sig);
        if (HAS_RHS) {
            const Vec<NT> v = gram<NT>(mat_neg<NT>(Gm), xn, yv);      // y - G^T x_n
            xn = gram<NT>(X, v);
            st_vec<NT>(wblk(a.mug, b, n, t, d), d, L, xn);
        }
        Sn = Sig;
    };
    for (int s = len - 2; s >= 0; --s) step(t0 + s, true);
    if (WANT_SUB && p > 0) step(t0 - 1, false);
    if (bad && L.lane == 0) flag_not_pd(a.info, a.lv.level, b * a.lv.P + p);
}

// ---- SSM parameters -> naturals / precision blocks, one wavefront per node (same outputs as k_ssm_to_naturals) ------------------------
// With X = chol^{-1}:  Qi = gram(X, X);  M = Qi_{t+1} A = gram(Qi_{t+1}, A);  diag = cD (Qi_t + gram(A, M));  sub = cS M;
// lin = gram(Qi_t, off_t) - gram(A, gram(Qi_{t+1}, off_{t+1})) -- Gram products only, no transposes.
template <int NT, bool WANT_LIN>
static __global__ __launch_bounds__(64) void km_ssm_to_naturals(int B, int T, int d, const double* __restrict__ Ag,
                                                               const double* __restrict__ offg, const double* __restrict__ cholg,
                                                               double cD, double cS, double* __restrict__ ling,
                                                               double* __restrict__ diagg, double* __restrict__ subg,
                                                               double* __restrict__ part_logdet) {
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int EF = d * d;
    const int b = blockIdx.x / T, t = blockIdx.x - b * T;
    (void)B;
    int bad = 0;
    const bool has_next = (t + 1 < T);
    // all loads first
    const Mat<NT> C0 = ld_mat<NT, false, true>(wblk(cholg, b, T, t, EF), d, L, 1.0);
    const Mat<NT> C1 = has_next ? ld_mat<NT, false, true>(wblk(cholg, b, T, t + 1, EF), d, L, 1.0) : mat_eye<NT>(L);
    const Mat<NT> A = has_next ? ld_mat<NT, false, false>(wblk(Ag, b, T, t, EF), d, L, 1.0) : mat_zero<NT>();
    Vec<NT> o0 = vec_zero<NT>(), o1 = vec_zero<NT>();
    if (WANT_LIN) {
        o0 = ld_vec<NT>(wblk(offg, b, T, t, d), d, L, 1.0);
        if (has_next) o1 = ld_vec<NT>(wblk(offg, b, T, t + 1, d), d, L, 1.0);
    }
    double prod = 1.0, dummy = 1.0;
    Mat<NT> X = mat_eye<NT>(L), Lm = C0;
    gj<NT, false>(Lm, X, L, prod, bad);
    Mat<NT> Qi = gram<NT>(X, X);
    Vec<NT> lin = WANT_LIN ? gram<NT>(Qi, o0) : vec_zero<NT>();
    Mat<NT> M = mat_zero<NT>();
    if (has_next) {
        Mat<NT> X1 = mat_eye<NT>(L);
        Lm = C1;
        gj<NT, false>(Lm, X1, L, dummy, bad);
        const Mat<NT> Q1 = gram<NT>(X1, X1);
        M = gram<NT>(Q1, A);
        Qi = gram<NT>(A, M, Qi);
        if (WANT_LIN) lin = gram<NT>(mat_neg<NT>(A), gram<NT>(Q1, o1), lin);
    }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Qi.t[I][J].r[i] *= cD;
                M.t[I][J].r[i] *= cS;
            }
    st_mat<NT, false>(wblk(diagg, b, T, t, EF), d, L, Qi);
    st_mat<NT, false>(wblk(subg, b, T, t, EF), d, L, M);
    if (WANT_LIN) st_vec<NT>(wblk(ling, b, T, t, d), d, L, lin);
    if (part_logdet && L.lane == 0) part_logdet[blockIdx.x] = -log(prod);      // sum log diag(chol_t)
}

}  // namespace mfgm
