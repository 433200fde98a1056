// MFMA sweeps for block sizes 8 < d <= 16: one wavefront per chain segment, every d x d block held as ONE 16 x 16 tile in the
// accumulator layout of v_mfma_f64_16x16x4_f64 (lane (g = lane>>4, c = lane&15), register i  <->  element (row g + 4i, col c);
// blocks with d < 16 are padded in registers with an identity diagonal).
//
// The f64 MFMA takes A[m = lane&15][k = lane>>4] and B[k = lane>>4][n = lane&15]; feeding register i of two tiles X and Y that
// are both in accumulator layout as the A and B operands of four chained MFMAs (i = 0..3) sums over k = g + 4i and yields
//
//        gram(X, Y) = X^T Y        in accumulator layout again, with no cross-lane movement at all.
//
// The three passes of the partitioned solver (mfgm_sweeps.h / mfgm_wide.h: reduce, forward, backward; same level recursion,
// same natural-layout arrays and outputs as the wide path) are therefore written with Gram products only:
//   forward / reduce :  Xt = L^{-T};  G^T = gram(Xt, S^T),  y = gram(Xt, h),  G G^T = gram(G^T, G^T),  G y = gram(G^T, y),
//                       spike  V = gram(Xt, W),  R += gram(V, V),  rho += gram(V, y),  W' = -gram(G^T, V)
//   backward         :  X = L^{-1};   H = G X = gram(G^T, X),  Sn H = gram(Sn, H),  Sigma = gram(X, X) + gram(H, Sn H),
//                       x = gram(X, y - gram(G, x_n))
// S^T / G^T are obtained by loading the block transposed from memory; vectors ride in column 0 of a tile.  The only part that is
// not a matrix product is the 16-pivot Cholesky / triangular inversion, done as Gauss-Jordan row operations on [F | I] with
// ds_bpermute row / column broadcasts (gj16), and one LDS transpose of L^{-1} per node in reduce / forward.
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

MFGM_DEV Tile tile_zero() { return Tile{{0.0, 0.0, 0.0, 0.0}}; }
MFGM_DEV Tile tile_eye(const LaneId& L) {
    Tile t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t.r[i] = (L.g + 4 * i == L.c) ? 1.0 : 0.0;
    return t;
}
MFGM_DEV Tile tile_neg(const Tile& a) { return Tile{{-a.r[0], -a.r[1], -a.r[2], -a.r[3]}}; }
MFGM_DEV Tile tile_sub(const Tile& a, const Tile& b) { return Tile{{a.r[0] - b.r[0], a.r[1] - b.r[1], a.r[2] - b.r[2], a.r[3] - b.r[3]}}; }
MFGM_DEV Tile tile_add(const Tile& a, const Tile& b) { return Tile{{a.r[0] + b.r[0], a.r[1] + b.r[1], a.r[2] + b.r[2], a.r[3] + b.r[3]}}; }

// acc + X^T Y
MFGM_DEV Tile gram(const Tile& X, const Tile& Y, const Tile& acc) {
    v4d a = {acc.r[0], acc.r[1], acc.r[2], acc.r[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) a = __builtin_amdgcn_mfma_f64_16x16x4f64(X.r[i], Y.r[i], a, 0, 0, 0);
    return Tile{{a[0], a[1], a[2], a[3]}};
}
MFGM_DEV Tile gram(const Tile& X, const Tile& Y) { return gram(X, Y, tile_zero()); }

// ---- loads / stores between natural row-major d x d blocks (or d-vectors) and the accumulator layout -----------------------------
// direct: tile = M (identity padded when PAD_EYE);  transposed: tile = M^T
template <bool TRANSPOSED, bool PAD_EYE>
MFGM_DEV Tile ld_tile(const double* __restrict__ blk, int d, const LaneId& L, double scale) {
    // branch-free: out-of-range lanes read element 0 and discard it, so that all four loads are in flight together
    Tile t;
    double x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = L.g + 4 * i;
        const bool ok = r < d && L.c < d;
        x[i] = blk[ok ? (TRANSPOSED ? L.c * d + r : r * d + L.c) : 0];
    }
    // the loaded value is consumed unconditionally (times 0 where out of range) so that the compiler cannot sink the load
    // back under a branch with a wait of its own
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = L.g + 4 * i;
        const bool ok = r < d && L.c < d;
        t.r[i] = __builtin_fma(x[i], ok ? scale : 0.0, (!ok && PAD_EYE && r == L.c) ? 1.0 : 0.0);
    }
    return t;
}
template <bool TRANSPOSED>
MFGM_DEV void st_tile(double* __restrict__ blk, int d, const LaneId& L, const Tile& t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = L.g + 4 * i;
        if (r < d && L.c < d) {
            if (TRANSPOSED) blk[L.c * d + r] = t.r[i];
            else blk[r * d + L.c] = t.r[i];
        }
    }
}
// vectors live in column 0
MFGM_DEV Tile ld_vec(const double* __restrict__ v, int d, const LaneId& L, double scale) {
    Tile t;
    double x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = L.g + 4 * i;
        x[i] = v[r < d ? r : 0];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = L.g + 4 * i;
        t.r[i] = x[i] * ((L.c == 0 && r < d) ? scale : 0.0);
    }
    return t;
}
MFGM_DEV void st_vec(double* __restrict__ v, int d, const LaneId& L, const Tile& t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = L.g + 4 * i;
        if (L.c == 0 && r < d) v[r] = t.r[i];
    }
}

// transpose through LDS (17-double row stride); the workgroup is one wavefront
MFGM_DEV Tile tile_transpose(const Tile& t, double* lds, const LaneId& L) {
#pragma unroll
    for (int i = 0; i < 4; ++i) lds[(L.g + 4 * i) * 17 + L.c] = t.r[i];
    __syncthreads();
    Tile o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.r[i] = lds[L.c * 17 + L.g + 4 * i];
    __syncthreads();
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
//         residue and must be masked by the caller), Bm = L^{-1};  prod *= prod_j 1/L_jj
//   else: A = L lower triangular         ->  Bm = L^{-1}  (A is consumed)
// Row j lives in lane row jr = j & 3, register ji = j >> 2: registers i < ji hold finished rows, registers i > ji only rows
// below the pivot (no per-lane selects), register ji mixes finished rows, the pivot row and rows below.
template <bool CHOL>
MFGM_DEV void gj16(Tile& A, Tile& Bm, const LaneId& L, double& prod, int& bad) {
    const int rowaddr[4] = {L.c << 2, (16 | L.c) << 2, (32 | L.c) << 2, (48 | L.c) << 2};   // lane (jr, c), as byte addresses
    const int colbase = (L.lane & 0x30) << 2;                                               // lane (g, 0)
    // Finished (pivot) rows stay unscaled in the registers during the loop; their scale factors are collected per row in
    // `srow` and applied once at the end, which keeps per-lane selects out of the pivot loop.
    Tile srow = {{1.0, 1.0, 1.0, 1.0}};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int jr = j & 3, ji = j >> 2;
        double rA = bperm(A.r[ji], rowaddr[jr]);                // row j of both tiles (unscaled), per column
        double rB = bperm(Bm.r[ji], rowaddr[jr]);
        double p = bcast(A.r[ji], (jr << 4) | j);               // pivot A[j][j], read at its home lane: the row broadcast
                                                                 // above is then off the critical path (it overlaps the rsqrt)
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
        }
        rA *= s;
        rB *= s;
        srow.r[ji] = (L.g == jr) ? s : srow.r[ji];
        const int colj = colbase + 4 * j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < ji) continue;
            double m = bperm(A.r[i], colj) * s2;               // A[row][j] * s2
            if (i == ji) m = (L.g > jr) ? m : 0.0;             // rows of this register at or above the pivot are left alone
            A.r[i] = __builtin_fma(-m, rA, A.r[i]);
            Bm.r[i] = __builtin_fma(-m, rB, Bm.r[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A.r[i] *= srow.r[i];
        Bm.r[i] *= srow.r[i];
    }
}

// keep the upper triangle (c >= r) of a tile
MFGM_DEV Tile tile_upper(const Tile& t, const LaneId& L) {
    Tile o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.r[i] = (L.c >= L.g + 4 * i) ? t.r[i] : 0.0;
    return o;
}

// sum of squares of column 0 (uniform result)
MFGM_DEV double vec_sumsq(const Tile& y, const LaneId& L) {
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) q = __builtin_fma(y.r[i], y.r[i], q);
    q = (L.c == 0) ? q : 0.0;
    return bcast(q, 0) + bcast(q, 16) + bcast(q, 32) + bcast(q, 48);
}

// ---- reduce ------------------------------------------------------------------------------------------------------------------------
template <bool HAS_RHS, bool HAS_CORR>
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
        Tile F = ld_tile<false, true>(wblk(a.Dg, b, n, t, EF), d, L, a.aD);
        if (HAS_CORR) F = tile_sub(F, ld_tile<false, false>(wblk(a.Dcorr, b, n, t, EF), d, L, 1.0));
        return F;
    };
    auto ld_h = [&](int t) {
        Tile h = ld_vec(wblk(a.rg, b, n, t, d), d, L, a.aR);
        if (HAS_CORR) h = tile_sub(h, ld_vec(wblk(a.rcorr, b, n, t, d), d, L, 1.0));
        return h;
    };
    Tile F = ld_F(t0);
    Tile W = (p > 0) ? ld_tile<false, false>(wblk(a.Sg, b, n, t0 - 1, EF), d, L, a.aS) : tile_zero();
    Tile h = HAS_RHS ? ld_h(t0) : tile_zero();
    Tile Racc = tile_zero(), rho = tile_zero();
    for (int s = 0; s < len - 1; ++s) {
        const int t = t0 + s;
        const Tile St = ld_tile<true, false>(wblk(a.Sg, b, n, t, EF), d, L, a.aS);
        Tile Fn = ld_F(t + 1);
        Tile hn = HAS_RHS ? ld_h(t + 1) : tile_zero();
        Tile X = tile_eye(L);
        gj16<true>(F, X, L, prod, bad);
        const Tile Xt = tile_transpose(X, lds, L);
        const Tile V = gram(Xt, W);
        const Tile Gt = gram(Xt, St);
        const Tile nGt = tile_neg(Gt);
        Racc = gram(V, V, Racc);
        Fn = gram(nGt, Gt, Fn);
        W = gram(nGt, V);
        if (HAS_RHS) {
            const Tile y = gram(Xt, h);
            rho = gram(V, y, rho);
            hn = gram(nGt, y, hn);
        }
        F = Fn;
        h = hn;
    }
    const int un = a.up.n;
    st_tile<false>(wblk(a.uDhat, b, un, p, EF), d, L, F);
    st_vec(wblk(a.urhat, b, un, p, d), d, L, h);
    if (p == P - 1) {
        st_tile<false>(wblk(a.uRsub, b, un, p, EF), d, L, tile_zero());
        st_tile<false>(wblk(a.uS, b, un, p, EF), d, L, tile_zero());
        st_vec(wblk(a.urho, b, un, p, d), d, L, tile_zero());
    }
    if (p > 0) {
        st_tile<false>(wblk(a.uS, b, un, p - 1, EF), d, L, W);
        st_tile<false>(wblk(a.uRsub, b, un, p - 1, EF), d, L, Racc);
        st_vec(wblk(a.urho, b, un, p - 1, d), d, L, rho);
    }
    if (bad && L.lane == 0) atomicMax(a.info, 1);
}

// ---- forward -----------------------------------------------------------------------------------------------------------------------
template <bool HAS_RHS, bool HAS_CORR, bool HAS_UP>
static __global__ __launch_bounds__(64) void km_forward(WideArgs a) {
    __shared__ double lds[16 * 17];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);   // this launch covers segments [seg_lo, seg_lo + nseg)
    const int t0 = p * R, len = min(R, n - t0);
    int bad = 0;
    Tile C = tile_zero(), cv = tile_zero();
    if (HAS_UP && p > 0) {
        // boundary state on the separator to the left:  F_a = Ltil Ltil^T + R_p,  h_a = Ltil ytil + rho_p
        const int un = a.up.n;
        const Tile Ltu = ld_tile<true, false>(wblk(a.uL, b, un, p - 1, EF), d, L, 1.0);
        Tile Fa = ld_tile<false, true>(wblk(a.uRsub, b, un, p - 1, EF), d, L, 1.0);
        Fa = gram(Ltu, Ltu, Fa);
        const Tile Sat = ld_tile<true, false>(wblk(a.Sg, b, n, t0 - 1, EF), d, L, a.aS);
        Tile X = tile_eye(L);
        double dummy = 1.0;
        gj16<true>(Fa, X, L, dummy, bad);
        const Tile Xt = tile_transpose(X, lds, L);
        const Tile Gat = gram(Xt, Sat);
        C = gram(Gat, Gat);
        if (HAS_RHS) {
            Tile ha = ld_vec(wblk(a.urho, b, un, p - 1, d), d, L, 1.0);
            ha = gram(Ltu, ld_vec(wblk(a.uy, b, un, p - 1, d), d, L, 1.0), ha);
            cv = gram(Gat, gram(Xt, ha));
        }
    }
    double quad = 0.0;
    LogAcc la;                                   // prod_j 1 / L_jj over this segment, mantissa / exponent form
    la.init();
    for (int s = 0; s < len; ++s) {
        const int t = t0 + s;
        Tile F = ld_tile<false, true>(wblk(a.Dg, b, n, t, EF), d, L, a.aD);
        if (HAS_CORR) F = tile_sub(F, ld_tile<false, false>(wblk(a.Dcorr, b, n, t, EF), d, L, 1.0));
        F = tile_sub(F, C);
        Tile h = tile_zero();
        if (HAS_RHS) {
            h = ld_vec(wblk(a.rg, b, n, t, d), d, L, a.aR);
            if (HAS_CORR) h = tile_sub(h, ld_vec(wblk(a.rcorr, b, n, t, d), d, L, 1.0));
            h = tile_sub(h, cv);
        }
        const bool has_next = (t + 1 < n);
        const Tile St = has_next ? ld_tile<true, false>(wblk(a.Sg, b, n, t, EF), d, L, a.aS) : tile_zero();
        Tile X = tile_eye(L);
        double prod = 1.0;
        gj16<true>(F, X, L, prod, bad);
        la.mul(prod);
        la.renorm();
        st_tile<true>(wblk(a.Lg, b, n, t, EF), d, L, tile_upper(F, L));            // F holds L^T
        const Tile Xt = tile_transpose(X, lds, L);
        const Tile Gt = gram(Xt, St);
        if (has_next) st_tile<true>(wblk(a.Gg, b, n, t, EF), d, L, Gt);
        C = gram(Gt, Gt);
        if (HAS_RHS) {
            const Tile y = gram(Xt, h);
            st_vec(wblk(a.yg, b, n, t, d), d, L, y);
            cv = gram(Gt, y);
            quad += vec_sumsq(y, L);
        }
    }
    if (a.part && L.lane == 0) {
        a.part[b * P + p] = -la.value();
        a.part[a.lv.Lpad + b * P + p] = quad;
    }
    if (bad && L.lane == 0) atomicMax(a.info, 1);
}

// ---- backward ----------------------------------------------------------------------------------------------------------------------
template <bool HAS_RHS, bool HAS_UP, bool WANT_SUB>
static __global__ __launch_bounds__(64) void km_backward(WideArgs a) {
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);   // this launch covers segments [seg_lo, seg_lo + nseg)
    const int t0 = p * R, len = min(R, n - t0), te = t0 + len - 1;
    int bad = 0;
    auto inv_L = [&](Tile Lm) {
        Tile X = tile_eye(L);
        double dummy = 1.0;
        gj16<false>(Lm, X, L, dummy, bad);
        return X;
    };
    Tile Sn, xn = tile_zero();
    if (HAS_UP) {
        Sn = ld_tile<false, false>(wblk(a.uSig, b, a.up.n, p, EF), d, L, 1.0);
        if (HAS_RHS) xn = ld_vec(wblk(a.umu, b, a.up.n, p, d), d, L, 1.0);
    } else {
        const Tile yv = HAS_RHS ? ld_vec(wblk(a.yg, b, n, te, d), d, L, 1.0) : tile_zero();
        const Tile X = inv_L(ld_tile<false, true>(wblk(a.Lg, b, n, te, EF), d, L, 1.0));
        Sn = gram(X, X);
        if (HAS_RHS) xn = gram(X, yv);
    }
    st_tile<false>(wblk(a.Sigg, b, n, te, EF), d, L, Sn);
    if (HAS_RHS) st_vec(wblk(a.mug, b, n, te, d), d, L, xn);
    auto step = [&](int t, bool write_node) {
        // every load of the step is issued before the elimination, whose latency then covers them
        const Tile Lm = ld_tile<false, true>(wblk(a.Lg, b, n, t, EF), d, L, 1.0);
        const Tile Gt = ld_tile<true, false>(wblk(a.Gg, b, n, t, EF), d, L, 1.0);
        Tile Gm = tile_zero(), yv = tile_zero();
        if (HAS_RHS) {
            Gm = ld_tile<false, false>(wblk(a.Gg, b, n, t, EF), d, L, 1.0);
            yv = ld_vec(wblk(a.yg, b, n, t, d), d, L, 1.0);
        }
        const Tile X = inv_L(Lm);
        const Tile H = gram(Gt, X);                 // G L^{-1}
        const Tile SnH = gram(Sn, H);               // Sigma_n H
        if (WANT_SUB) st_tile<false>(wblk(a.Subg, b, n, t, EF), d, L, tile_neg(SnH));
        if (!write_node) return;
        const Tile Sig = gram(H, SnH, gram(X, X));
        st_tile<false>(wblk(a.Sigg, b, n, t, EF), d, L, Sig);
        if (HAS_RHS) {
            const Tile v = gram(tile_neg(Gm), xn, yv);      // y - G^T x_n
            xn = gram(X, v);
            st_vec(wblk(a.mug, b, n, t, d), d, L, xn);
        }
        Sn = Sig;
    };
    for (int s = len - 2; s >= 0; --s) step(t0 + s, true);
    if (WANT_SUB && p > 0) step(t0 - 1, false);
}

}  // namespace mfgm
