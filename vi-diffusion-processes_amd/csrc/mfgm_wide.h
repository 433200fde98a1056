// "Wide" sweeps for block sizes 8 < d <= 32: one WAVEFRONT owns one chain segment, lane i holds row i of every d x d
// block in registers, and rows of other matrices reach it through v_readlane broadcasts (no LDS, no barriers).
// Same three-pass partitioned algorithm as mfgm_sweeps.h (reduce / forward / backward, same level recursion and the
// same natural-order outputs); arrays are in the reference's natural layout [B][n][d*d] / [B][n][d] (full blocks; only the
// lower triangle of symmetric inputs is read), so no re-layout is needed: a wavefront reads a whole block contiguously.
// Blocks are zero-padded to DM in {16, 32} inside registers with an identity diagonal, which leaves every result unchanged.
// This path is functional rather than tuned (the d = 16 MFMA variant is future work).
#pragma once
#include "mfgm_layout.h"
#include "mfgm_math.h"

namespace mfgm {

MFGM_DEV double bcast(double x, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

struct WideArgs {
    LevelDesc lv, up;
    int d;
    int seg_lo, nseg;   // segments of this level covered by the launch (all of them except below the exchange level of a sharded chain)
    int store_left;     // forward sweep of a sharded chain: the first covered segment also stores the factor blocks (L, L_{t+1,t}, y) of
                        // the separator on its left, which it reconstructs anyway (that node belongs to the neighbouring process)
    const double* Dg; const double* Sg; const double* rg; const double* Dcorr; const double* rcorr;
    double aD, aS, aR;
    double* Lg; double* Gg; double* yg; double* part;
    double* Sigg; double* Subg; double* mug;
    double* uDhat; double* uRsub; double* uS; double* urhat; double* urho;
    const double* uL; const double* uy; const double* uSig; const double* umu;
    int* info;
    // sparse-CVI inputs (inverse-form kernels, level 0, one chain): when site2 != NULL the arrays Dg, Sg, rg are the PRIOR naturals and
    // the posterior naturals are formed on load by overlap-adding the sites site1 [n + 1, 2d], site2 [n + 1, 2d, 2d] on pairs of
    // consecutive states (sparse_variational_cvi.py:160-172; what k_sparse_theta would write out)
    const double* site1; const double* site2;
    // site2 layout: 0 = the reference's [n + 1, 2d, 2d]; 1 = quadrant-packed [n + 1, 2 ET + EF] (ET = d (d + 1) / 2, EF = d^2): the upper
    // left block (first state of the pair) as a packed lower triangle, the lower left block (second x first state) in full, the lower
    // right block as a packed lower triangle -- the upper right block is the transpose of the lower left one and is never read
    int site_packed;
};

// row i of a d x d block (zero padded), column i of a block
template <int DM>
MFGM_DEV void ld_row(const double* __restrict__ blk, int d, int i, double scale, double (&out)[DM]) {
    // branch-free (out-of-range elements read element 0 and are discarded): all DM loads are in flight together
#pragma unroll
    for (int k = 0; k < DM; ++k) out[k] = blk[(i < d && k < d) ? i * d + k : 0];
#pragma unroll
    for (int k = 0; k < DM; ++k) out[k] = (i < d && k < d) ? scale * out[k] : 0.0;
}
template <int DM>
MFGM_DEV void ld_col(const double* __restrict__ blk, int d, int i, double scale, double (&out)[DM]) {
#pragma unroll
    for (int k = 0; k < DM; ++k) out[k] = blk[(i < d && k < d) ? k * d + i : 0];
#pragma unroll
    for (int k = 0; k < DM; ++k) out[k] = (i < d && k < d) ? scale * out[k] : 0.0;
}
template <int DM>
MFGM_DEV void st_row(double* __restrict__ blk, int d, int i, const double (&v)[DM]) {
#pragma unroll
    for (int k = 0; k < DM; ++k)
        if (i < d && k < d) blk[i * d + k] = v[k];
}
template <int DM>
MFGM_DEV void st_col(double* __restrict__ blk, int d, int i, const double (&v)[DM]) {
#pragma unroll
    for (int k = 0; k < DM; ++k)
        if (i < d && k < d) blk[k * d + i] = v[k];
}

// C_i[j] += alpha * dot(A_i, B_j)     (A B^T, rows of B broadcast)
template <int DM>
MFGM_DEV void mm_abt(const double (&A)[DM], const double (&B)[DM], double alpha, double (&C)[DM]) {
#pragma unroll
    for (int j = 0; j < DM; ++j) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) t = __builtin_fma(A[k], bcast(B[k], j), t);
        C[j] = __builtin_fma(alpha, t, C[j]);
    }
}
// C_i[:] += alpha * sum_k A_i[k] B_k[:]   (A B, rows of B broadcast)
template <int DM>
MFGM_DEV void mm_ab(const double (&A)[DM], const double (&B)[DM], double alpha, double (&C)[DM]) {
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        const double a = alpha * A[k];
#pragma unroll
        for (int j = 0; j < DM; ++j) C[j] = __builtin_fma(a, bcast(B[j], k), C[j]);
    }
}
// sum_k A_i[k] v_k with v distributed one element per lane
template <int DM>
MFGM_DEV double mv(const double (&A)[DM], double v) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < DM; ++k) t = __builtin_fma(A[k], bcast(v, k), t);
    return t;
}

// In-place Cholesky of the matrix whose row i (lower part) lane i holds in F; then the right-solves X <- X L^{-T} for the
// rows X1, X2 of two other matrices, sharing the broadcasts of L.  invd[j] = 1 / L_jj (uniform).
template <int DM, bool TWO>
MFGM_DEV void chol_rsolve(double (&F)[DM], double (&invd)[DM], double (&X1)[DM], double (&X2)[DM], int lane, int& bad) {
#pragma unroll
    for (int j = 0; j < DM; ++j) {
        double acc = F[j], x1 = X1[j], x2 = TWO ? X2[j] : 0.0;
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const double ljk = bcast(F[k], j);
            acc = __builtin_fma(-F[k], ljk, acc);
            x1 = __builtin_fma(-X1[k], ljk, x1);
            if (TWO) x2 = __builtin_fma(-X2[k], ljk, x2);
        }
        double piv = bcast(acc, j);
        if (!(piv > 0.0)) { bad = 1; piv = 1.0; }
        const double inv = rsqrt_nr(piv);
        invd[j] = inv;
        F[j] = (lane > j) ? acc * inv : ((lane == j) ? piv * inv : 0.0);
        X1[j] = x1 * inv;
        if (TWO) X2[j] = x2 * inv;
    }
}

// y = L^{-1} h, h and y one element per lane
template <int DM>
MFGM_DEV double fsolve(const double (&L)[DM], const double (&invd)[DM], double h, int lane) {
#pragma unroll
    for (int j = 0; j < DM; ++j) {
        const double yj = bcast(h, j) * invd[j];
        h = (lane == j) ? yj : ((lane > j) ? __builtin_fma(-L[j], yj, h) : h);
    }
    return h;
}

// rows of X^T for X = L^{-1}, L given by rows (lower triangular, identity padded): forward substitution on rows, then a
// transpose through LDS (tile: DM*(DM+1) doubles)
template <int DM>
MFGM_DEV void inv_t_rows(const double (&L)[DM], double (&Xt)[DM], double* tile, int lane) {
    double X[DM];
#pragma unroll
    for (int k = 0; k < DM; ++k) X[k] = (k == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        const double invk = 1.0 / bcast(L[k], k);
#pragma unroll
        for (int j = 0; j <= k; ++j) {
            if (lane == k) X[j] *= invk;
            const double xkj = bcast(X[j], k);
            if (lane > k) X[j] = __builtin_fma(-L[k], xkj, X[j]);
        }
    }
    if (lane < DM) {
#pragma unroll
        for (int k = 0; k < DM; ++k) tile[lane * (DM + 1) + k] = X[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DM; ++k) Xt[k] = (lane < DM) ? tile[k * (DM + 1) + lane] : 0.0;
    __syncthreads();
}

// element `lane` of a d-vector (0 beyond d), branch-free
MFGM_DEV double ld_elem(const double* __restrict__ v, int lane, int d) {
    const double x = v[lane < d ? lane : 0];
    return lane < d ? x : 0.0;
}

MFGM_DEV const double* wblk(const double* base, int b, int n, int t, int E) { return base + ((size_t)b * n + t) * E; }
MFGM_DEV double* wblk(double* base, int b, int n, int t, int E) { return base + ((size_t)b * n + t) * E; }

// ---- reduce ---------------------------------------------------------------------------------------------------------
template <int DM, bool HAS_RHS, bool HAS_CORR>
static __global__ __launch_bounds__(64) void kw_reduce(WideArgs a) {
    const int lane = threadIdx.x, d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);   // this launch covers segments [seg_lo, seg_lo + nseg)
    const int t0 = p * R, len = min(R, n - t0);
    int bad = 0;
    double F[DM], Z[DM], Racc[DM], h = 0.0, rho = 0.0;
    ld_row<DM>(wblk(a.Dg, b, n, t0, EF), d, lane, a.aD, F);
    if (HAS_CORR) {
        double c[DM];
        ld_row<DM>(wblk(a.Dcorr, b, n, t0, EF), d, lane, 1.0, c);
#pragma unroll
        for (int k = 0; k < DM; ++k) F[k] -= c[k];
    }
    if (lane >= d && lane < DM) F[lane] = 1.0;
    if (p > 0) ld_col<DM>(wblk(a.Sg, b, n, t0 - 1, EF), d, lane, a.aS, Z);   // Z = W^T, W = S_a
    else {
#pragma unroll
        for (int k = 0; k < DM; ++k) Z[k] = 0.0;
    }
    if (HAS_RHS) {
        h = a.aR * ld_elem(wblk(a.rg, b, n, t0, d), lane, d);
        if (HAS_CORR && lane < d) h -= wblk(a.rcorr, b, n, t0, d)[lane];
    }
#pragma unroll
    for (int k = 0; k < DM; ++k) Racc[k] = 0.0;
    for (int s = 0; s < len - 1; ++s) {
        const int t = t0 + s;
        double G[DM], Fn[DM], invd[DM];
        ld_row<DM>(wblk(a.Sg, b, n, t, EF), d, lane, a.aS, G);
        ld_row<DM>(wblk(a.Dg, b, n, t + 1, EF), d, lane, a.aD, Fn);
        if (HAS_CORR) {
            double c[DM];
            ld_row<DM>(wblk(a.Dcorr, b, n, t + 1, EF), d, lane, 1.0, c);
#pragma unroll
            for (int k = 0; k < DM; ++k) Fn[k] -= c[k];
        }
        if (lane >= d && lane < DM) Fn[lane] = 1.0;
        double hn = 0.0;
        if (HAS_RHS) {
            hn = a.aR * ld_elem(wblk(a.rg, b, n, t + 1, d), lane, d);
            if (HAS_CORR && lane < d) hn -= wblk(a.rcorr, b, n, t + 1, d)[lane];
        }
        chol_rsolve<DM, true>(F, invd, G, Z, lane, bad);   // G <- S L^{-T},  Z <- Z L^{-T}  (Z^T = L^{-1} W)
        double y = 0.0;
        if (HAS_RHS) {
            y = fsolve<DM>(F, invd, h, lane);
            rho += mv<DM>(Z, y);                               // (W^T y)_i = sum_k Z_i[k] y_k
        }
        mm_abt<DM>(Z, Z, 1.0, Racc);                            // R += W^T W = Z Z^T
        mm_abt<DM>(G, G, -1.0, Fn);                             // F' = D' - G G^T
        double Zn[DM];
#pragma unroll
        for (int k = 0; k < DM; ++k) Zn[k] = 0.0;
        mm_abt<DM>(Z, G, -1.0, Zn);                             // W' = -G W  =>  Z' = -Z G^T
        if (HAS_RHS) hn -= mv<DM>(G, y);
#pragma unroll
        for (int k = 0; k < DM; ++k) { F[k] = Fn[k]; Z[k] = Zn[k]; }
        h = hn;
    }
    const int uP = a.up.P, un = a.up.n;
    (void)uP;
    st_row<DM>(wblk(a.uDhat, b, un, p, EF), d, lane, F);
    if (lane < d) wblk(a.urhat, b, un, p, d)[lane] = h;
    if (p == P - 1) {
        double z[DM];
#pragma unroll
        for (int k = 0; k < DM; ++k) z[k] = 0.0;
        st_row<DM>(wblk(a.uRsub, b, un, p, EF), d, lane, z);
        st_row<DM>(wblk(a.uS, b, un, p, EF), d, lane, z);
        if (lane < d) wblk(a.urho, b, un, p, d)[lane] = 0.0;
    }
    if (p > 0) {
        st_col<DM>(wblk(a.uS, b, un, p - 1, EF), d, lane, Z);       // S~ = W = Z^T
        st_row<DM>(wblk(a.uRsub, b, un, p - 1, EF), d, lane, Racc);
        if (lane < d) wblk(a.urho, b, un, p - 1, d)[lane] = rho;
    }
    if (bad && lane == 0) flag_not_pd(a.info, a.lv.level, b * a.lv.P + p);
}

// ---- forward --------------------------------------------------------------------------------------------------------
template <int DM, bool HAS_RHS, bool HAS_CORR, bool HAS_UP>
static __global__ __launch_bounds__(64) void kw_forward(WideArgs a) {
    const int lane = threadIdx.x, d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);   // this launch covers segments [seg_lo, seg_lo + nseg)
    const int t0 = p * R, len = min(R, n - t0);
    int bad = 0;
    double C[DM], c = 0.0;
#pragma unroll
    for (int k = 0; k < DM; ++k) C[k] = 0.0;
    if (HAS_UP && p > 0) {
        // F_a = Ltil Ltil^T + R_p,  h_a = Ltil ytil + rho_p   (natural-order state at the separator on the left)
        const int un = a.up.n;
        double Lt[DM], Fa[DM], Ga[DM], invd[DM], dummy[DM];
        ld_row<DM>(wblk(a.uL, b, un, p - 1, EF), d, lane, 1.0, Lt);
        ld_row<DM>(wblk(a.uRsub, b, un, p - 1, EF), d, lane, 1.0, Fa);
        mm_abt<DM>(Lt, Lt, 1.0, Fa);
        if (lane >= d && lane < DM) Fa[lane] = 1.0;
        double ha = 0.0;
        if (HAS_RHS) {
            const double yt = ld_elem(wblk(a.uy, b, un, p - 1, d), lane, d);
            ha = mv<DM>(Lt, yt) + (ld_elem(wblk(a.urho, b, un, p - 1, d), lane, d));
        }
        ld_row<DM>(wblk(a.Sg, b, n, t0 - 1, EF), d, lane, a.aS, Ga);
        chol_rsolve<DM, false>(Fa, invd, Ga, dummy, lane, bad);
        mm_abt<DM>(Ga, Ga, 1.0, C);
        double ya = 0.0;
        if (HAS_RHS) {
            ya = fsolve<DM>(Fa, invd, ha, lane);
            c = mv<DM>(Ga, ya);
        }
        if (a.store_left && p == a.seg_lo) {
            st_row<DM>(wblk(a.Lg, b, n, t0 - 1, EF), d, lane, Fa);
            st_row<DM>(wblk(a.Gg, b, n, t0 - 1, EF), d, lane, Ga);
            if (HAS_RHS && lane < d) wblk(a.yg, b, n, t0 - 1, d)[lane] = ya;
        }
    }
    double logacc = 0.0, quad = 0.0;
    for (int s = 0; s < len; ++s) {
        const int t = t0 + s;
        double F[DM], G[DM], invd[DM], dummy[DM];
        ld_row<DM>(wblk(a.Dg, b, n, t, EF), d, lane, a.aD, F);
        if (HAS_CORR) {
            double cc[DM];
            ld_row<DM>(wblk(a.Dcorr, b, n, t, EF), d, lane, 1.0, cc);
#pragma unroll
            for (int k = 0; k < DM; ++k) F[k] -= cc[k];
        }
#pragma unroll
        for (int k = 0; k < DM; ++k) F[k] -= C[k];
        if (lane >= d && lane < DM) F[lane] = 1.0;
        double h = 0.0;
        if (HAS_RHS) {
            h = a.aR * ld_elem(wblk(a.rg, b, n, t, d), lane, d);
            if (HAS_CORR && lane < d) h -= wblk(a.rcorr, b, n, t, d)[lane];
            h -= c;
        }
        const bool has_next = (t + 1 < n);
        if (has_next) ld_row<DM>(wblk(a.Sg, b, n, t, EF), d, lane, a.aS, G);
        else {
#pragma unroll
            for (int k = 0; k < DM; ++k) G[k] = 0.0;
        }
        chol_rsolve<DM, false>(F, invd, G, dummy, lane, bad);
        double y = 0.0;
        if (HAS_RHS) y = fsolve<DM>(F, invd, h, lane);
        st_row<DM>(wblk(a.Lg, b, n, t, EF), d, lane, F);
        if (has_next) st_row<DM>(wblk(a.Gg, b, n, t, EF), d, lane, G);
        if (HAS_RHS && lane < d) wblk(a.yg, b, n, t, d)[lane] = y;
#pragma unroll
        for (int k = 0; k < DM; ++k) C[k] = 0.0;
        mm_abt<DM>(G, G, 1.0, C);
        if (HAS_RHS) c = mv<DM>(G, y);
        // log|L_tt| = -sum_j log invd_j ; |y|^2
        double pr = 1.0;
#pragma unroll
        for (int j = 0; j < DM; ++j) pr *= invd[j];
        logacc -= log(pr);
        if (HAS_RHS) {
            double q = y * y;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
            quad += q;
        }
    }
    if (a.part && lane == 0) {
        a.part[b * P + p] = logacc;
        a.part[a.lv.Lpad + b * P + p] = quad;
    }
    if (bad && lane == 0) flag_not_pd(a.info, a.lv.level, b * a.lv.P + p);
}

// ---- backward -------------------------------------------------------------------------------------------------------
// Works with transposed factors held by rows: Lt = L^T (column loads), Gt = G^T, Xt = L^{-T}, Ht = Xt Gt, so that
//   Sigma_t = Xt Xt^T + Ht Sigma_n Ht^T,   Sigma_{t+1,t} = -Sigma_n Ht^T,   x_t = Xt (y - Gt x_n).
template <int DM, bool HAS_RHS, bool HAS_UP, bool WANT_SUB>
static __global__ __launch_bounds__(64) void kw_backward(WideArgs a) {
    const int lane = threadIdx.x, d = a.d, EF = d * d;
    const int P = a.lv.P, R = a.lv.R, n = a.lv.n;
    const int b = blockIdx.x / a.nseg, p = a.seg_lo + ((int)blockIdx.x - b * a.nseg);   // this launch covers segments [seg_lo, seg_lo + nseg)
    const int t0 = p * R, len = min(R, n - t0), te = t0 + len - 1;
    __shared__ double tile[DM * (DM + 1)];
    // Xt = L^{-T} held by rows
    auto inv_t = [&](int t, double (&Xt)[DM]) {
        double L[DM];
        ld_row<DM>(wblk(a.Lg, b, n, t, EF), d, lane, 1.0, L);
        if (lane >= d && lane < DM) L[lane] = 1.0;
        inv_t_rows<DM>(L, Xt, tile, lane);
    };
    double Sn[DM], xn = 0.0;
    if (HAS_UP) {
        ld_row<DM>(wblk(a.uSig, b, a.up.n, p, EF), d, lane, 1.0, Sn);
        if (HAS_RHS) xn = ld_elem(wblk(a.umu, b, a.up.n, p, d), lane, d);
    } else {
        double Xt[DM];
        inv_t(te, Xt);
#pragma unroll
        for (int k = 0; k < DM; ++k) Sn[k] = 0.0;
        mm_abt<DM>(Xt, Xt, 1.0, Sn);
        if (HAS_RHS) {
            const double y = ld_elem(wblk(a.yg, b, n, te, d), lane, d);
            xn = mv<DM>(Xt, y);
        }
    }
    st_row<DM>(wblk(a.Sigg, b, n, te, EF), d, lane, Sn);
    if (HAS_RHS && lane < d) wblk(a.mug, b, n, te, d)[lane] = xn;
    auto step = [&](int t, bool write_node) {
        double Xt[DM], Gt[DM], Ht[DM], T1[DM], Sig[DM], Ssub[DM];
        inv_t(t, Xt);
        ld_col<DM>(wblk(a.Gg, b, n, t, EF), d, lane, 1.0, Gt);
#pragma unroll
        for (int k = 0; k < DM; ++k) { Ht[k] = 0.0; T1[k] = 0.0; Sig[k] = 0.0; Ssub[k] = 0.0; }
        mm_ab<DM>(Xt, Gt, 1.0, Ht);          // Ht = Xt Gt
        mm_abt<DM>(Sn, Ht, -1.0, Ssub);      // Sigma_{t+1,t} = -Sigma_n H = -Sigma_n Ht^T
        if (WANT_SUB) st_row<DM>(wblk(a.Subg, b, n, t, EF), d, lane, Ssub);
        if (!write_node) return;
        mm_ab<DM>(Ht, Sn, 1.0, T1);          // T1 = Ht Sigma_n
        mm_abt<DM>(Xt, Xt, 1.0, Sig);
        mm_abt<DM>(T1, Ht, 1.0, Sig);        // Sigma_t = Xt Xt^T + Ht Sigma_n Ht^T
        st_row<DM>(wblk(a.Sigg, b, n, t, EF), d, lane, Sig);
        if (HAS_RHS) {
            const double y = ld_elem(wblk(a.yg, b, n, t, d), lane, d);
            const double v = y - mv<DM>(Gt, xn);
            xn = mv<DM>(Xt, v);
            if (lane < d) wblk(a.mug, b, n, t, d)[lane] = xn;
        }
#pragma unroll
        for (int k = 0; k < DM; ++k) Sn[k] = Sig[k];
    };
    for (int s = len - 2; s >= 0; --s) step(t0 + s, true);
    if (WANT_SUB && p > 0) step(t0 - 1, false);
}

// SSM parameters -> naturals / precision blocks, one wavefront per node (same outputs as k_ssm_to_naturals)
template <int DM, bool WANT_LIN>
static __global__ __launch_bounds__(64) void kw_ssm_to_naturals(int B, int T, int d, const double* __restrict__ Ag,
                                                        const double* __restrict__ offg, const double* __restrict__ cholg,
                                                        double cD, double cS, double* __restrict__ ling,
                                                        double* __restrict__ diagg, double* __restrict__ subg,
                                                        double* __restrict__ part_logdet) {
    __shared__ double tile[DM * (DM + 1)];
    const int lane = threadIdx.x, EF = d * d;
    const int b = blockIdx.x / T, t = blockIdx.x - b * T;
    auto qinv = [&](int node, double (&Qi)[DM]) {
        double C[DM], Xt[DM];
        ld_row<DM>(wblk(cholg, b, T, node, EF), d, lane, 1.0, C);
        if (lane >= d && lane < DM) C[lane] = 1.0;
        inv_t_rows<DM>(C, Xt, tile, lane);
#pragma unroll
        for (int k = 0; k < DM; ++k) Qi[k] = 0.0;
        mm_abt<DM>(Xt, Xt, 1.0, Qi);                 // (chol chol^T)^{-1} = X^T X
    };
    double Qi[DM];
    qinv(t, Qi);
    double lin = 0.0;
    if (WANT_LIN) lin = mv<DM>(Qi, ld_elem(wblk(offg, b, T, t, d), lane, d));
    if (t + 1 < T) {
        double Q1[DM], A[DM], At[DM], M[DM];
        qinv(t + 1, Q1);
        ld_row<DM>(wblk(Ag, b, T, t, EF), d, lane, 1.0, A);
        ld_col<DM>(wblk(Ag, b, T, t, EF), d, lane, 1.0, At);
#pragma unroll
        for (int k = 0; k < DM; ++k) M[k] = 0.0;
        mm_ab<DM>(Q1, A, 1.0, M);                     // Qi_{t+1} A
        mm_ab<DM>(At, M, 1.0, Qi);                    // + A^T Qi_{t+1} A
        if (WANT_LIN) {
            const double z1 = mv<DM>(Q1, ld_elem(wblk(offg, b, T, t + 1, d), lane, d));
            lin -= mv<DM>(At, z1);
        }
#pragma unroll
        for (int k = 0; k < DM; ++k) M[k] *= cS;
        st_row<DM>(wblk(subg, b, T, t, EF), d, lane, M);
    } else {
        double z[DM];
#pragma unroll
        for (int k = 0; k < DM; ++k) z[k] = 0.0;
        st_row<DM>(wblk(subg, b, T, t, EF), d, lane, z);
    }
#pragma unroll
    for (int k = 0; k < DM; ++k) Qi[k] *= cD;
    st_row<DM>(wblk(diagg, b, T, t, EF), d, lane, Qi);
    if (WANT_LIN && lane < d) wblk(ling, b, T, t, d)[lane] = lin;
    if (part_logdet) {
        double l = (lane < d) ? log(wblk(cholg, b, T, t, EF)[lane * d + lane]) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
        if (lane == 0) part_logdet[blockIdx.x] = l;
    }
}

// KL(q || p) local terms, one wavefront per node; partials part[b*T + t] (trace) and part[B*T + b*T + t] (Mahalanobis)
static __global__ __launch_bounds__(64) void kw_kl_terms(int B, int T, int d, const double* __restrict__ Sigg,
                                                 const double* __restrict__ Subg, const double* __restrict__ mug,
                                                 const double* __restrict__ Pdg, const double* __restrict__ Psg, double aD,
                                                 double aS, const double* __restrict__ mupg, double* __restrict__ part) {
    const int lane = threadIdx.x, EF = d * d;
    const int b = blockIdx.x / T, t = blockIdx.x - b * T;
    const double* m = wblk(mug, b, T, t, d);
    const double* mp = wblk(mupg, b, T, t, d);
    const double* Sg = wblk(Sigg, b, T, t, EF);
    const double* Pd = wblk(Pdg, b, T, t, EF);
    double tr = 0.0, mh = 0.0;
    for (int e = lane; e < EF; e += 64) {
        const int i = e / d, j = e - i * d;
        const double pd = aD * Pd[e];
        tr = __builtin_fma(pd, Sg[e], tr);
        mh = __builtin_fma(pd * (mp[i] - m[i]), mp[j] - m[j], mh);
    }
    if (t + 1 < T) {
        const double* m1 = wblk(mug, b, T, t + 1, d);
        const double* mp1 = wblk(mupg, b, T, t + 1, d);
        const double* Sb = wblk(Subg, b, T, t, EF);
        const double* Ps = wblk(Psg, b, T, t, EF);
        for (int e = lane; e < EF; e += 64) {
            const int i = e / d, j = e - i * d;
            const double ps = 2.0 * aS * Ps[e];
            tr = __builtin_fma(ps, Sb[e], tr);
            mh = __builtin_fma(ps * (mp1[i] - m1[i]), mp[j] - m[j], mh);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tr += __shfl_xor(tr, off, 64);
        mh += __shfl_xor(mh, off, 64);
    }
    if (lane == 0) {
        part[blockIdx.x] = tr;
        part[(size_t)B * T + blockIdx.x] = mh;
    }
}

}  // namespace mfgm
