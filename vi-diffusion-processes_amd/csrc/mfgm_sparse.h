// Local kernels of the sparse / inducing-state CVI model (markovflow/models/sparse_variational_cvi.py:140-221;
// conditionals.py:380-470; posterior.py:207-260), on natural-layout arrays, any state dimension d <= 32, one chain:
//
//   k_sparse_theta   : posterior naturals = prior naturals + the [M+1, 2d, 2d] sites overlap-added into the block-tri-diagonal
//                      structure (:160-172), in one pass
//   k_sparse_predict : q(f(t_i)) at the data points from the pairwise posterior marginals of the two inducing states around t_i
//                      (pairwise_marginals + base_conditional_predict + the projection onto f), without materialising the
//                      [M+1, 2d, 2d] pairwise covariances or any per-data-point [2d, 2d] tensor
//   k_sparse_sites   : data -> site sums  s_m = sum_{i in interval m} (g1_i w_i, g2_i w_i w_i^T)  (the reference's dynamic_partition +
//                      Python list of reduce_sums, :199-213) fused with the damped site update (:215-221)
//
// w_i = H P_i [2d] is the projection of data point i onto the pair of inducing states around it (conditionals.py:207-256); it and
// c_i = H T_i H^T depend only on the time points and the kernel, and are computed once per data set by the caller.
// Intervals: m = 0 .. M (M inducing points); interval m lies between inducing states m-1 and m; the data points are sorted in time,
// seg[m] .. seg[m+1] are those of interval m.  One workgroup per interval.
#pragma once
#include <hip/hip_runtime.h>

namespace mfgm {

struct SparseArgs {
    int M, d, N;
    int m_lo, m_hi;          // intervals this process owns (0, M + 1 unless one chain is shared between processes): seg, w, c and every
                             // per-data-point array hold the data points of these intervals only
    const int* seg;          // [m_hi - m_lo + 1] CSR offsets of the data points per owned interval
    const double* w;         // [N, 2d]
    const double* c;         // [N]
    const double* prior_mean;  // [d]   the kernel's initial mean (pads the chain at both ends)
    const double* prior_cov;   // [d, d] the kernel's initial covariance
};

// optional KL[q || p] terms taken in the same pass as the predictions (the pair covariance of an interval holds Sigma_m and Sigma_{m,m-1}):
// part[m] = trace term, part[M + 1 + m] = Mahalanobis term of node m / pair (m, m - 1), as kw_kl_terms (mfgm_wide.h) defines them
struct SparseKl {
    const double* Pd;        // [T, d, d] prior precision blocks (times aD)
    const double* Ps;        // [T, d, d] sub-diagonal blocks P_{t+1,t} at t (times aS)
    const double* mup;       // [T, d] prior marginal means
    double aD, aS;
    double* part;            // [2 (M + 1)] or NULL
};

// theta = prior naturals + overlap-added sites.  nat1 [M+1, 2d], nat2 [M+1, 2d, 2d]; plin [T, d], pdiag / psub [T, d, d] with T = M.
static __global__ void k_sparse_theta(int T, int d, const double* __restrict__ nat1, const double* __restrict__ nat2,
                                      const double* __restrict__ plin, const double* __restrict__ pdiag,
                                      const double* __restrict__ psub, double* __restrict__ lin, double* __restrict__ diag,
                                      double* __restrict__ sub) {
    const int d2 = 2 * d, dd = d * d;
    const size_t total = (size_t)T * dd;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t t = e / dd;
        const int r = (int)(e - t * dd), i = r / d, j = r - i * d;
        const double* n_hi = nat2 + (t + 1) * (size_t)(d2 * d2);       // site t+1: its first state is inducing state t
        const double* n_lo = nat2 + t * (size_t)(d2 * d2);             // site t:   its second state is inducing state t
        diag[e] = pdiag[e] + n_hi[i * d2 + j] + n_lo[(d + i) * d2 + (d + j)];
        sub[e] = (t + 1 < (size_t)T) ? psub[e] + 2.0 * n_hi[(d + i) * d2 + j] : 0.0;
        if (j == 0) lin[t * d + i] = (plin ? plin[t * d + i] : 0.0) + nat1[(t + 1) * d2 + i] + nat1[t * d2 + d + i];
    }
}

// marginals: mu [T, d], Sig [T, d, d] (full symmetric), Sub [T, d, d] (Sigma_{t+1,t} at t); fmu, fvar [N].
// One wavefront per interval, no LDS and no barriers: lane l owns column c = l mod 2d' of the pair covariance PC [2d, 2d] for the rows
// r = h, h + H, ... (h = l / 2d', H = 64 / 2d'; 2d' = 2d rounded up to a power of two <= 64), loaded straight from the marginal
// blocks, and sums  w_r PC[r][c] w_c  over its entries for every data point of the interval; one wavefront reduction per point.
template <int D2P>
static __global__ __launch_bounds__(64) void k_sparse_predict(SparseArgs a, const double* __restrict__ mu, const double* __restrict__ Sig,
                                                             const double* __restrict__ Sub, double* __restrict__ fmu,
                                                             double* __restrict__ fvar, SparseKl kl) {
    constexpr int H = (64 / D2P < D2P) ? 64 / D2P : D2P, NR = D2P / H;          // row groups (lanes beyond H D2P idle), rows per lane
    const int m = a.m_lo + blockIdx.x, d = a.d, d2 = 2 * d, lane = threadIdx.x;
    const int i0 = a.seg[blockIdx.x], i1 = a.seg[blockIdx.x + 1];
    if (i0 >= i1 && !kl.part) return;
    const int c = lane % D2P, h = lane / D2P;
    const bool lo_prior = (m == 0), hi_prior = (m == a.M);
    const double* S_lo = lo_prior ? a.prior_cov : Sig + (size_t)(m - 1) * d * d;
    const double* S_hi = hi_prior ? a.prior_cov : Sig + (size_t)m * d * d;
    const double* C = (lo_prior || hi_prior) ? nullptr : Sub + (size_t)(m - 1) * d * d;       // Cov(x_m, x_{m-1})
    double pc[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int r = h + H * k;
        double v = 0.0;
        if (h < H && r < d2 && c < d2) {
            if (r < d && c < d) v = S_lo[r * d + c];
            else if (r >= d && c >= d) v = S_hi[(r - d) * d + (c - d)];
            else if (r >= d) v = C ? C[(r - d) * d + c] : 0.0;                  // lower-left: Cov(x_hi, x_lo)
            else v = C ? C[(c - d) * d + r] : 0.0;                               // upper-right: its transpose
        }
        pc[k] = v;
    }
    double pmc = 0.0;                                   // pair mean, entry c (counted by the row group 0 only)
    if (h == 0 && c < d2) {
        const bool hi = c >= d;
        const int kk = hi ? c - d : c;
        pmc = hi ? (hi_prior ? a.prior_mean[kk] : mu[(size_t)m * d + kk]) : (lo_prior ? a.prior_mean[kk] : mu[(size_t)(m - 1) * d + kk]);
    }
    if (kl.part) {
        // node m: aD Pd_m . (Sigma_m + dv_m dv_m^T);  pair (m, m-1): 2 aS Ps_{m-1} . (Sigma_{m,m-1} + dv_m dv_{m-1}^T);  dv = mu_prior - mu
        double tr = 0.0, mh = 0.0;
        if (!hi_prior && c < d2) {
            const bool diag_col = c >= d;
            const int j = diag_col ? c - d : c;
            const bool has = diag_col || !lo_prior;
            const double* Pblk = diag_col ? kl.Pd + (size_t)m * d * d : kl.Ps + (size_t)(has ? m - 1 : 0) * d * d;
            const double sc = diag_col ? kl.aD : 2.0 * kl.aS;
            const size_t tj = diag_col ? (size_t)m : (size_t)(has ? m - 1 : 0);
            const double dvc = has ? kl.mup[tj * d + j] - mu[tj * d + j] : 0.0;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int r = h + H * k;
                if (h < H && r >= d && r < d2 && has) {
                    const int i = r - d;
                    const double pv = sc * Pblk[i * d + j];
                    const double dvr = kl.mup[(size_t)m * d + i] - mu[(size_t)m * d + i];
                    tr = __builtin_fma(pv, pc[k], tr);
                    mh = __builtin_fma(pv * dvr, dvc, mh);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tr += __shfl_down(tr, off, 64);
            mh += __shfl_down(mh, off, 64);
        }
        if (lane == 0) {
            kl.part[m] = tr;
            kl.part[a.M + 1 + m] = mh;
        }
    }
    for (int i = i0; i < i1; ++i) {
        const double* w = a.w + (size_t)i * d2;
        const double wc = (c < d2) ? w[c] : 0.0;
        double u = 0.0;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = h + H * k;
            u = __builtin_fma(pc[k], (h < H && r < d2) ? w[r] : 0.0, u);
        }
        double qv = wc * u, qm = wc * pmc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            qm += __shfl_down(qm, off, 64);
            qv += __shfl_down(qv, off, 64);
        }
        if (lane == 0) {
            fmu[i] = qm;
            fvar[i] = a.c[i] + qv;
        }
    }
}

// The same pass for even d with every load coalesced (the route of config 5; k_sparse_predict above stays for odd d).  One wavefront per
// interval; the three blocks of the pair covariance that matter -- Sigma_{m-1}, Sigma_m and the cross block C = Sigma_{m,m-1}: the upper
// right block is C^T, so  w^T PC w = w_lo^T S_lo w_lo + w_hi^T S_hi w_hi + 2 w_hi^T C w_lo  -- are read as 16-byte pairs, pair p = lane +
// 64 j of each block to lane `lane` (k_sparse_predict's column-per-lane mapping read C^T with a stride of a row: 16 sectors per request,
// and fetched w from memory inside the point loop; 0.88 ms at config 5, ~2 TB/s).  The KL terms are element-wise products with the
// prior's blocks in the same mapping; the vectors a lane needs by row / column (w, the pair mean, mu_prior - mu) go through LDS.
template <int NJ>
static __global__ __launch_bounds__(64) void k_sparse_predict_v(SparseArgs a, const double* __restrict__ mu, const double* __restrict__ Sig,
                                                               const double* __restrict__ Sub, double* __restrict__ fmu,
                                                               double* __restrict__ fvar, SparseKl kl) {
    __shared__ double sh_w[64], sh_pm[64], sh_dv[64];
    const int m = a.m_lo + blockIdx.x, d = a.d, d2 = 2 * d, dd = d * d, np = dd / 2, lane = threadIdx.x;
    const int i0 = a.seg[blockIdx.x], i1 = a.seg[blockIdx.x + 1];
    if (i0 >= i1 && !kl.part) return;
    const bool lo_prior = (m == 0), hi_prior = (m == a.M);
    const double2* S_lo = reinterpret_cast<const double2*>(lo_prior ? a.prior_cov : Sig + (size_t)(m - 1) * dd);
    const double2* S_hi = reinterpret_cast<const double2*>(hi_prior ? a.prior_cov : Sig + (size_t)m * dd);
    const double2* C = (lo_prior || hi_prior) ? nullptr : reinterpret_cast<const double2*>(Sub + (size_t)(m - 1) * dd);
    double2 lo[NJ], hi[NJ], cc[NJ];
    int row[NJ], col[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = lane + 64 * j;
        const bool ok = p < np;
        row[j] = ok ? (2 * p) / d : 0;
        col[j] = ok ? 2 * p - row[j] * d : 0;
        lo[j] = ok ? S_lo[p] : make_double2(0.0, 0.0);
        hi[j] = ok ? S_hi[p] : make_double2(0.0, 0.0);
        cc[j] = (ok && C) ? C[p] : make_double2(0.0, 0.0);
    }
    if (lane < d2) {
        const bool h = lane >= d;
        const int kk = h ? lane - d : lane;
        sh_pm[lane] = h ? (hi_prior ? a.prior_mean[kk] : mu[(size_t)m * d + kk]) : (lo_prior ? a.prior_mean[kk] : mu[(size_t)(m - 1) * d + kk]);
    }
    if (kl.part) {
        // node m: aD Pd_m . (Sigma_m + dv_m dv_m^T);  pair (m, m-1): 2 aS Ps_{m-1} . (Sigma_{m,m-1} + dv_m dv_{m-1}^T);  dv = mu_prior - mu
        double tr = 0.0, mh = 0.0;
        if (!hi_prior) {
            if (lane < d2) {
                const bool h = lane >= d;
                const int kk = h ? lane - d : lane;
                const size_t t = h ? (size_t)m : (size_t)(lo_prior ? 0 : m - 1);
                sh_dv[lane] = (h || !lo_prior) ? kl.mup[t * d + kk] - mu[t * d + kk] : 0.0;
            }
            __syncthreads();
            const double2* Pd = reinterpret_cast<const double2*>(kl.Pd + (size_t)m * dd);
            const double2* Ps = lo_prior ? nullptr : reinterpret_cast<const double2*>(kl.Ps + (size_t)(m - 1) * dd);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int p = lane + 64 * j;
                if (p < np) {
                    const double2 pd = Pd[p];
                    const double2 ps = Ps ? Ps[p] : make_double2(0.0, 0.0);
                    const double dr = sh_dv[d + row[j]];
                    tr += kl.aD * (pd.x * hi[j].x + pd.y * hi[j].y) + 2.0 * kl.aS * (ps.x * cc[j].x + ps.y * cc[j].y);
                    mh += dr * (kl.aD * (pd.x * sh_dv[d + col[j]] + pd.y * sh_dv[d + col[j] + 1])
                                + 2.0 * kl.aS * (ps.x * sh_dv[col[j]] + ps.y * sh_dv[col[j] + 1]));
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tr += __shfl_down(tr, off, 64);
            mh += __shfl_down(mh, off, 64);
        }
        if (lane == 0) {
            kl.part[m] = tr;
            kl.part[a.M + 1 + m] = mh;
        }
    }
    for (int i = i0; i < i1; ++i) {
        __syncthreads();                                   // the previous point's (and sh_pm's) readers / writers
        const double wl = (lane < d2) ? a.w[(size_t)i * d2 + lane] : 0.0;
        if (lane < d2) sh_w[lane] = wl;
        __syncthreads();
        double qv = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const double wlr = sh_w[row[j]], whr = sh_w[d + row[j]];
            const double wl0 = sh_w[col[j]], wl1 = sh_w[col[j] + 1], wh0 = sh_w[d + col[j]], wh1 = sh_w[d + col[j] + 1];
            qv += wlr * (lo[j].x * wl0 + lo[j].y * wl1) + whr * (hi[j].x * wh0 + hi[j].y * wh1) + 2.0 * whr * (cc[j].x * wl0 + cc[j].y * wl1);
        }
        double qm = (lane < d2) ? wl * sh_pm[lane] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            qm += __shfl_down(qm, off, 64);
            qv += __shfl_down(qv, off, 64);
        }
        if (lane == 0) {
            fmu[i] = qm;
            fvar[i] = a.c[i] + qv;
        }
    }
}

// State prediction at arbitrary sorted query points (conditional_predict + base_conditional_predict, conditionals.py:29-76, 380-421;
// posterior.py:207-229): p(x(t_i)) = N(P_i m_pair, T_i + P_i S_pair P_i^T) with the pair of conditioning states around t_i taken
// straight from the marginal blocks (the prior pads both ends).  Pm [N, d, 2d], Tm [N, d, d], idx [N] = interval of query i;
// out_mean [N, d], out_cov [N, d, d].  One wavefront per query point.
static __global__ __launch_bounds__(64) void k_cond_predict(int M, int d, int N, const int* __restrict__ idx, const double* __restrict__ Pm,
                                                           const double* __restrict__ Tm, const double* __restrict__ prior_mean,
                                                           const double* __restrict__ prior_cov, const double* __restrict__ mu,
                                                           const double* __restrict__ Sig, const double* __restrict__ Sub,
                                                           double* __restrict__ out_mean, double* __restrict__ out_cov) {
    extern __shared__ double sh[];     // PC [2d][2d], pm [2d], P [d][2d], U = P PC [d][2d]
    const int i = blockIdx.x, d2 = 2 * d, lane = threadIdx.x;
    if (i >= N) return;
    const int m = idx[i];
    double* PC = sh;
    double* pm = PC + d2 * d2;
    double* P = pm + d2;
    double* U = P + d * d2;
    const bool lo_prior = (m == 0), hi_prior = (m == M);
    const double* S_lo = lo_prior ? prior_cov : Sig + (size_t)(m - 1) * d * d;
    const double* S_hi = hi_prior ? prior_cov : Sig + (size_t)m * d * d;
    const double* C = (lo_prior || hi_prior) ? nullptr : Sub + (size_t)(m - 1) * d * d;
    for (int e = lane; e < d2 * d2; e += 64) {
        const int r = e / d2, c = e - r * d2;
        double v;
        if (r < d && c < d) v = S_lo[r * d + c];
        else if (r >= d && c >= d) v = S_hi[(r - d) * d + (c - d)];
        else if (r >= d) v = C ? C[(r - d) * d + c] : 0.0;
        else v = C ? C[(c - d) * d + r] : 0.0;
        PC[e] = v;
    }
    for (int e = lane; e < d2; e += 64) {
        const bool hi = e >= d;
        const int k = hi ? e - d : e;
        pm[e] = hi ? (hi_prior ? prior_mean[k] : mu[(size_t)m * d + k]) : (lo_prior ? prior_mean[k] : mu[(size_t)(m - 1) * d + k]);
    }
    for (int e = lane; e < d * d2; e += 64) P[e] = Pm[(size_t)i * d * d2 + e];
    __syncthreads();
    for (int e = lane; e < d * d2; e += 64) {         // U = P PC
        const int r = e / d2, c = e - r * d2;
        double acc = 0.0;
        for (int k = 0; k < d2; ++k) acc += P[r * d2 + k] * PC[k * d2 + c];
        U[e] = acc;
    }
    for (int e = lane; e < d; e += 64) {
        double acc = 0.0;
        for (int k = 0; k < d2; ++k) acc += P[e * d2 + k] * pm[k];
        out_mean[(size_t)i * d + e] = acc;
    }
    __syncthreads();
    for (int e = lane; e < d * d; e += 64) {          // cov = T + U P^T
        const int r = e / d, c = e - r * d;
        double acc = Tm[(size_t)i * d * d + e];
        for (int k = 0; k < d2; ++k) acc += U[r * d2 + k] * P[c * d2 + k];
        out_cov[(size_t)i * d * d + e] = acc;
    }
}

// sites <- (1 - lr) sites + lr (sum_i g1_i w_i, sum_i g2_i w_i w_i^T), in place; g1, g2 [N].  A streaming read-modify-write of the
// [M + 1, 2d, 2d] site tensor: what bounds it is the number of bytes a compute unit keeps in flight, so a workgroup of 256 threads
// takes G = 8 / SPI consecutive intervals (SPI = slots of 256 entry pairs per [2d, 2d] block: 2 for 2d = 32), requests all their old
// values (16-byte accesses, 8 per thread) before anything else, and only then walks the data points of each interval (w, g1, g2 of up
// to kSitesChunk points staged through LDS at a time; per-thread gathers of w from memory were measured 1.6x slower).
constexpr int kSitesChunk = 8;
template <int SPI>
static __global__ __launch_bounds__(256) void k_sparse_sites(SparseArgs a, const double* __restrict__ g1, const double* __restrict__ g2,
                                                            double lr, double* __restrict__ nat1, double* __restrict__ nat2) {
    constexpr int G = 8 / SPI;
    extern __shared__ double sh[];     // kSitesChunk x (w [2d], g1, g2)
    const int d2 = 2 * a.d, tid = threadIdx.x, m0 = a.m_lo + blockIdx.x * G;
    const int ne = d2 * d2;            // even, and a row never splits a pair
    double2 old[8], acc[8];
    int rr[SPI], cc[SPI];
#pragma unroll
    for (int kk = 0; kk < SPI; ++kk) {
        const int e = 2 * (tid + kk * 256);
        rr[kk] = (e < ne) ? e / d2 : 0;
        cc[kk] = (e < ne) ? e - rr[kk] * d2 : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int gi = k / SPI, p = tid + (k % SPI) * 256;
        const bool ok = (m0 + gi < a.m_hi) && 2 * p < ne;
        old[k] = ok ? reinterpret_cast<const double2*>(nat2 + (size_t)(m0 + gi) * ne)[p] : make_double2(0.0, 0.0);
        acc[k] = make_double2(0.0, 0.0);
    }
    double old1[G], acc1[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        old1[gi] = (tid < d2 && m0 + gi < a.m_hi) ? nat1[(size_t)(m0 + gi) * d2 + tid] : 0.0;
        acc1[gi] = 0.0;
    }
    const int st = d2 + 2;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const int m = min(m0 + gi, a.m_hi - 1) - a.m_lo;
        const int i0 = a.seg[m], i1 = (m0 + gi < a.m_hi) ? a.seg[m + 1] : i0;
        for (int c0 = i0; c0 < i1; c0 += kSitesChunk) {
            const int np = min(kSitesChunk, i1 - c0);
            __syncthreads();
            for (int e = tid; e < np * d2; e += 256) {
                const int pt = e / d2, j = e - pt * d2;
                sh[pt * st + j] = a.w[(size_t)(c0 + pt) * d2 + j];
            }
            if (tid < np) { sh[tid * st + d2] = g1[c0 + tid]; sh[tid * st + d2 + 1] = g2[c0 + tid]; }
            __syncthreads();
            for (int pt = 0; pt < np; ++pt) {
                const double* w = sh + pt * st;
                const double gg = w[d2 + 1];
#pragma unroll
                for (int kk = 0; kk < SPI; ++kk) {
                    const double gr = gg * w[rr[kk]];
                    acc[gi * SPI + kk].x = __builtin_fma(gr, w[cc[kk]], acc[gi * SPI + kk].x);
                    acc[gi * SPI + kk].y = __builtin_fma(gr, w[cc[kk] + 1], acc[gi * SPI + kk].y);
                }
                if (tid < d2) acc1[gi] = __builtin_fma(w[d2], w[tid], acc1[gi]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int gi = k / SPI, p = tid + (k % SPI) * 256;
        if ((m0 + gi < a.m_hi) && 2 * p < ne)
            reinterpret_cast<double2*>(nat2 + (size_t)(m0 + gi) * ne)[p] =
                make_double2(__builtin_fma(lr, acc[k].x, (1.0 - lr) * old[k].x), __builtin_fma(lr, acc[k].y, (1.0 - lr) * old[k].y));
    }
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
        if (tid < d2 && m0 + gi < a.m_hi) nat1[(size_t)(m0 + gi) * d2 + tid] = __builtin_fma(lr, acc1[gi], (1.0 - lr) * old1[gi]);
}

// The same update on the QUADRANT-PACKED site tensor nat2q [M + 1, QS], QS = 2 ET + EF (WideArgs::site_packed, mfgm_wide.h: upper-left
// block as a packed lower triangle, lower-left block in full, lower-right block as a packed lower triangle; the upper-right block of
// the symmetric [2d, 2d] site is the transpose of the lower-left one and is not stored): 528 instead of 1 024 doubles per site at
// d = 16, for this read-modify-write and for the two factor passes that read the sites.  A workgroup of 256 threads takes kSitesQG
// consecutive intervals; thread tid owns the packed entries tid, tid + 256, ... of each.
constexpr int kSitesQG = 4;          // sites whose old values a workgroup has in flight together
constexpr int kSitesQRounds = 8;     // rounds of kSitesQG sites per workgroup: the index decode and the launch are paid once per 32 sites
constexpr int kSitesQChunk = 32;     // data points staged through LDS at a time
template <int NE>           // packed entries per thread and site: ceil(QS / 256)
static __global__ __launch_bounds__(256) void k_sparse_sites_q(SparseArgs a, const double* __restrict__ g1, const double* __restrict__ g2,
                                                              double lr, double* __restrict__ nat1, double* __restrict__ nat2q) {
    extern __shared__ double sh[];     // kSitesChunk x (w [2d], g1, g2)
    const int d = a.d, d2 = 2 * d, ET = d * (d + 1) / 2, EF = d * d, QS = 2 * ET + EF;
    const int tid = threadIdx.x;
    int rr[NE], cc[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = tid + k * 256;
        int r = 0, c = 0;
        if (e < QS) {
            if (e >= ET && e < ET + EF) {
                r = d + (e - ET) / d;
                c = (e - ET) % d;
            } else {
                const int t = (e < ET) ? e : e - ET - EF;
                int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
                while (i * (i + 1) / 2 > t) --i;
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                const int j = t - i * (i + 1) / 2;
                r = (e < ET) ? i : d + i;
                c = (e < ET) ? j : d + j;
            }
        }
        rr[k] = r;
        cc[k] = c;
    }
    const int st = d2 + 2;
    double old[kSitesQG][NE], nxt[kSitesQG][NE], old1[kSitesQG], nxt1[kSitesQG];
    auto load = [&](int m0, double (&o)[kSitesQG][NE], double (&o1)[kSitesQG]) {
#pragma unroll
        for (int gi = 0; gi < kSitesQG; ++gi) {
            const bool own = (m0 + gi < a.m_hi);
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int e = tid + k * 256;
                o[gi][k] = (own && e < QS) ? nat2q[(size_t)(m0 + gi) * QS + e] : 0.0;
            }
            o1[gi] = (own && tid < d2) ? nat1[(size_t)(m0 + gi) * d2 + tid] : 0.0;
        }
    };
    const int mbase = a.m_lo + blockIdx.x * (kSitesQG * kSitesQRounds);
    load(mbase, old, old1);
    for (int rd = 0; rd < kSitesQRounds; ++rd) {
        const int m0 = mbase + rd * kSitesQG;
        if (m0 >= a.m_hi) break;
        // the next round's old values are requested before this round's data points are walked
        if (rd + 1 < kSitesQRounds) load(m0 + kSitesQG, nxt, nxt1);
        double acc[kSitesQG][NE], acc1[kSitesQG];
#pragma unroll
        for (int gi = 0; gi < kSitesQG; ++gi) {
#pragma unroll
            for (int k = 0; k < NE; ++k) acc[gi][k] = 0.0;
            acc1[gi] = 0.0;
        }
        // the data points of the round's sites are one contiguous range (the intervals are consecutive): staged through LDS together,
        // kSitesQChunk at a time -- one pair of barriers per round instead of one per site (with a load -> LDS -> barrier chain per
        // site the kernel ran at the latency of 32 dependent round trips per workgroup, 2.7 TB/s)
        int segv[kSitesQG + 1];
#pragma unroll
        for (int gi = 0; gi <= kSitesQG; ++gi) segv[gi] = a.seg[min(m0 + gi, a.m_hi) - a.m_lo];
        for (int c0 = segv[0]; c0 < segv[kSitesQG]; c0 += kSitesQChunk) {
            const int np = min(kSitesQChunk, segv[kSitesQG] - c0);
            __syncthreads();
            for (int e = tid; e < np * d2; e += 256) {
                const int pt = e / d2, j = e - pt * d2;
                sh[pt * st + j] = a.w[(size_t)(c0 + pt) * d2 + j];
            }
            if (tid < np) { sh[tid * st + d2] = g1[c0 + tid]; sh[tid * st + d2 + 1] = g2[c0 + tid]; }
            __syncthreads();
#pragma unroll
            for (int gi = 0; gi < kSitesQG; ++gi) {
                const int p0 = max(segv[gi], c0) - c0, p1 = min(segv[gi + 1], c0 + np) - c0;
                for (int pt = p0; pt < p1; ++pt) {
                    const double* w = sh + pt * st;
                    const double gg = w[d2 + 1];
#pragma unroll
                    for (int k = 0; k < NE; ++k) acc[gi][k] = __builtin_fma(gg * w[rr[k]], w[cc[k]], acc[gi][k]);
                    if (tid < d2) acc1[gi] = __builtin_fma(w[d2], w[tid], acc1[gi]);
                }
            }
        }
#pragma unroll
        for (int gi = 0; gi < kSitesQG; ++gi) {
            if (m0 + gi < a.m_hi) {
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    const int e = tid + k * 256;
                    if (e < QS) nat2q[(size_t)(m0 + gi) * QS + e] = __builtin_fma(lr, acc[gi][k], (1.0 - lr) * old[gi][k]);
                }
                if (tid < d2) nat1[(size_t)(m0 + gi) * d2 + tid] = __builtin_fma(lr, acc1[gi], (1.0 - lr) * old1[gi]);
            }
#pragma unroll
            for (int k = 0; k < NE; ++k) old[gi][k] = nxt[gi][k];
            old1[gi] = nxt1[gi];
        }
    }
}

}  // namespace mfgm
