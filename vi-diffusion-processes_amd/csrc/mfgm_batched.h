// Batched small dense SPD algebra on natural-layout arrays [N, d, d] (d <= 32): Cholesky and triangular solves, one
// thread per matrix (per right-hand-side column for the solves).  These serve the per-time-step algebra around the
// sweeps (the reference's tf.linalg.cholesky / cholesky_solve / triangular_solve calls, e.g.
// ssm_gaussian_transformations.py:93-178, 515-593; conditionals.py:207-256) that has no fused kernel of its own, so that
// no part of the path depends on a vendor batched-LAPACK; sqrt and division are IEEE here (accuracy over speed).
#pragma once
#include "mfgm_math.h"
#include "mfgm_wide.h"   // row-per-lane primitives (ld_row, chol_rsolve) for the wavefront-per-matrix variant

namespace mfgm {

// L = chol(A) (lower; strict upper written as zero).  A non-positive pivot sets *info and is replaced by 1.
static __global__ __launch_bounds__(128) void k_batched_chol(int N, int d, const double* __restrict__ A, double* __restrict__ L,
                                                      int* info) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double* a = A + (size_t)n * d * d;
    double* l = L + (size_t)n * d * d;
    int bad = 0;
    for (int j = 0; j < d; ++j) {
        double s = a[j * d + j];
        for (int k = 0; k < j; ++k) s = __builtin_fma(-l[j * d + k], l[j * d + k], s);
        if (!(s > 0.0)) { bad = 1; s = 1.0; }
        const double ljj = sqrt(s);
        l[j * d + j] = ljj;
        for (int i = 0; i < j; ++i) l[i * d + j] = 0.0;
        for (int i = j + 1; i < d; ++i) {
            double t = a[i * d + j];
            for (int k = 0; k < j; ++k) t = __builtin_fma(-l[i * d + k], l[j * d + k], t);
            l[i * d + j] = t / ljj;
        }
    }
    if (bad) atomicMax(info, 1);
}

// Wavefront-per-matrix Cholesky for 8 < d <= 32: lane i holds row i (coalesced row loads / stores), pivots are broadcast with
// v_readlane; 4 matrices per 256-thread block.
template <int DM>
static __global__ __launch_bounds__(256) void k_batched_chol_wave(int N, int d, const double* __restrict__ A, double* __restrict__ L,
                                                                  int* info) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;                       // whole wavefronts leave together
    double F[DM];
    ld_row<DM>(A + (size_t)n * d * d, d, lane, 1.0, F);
    if (lane >= d && lane < DM) F[lane] = 1.0;
    int bad = 0;
#pragma unroll
    for (int j = 0; j < DM; ++j) {                  // row Cholesky-Crout, IEEE sqrt / division as in k_batched_chol
        double acc = F[j];
#pragma unroll
        for (int k = 0; k < j; ++k) acc = __builtin_fma(-F[k], bcast(F[k], j), acc);
        double piv = bcast(acc, j);
        if (!(piv > 0.0)) { bad = 1; piv = 1.0; }
        const double ljj = sqrt(piv);
        F[j] = (lane > j) ? acc / ljj : ((lane == j) ? ljj : 0.0);
    }
    st_row<DM>(L + (size_t)n * d * d, d, lane, F);
    if (bad && lane == 0) atomicMax(info, 1);
}

// X = op(L) B for B [N, d, m]:  mode 1: L^{-1} B,  2: L^{-T} B,  3: (L L^T)^{-1} B.  lbatch = 1 shares one L.
static __global__ __launch_bounds__(128) void k_batched_trsm(int N, int d, int m, int lbatch, const double* __restrict__ L,
                                                      const double* __restrict__ B, double* __restrict__ X, int mode) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)N * m) return;
    const int n = (int)(id / m), c = (int)(id - (long long)n * m);
    const double* l = L + (lbatch == 1 ? 0 : (size_t)n * d * d);
    const double* b = B + (size_t)n * d * m + c;
    double* x = X + (size_t)n * d * m + c;
    for (int i = 0; i < d; ++i) x[i * m] = b[i * m];
    if (mode & 1) {
        for (int i = 0; i < d; ++i) {
            double t = x[i * m];
            for (int k = 0; k < i; ++k) t = __builtin_fma(-l[i * d + k], x[k * m], t);
            x[i * m] = t / l[i * d + i];
        }
    }
    if (mode & 2) {
        for (int i = d - 1; i >= 0; --i) {
            double t = x[i * m];
            for (int k = i + 1; k < d; ++k) t = __builtin_fma(-l[k * d + i], x[k * m], t);
            x[i * m] = t / l[i * d + i];
        }
    }
}


// Triangular solves for wide right-hand sides (m >= 8): L staged in LDS, one right-hand-side column per thread held in
// registers; loads and stores of B / X are coalesced across the columns.  A 64-thread block serves 64 / CP matrices with CP
// column slots each (CP = 16, 32 or 64, the smallest that covers m, or 64 with a column loop).
template <int DM, int CP>
static __global__ __launch_bounds__(64) void k_batched_trsm_cols(int N, int d, int m, int lbatch, const double* __restrict__ L,
                                                                 const double* __restrict__ B, double* __restrict__ X, int mode) {
    constexpr int MPB = 64 / CP;                                 // matrices per block
    __shared__ double sl_all[MPB * DM * (DM + 1)];
    const int sub = threadIdx.x / CP, c0 = threadIdx.x - sub * CP;
    const int n = blockIdx.x * MPB + sub;
    double* sl = sl_all + sub * DM * (DM + 1);
    if (n < N) {
        const double* l = L + (lbatch == 1 ? 0 : (size_t)n * d * d);
        for (int e = c0; e < d * d; e += CP) sl[(e / d) * (DM + 1) + (e % d)] = l[e];
    }
    __syncthreads();
    if (n >= N) return;
    for (int c = c0; c < m; c += CP) {
        const double* b = B + (size_t)n * d * m + c;
        double* x = X + (size_t)n * d * m + c;
        double v[DM];
#pragma unroll
        for (int i = 0; i < DM; ++i) v[i] = (i < d) ? b[(size_t)i * m] : 0.0;
        if (mode & 1) {
#pragma unroll
            for (int i = 0; i < DM; ++i) {
                if (i < d) {
                    double t = v[i];
#pragma unroll
                    for (int k = 0; k < i; ++k) t = __builtin_fma(-sl[i * (DM + 1) + k], v[k], t);
                    v[i] = t / sl[i * (DM + 1) + i];
                }
            }
        }
        if (mode & 2) {
#pragma unroll
            for (int i = DM - 1; i >= 0; --i) {
                if (i < d) {
                    double t = v[i];
#pragma unroll
                    for (int k = i + 1; k < DM; ++k)
                        if (k < d) t = __builtin_fma(-sl[k * (DM + 1) + i], v[k], t);
                    v[i] = t / sl[i * (DM + 1) + i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DM; ++i)
            if (i < d) x[(size_t)i * m] = v[i];
    }
}

}  // namespace mfgm
