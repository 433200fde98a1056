// Direct solve with a lower block-bidiagonal factor (LowerTriangularBlockTriDiagonal.solve, block_tri_diag.py:339-351 ->
// solve_triang_mat):  L x = r  or  L^T x = r  for  L = blockbidiag(Ld_t lower-triangular, Ls_t = L_{t+1,t}), natural layout, d <= 32.
//
// The substitution  x_t = Ld_t^{-1} (r_t - Ls_{t-1} x_{t-1})  is an affine recurrence, parallelised exactly over segments of
// kBidiagR nodes: (1) every segment composes its affine map x_out = Phi x_in + c (one wavefront per segment, lane k carries column k
// of [Phi | c] through the segment), (2) one wavefront per chain chains the maps, (3) every segment replays the recurrence from
// its incoming value.  Unlike the route through the Gram matrix L L^T (which squares the condition number), this is the plain
// substitution: its error is that of a sequential triangular solve.
#pragma once
#include <hip/hip_runtime.h>

namespace mfgm {

constexpr int kBidiagR = 32;

// MODE 0: segment maps (Phi [B, P, d, d], cvec [B, P, d]);  MODE 1: replay from xin [B, P, d], writing x [B, T, d]
template <int DM, bool TRANS, int MODE>
static __global__ __launch_bounds__(64) void k_bidiag(int T, int d, int P, const double* __restrict__ Ld, const double* __restrict__ Ls,
                                                     const double* __restrict__ r, double* __restrict__ Phi, double* __restrict__ cvec,
                                                     const double* __restrict__ xin, double* __restrict__ x) {
    __shared__ double sLd[DM * DM], sLs[DM * DM], sr[DM];
    const int seg = blockIdx.x, b = seg / P, p = seg - b * P, k = threadIdx.x;
    const int t0 = p * kBidiagR, len = min(kBidiagR, T - t0);
    const size_t dd = (size_t)d * d;
    double v[DM];
#pragma unroll
    for (int i = 0; i < DM; ++i) v[i] = (MODE == 0) ? ((i == k && k < d) ? 1.0 : 0.0) : ((k == d && i < d) ? xin[((size_t)b * P + p) * d + i] : 0.0);
    for (int s = 0; s < len; ++s) {
        const int t = TRANS ? t0 + len - 1 - s : t0 + s;
        const bool coupled = TRANS ? (t + 1 < T) : (t > 0);
        const double* Ldt = Ld + ((size_t)b * T + t) * dd;
        const double* Lst = (coupled && Ls) ? Ls + ((size_t)b * (T - 1) + (TRANS ? t : t - 1)) * dd : nullptr;
        __syncthreads();
        for (int e = k; e < DM * DM; e += 64) {
            const int i = e / DM, j = e - i * DM;
            const bool in = (i < d && j < d);
            sLd[e] = in ? (j <= i ? Ldt[i * d + j] : 0.0) : (i == j ? 1.0 : 0.0);
            sLs[e] = (in && Lst) ? Lst[i * d + j] : 0.0;
        }
        if (k < DM) sr[k] = (k < d) ? r[((size_t)b * T + t) * d + k] : 0.0;
        __syncthreads();
        if (k <= d) {
            double z[DM];
#pragma unroll
            for (int i = 0; i < DM; ++i) {
                double acc = (k == d) ? sr[i] : 0.0;
#pragma unroll
                for (int j = 0; j < DM; ++j) acc -= (TRANS ? sLs[j * DM + i] : sLs[i * DM + j]) * v[j];
                z[i] = acc;
            }
            if (!TRANS) {
#pragma unroll
                for (int i = 0; i < DM; ++i) {
                    double acc = z[i];
#pragma unroll
                    for (int j = 0; j < i; ++j) acc -= sLd[i * DM + j] * v[j];
                    v[i] = acc / sLd[i * DM + i];
                }
            } else {
#pragma unroll
                for (int i = DM - 1; i >= 0; --i) {
                    double acc = z[i];
#pragma unroll
                    for (int j = i + 1; j < DM; ++j) acc -= sLd[j * DM + i] * v[j];
                    v[i] = acc / sLd[i * DM + i];
                }
            }
            if (MODE == 1 && k == d) {
#pragma unroll
                for (int i = 0; i < DM; ++i)
                    if (i < d) x[((size_t)b * T + t) * d + i] = v[i];
            }
        }
    }
    if (MODE == 0 && k <= d) {
#pragma unroll
        for (int i = 0; i < DM; ++i) {
            if (i < d) {
                if (k < d) Phi[(((size_t)b * P + p) * d + i) * d + k] = v[i];
                else cvec[((size_t)b * P + p) * d + i] = v[i];
            }
        }
    }
}

// xin of every segment: forward  xin[0] = 0, xin[p+1] = Phi_p xin[p] + c_p;  transposed: from the last segment down
template <bool TRANS>
static __global__ __launch_bounds__(64) void k_bidiag_scan(int d, int P, const double* __restrict__ Phi, const double* __restrict__ cvec,
                                                          double* __restrict__ xin) {
    __shared__ double sx[32];
    const int b = blockIdx.x, i = threadIdx.x;
    if (i < 32) sx[i] = 0.0;
    __syncthreads();
    for (int q = 0; q < P; ++q) {
        const int p = TRANS ? P - 1 - q : q;
        if (i < d) xin[((size_t)b * P + p) * d + i] = sx[i];
        double acc = 0.0;
        if (i < d) {
            acc = cvec[((size_t)b * P + p) * d + i];
            const double* row = Phi + (((size_t)b * P + p) * d + i) * d;
            for (int j = 0; j < d; ++j) acc += row[j] * sx[j];
        }
        __syncthreads();
        if (i < d) sx[i] = acc;
        __syncthreads();
    }
}

}  // namespace mfgm
