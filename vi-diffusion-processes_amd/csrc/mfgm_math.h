// Register-resident dense d x d fp64 block algebra for the lane-per-chain-segment kernels.
// Everything is fully unrolled on the compile-time block size D so the arrays live in VGPRs.
//
// Storage conventions
//   full  : row-major  a[i*D + j]
//   tri   : packed lower triangle  a[i*(i+1)/2 + j], j <= i   (symmetric matrices and Cholesky factors)
#pragma once
#include <hip/hip_runtime.h>

#define MFGM_NTRI(D) ((D) * ((D) + 1) / 2)
#define MFGM_DEV __device__ __forceinline__
#define MFGM_HD __host__ __device__ inline

namespace mfgm {

// Not-positive-definite report.  `info` keeps the FIRST failing location -- the smallest code = 1 + lane + (level << 27), lane = chain *
// segments + segment at that level -- stored as INT_MAX - code so that one atomicMax does it.  0: no failure; 1: flagged by a kernel
// that has no location to give.  mfgm_plan_decode_info turns the word into (chain, node range).
MFGM_DEV void flag_not_pd(int* info, int level, int lane) { atomicMax(info, 0x7fffffff - (1 + lane + (level << 27))); }

MFGM_DEV constexpr int tix(int i, int j) { return i * (i + 1) / 2 + j; }          // j <= i
MFGM_DEV constexpr int six(int i, int j) { return i >= j ? tix(i, j) : tix(j, i); }  // symmetric access

// 1/sqrt(x) in fp64: hardware estimate + two Newton steps (~1 ulp), no division.
MFGM_DEV double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
}

// 1/x in fp64: hardware estimate + two Newton steps.
MFGM_DEV double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
}

// In-place lower Cholesky of the SPD matrix held (lower triangle) in a[]; invd[j] = 1/L_jj.
// A non-positive (or NaN) pivot sets bad = 1 and is replaced by 1 so the sweep stays finite.
template <int D>
MFGM_DEV void chol_inplace(double (&a)[MFGM_NTRI(D)], double (&invd)[D], int& bad) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double s = a[tix(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s = __builtin_fma(-a[tix(j, k)], a[tix(j, k)], s);
        if (!(s > 0.0)) { bad = 1; s = 1.0; }
        double inv = rsqrt_nr(s);
        invd[j] = inv;
        a[tix(j, j)] = s * inv;
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double t = a[tix(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) t = __builtin_fma(-a[tix(i, k)], a[tix(j, k)], t);
            a[tix(i, j)] = t * inv;
        }
    }
}

// v := L^{-1} v
template <int D>
MFGM_DEV void trsv_lower(const double (&L)[MFGM_NTRI(D)], const double (&invd)[D], double (&v)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double t = v[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t = __builtin_fma(-L[tix(i, k)], v[k], t);
        v[i] = t * invd[i];
    }
}

// v := L^{-T} v
template <int D>
MFGM_DEV void trsv_lower_t(const double (&L)[MFGM_NTRI(D)], const double (&invd)[D], double (&v)[D]) {
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
        double t = v[i];
#pragma unroll
        for (int k = i + 1; k < D; ++k) t = __builtin_fma(-L[tix(k, i)], v[k], t);
        v[i] = t * invd[i];
    }
}

// M := L^{-1} M   (full D x D, column by column)
template <int D>
MFGM_DEV void trsm_left_lower(const double (&L)[MFGM_NTRI(D)], const double (&invd)[D], double (&M)[D * D]) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double t = M[i * D + c];
#pragma unroll
            for (int k = 0; k < i; ++k) t = __builtin_fma(-L[tix(i, k)], M[k * D + c], t);
            M[i * D + c] = t * invd[i];
        }
    }
}

// G := G L^{-T}   (each row g solves L g^T = s^T by forward substitution)
template <int D>
MFGM_DEV void trsm_right_lower_t(const double (&L)[MFGM_NTRI(D)], const double (&invd)[D], double (&G)[D * D]) {
#pragma unroll
    for (int r = 0; r < D; ++r) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double t = G[r * D + i];
#pragma unroll
            for (int k = 0; k < i; ++k) t = __builtin_fma(-L[tix(i, k)], G[r * D + k], t);
            G[r * D + i] = t * invd[i];
        }
    }
}

// C(sym, tri-packed) := alpha * G G^T + beta-less overwrite:  C = G G^T
template <int D>
MFGM_DEV void syrk_set(const double (&G)[D * D], double (&C)[MFGM_NTRI(D)]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(G[i * D + k], G[j * D + k], t);
            C[tix(i, j)] = t;
        }
}

// C(sym) += W^T W
template <int D>
MFGM_DEV void syrk_t_acc(const double (&W)[D * D], double (&C)[MFGM_NTRI(D)]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = C[tix(i, j)];
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(W[k * D + i], W[k * D + j], t);
            C[tix(i, j)] = t;
        }
}

// out := G v
template <int D>
MFGM_DEV void gemv(const double (&G)[D * D], const double (&v)[D], double (&out)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t = __builtin_fma(G[i * D + k], v[k], t);
        out[i] = t;
    }
}

// out := G^T v
template <int D>
MFGM_DEV void gemv_t(const double (&G)[D * D], const double (&v)[D], double (&out)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t = __builtin_fma(G[k * D + i], v[k], t);
        out[i] = t;
    }
}

// out := A B (full)
template <int D>
MFGM_DEV void gemm(const double (&A)[D * D], const double (&B)[D * D], double (&out)[D * D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(A[i * D + k], B[k * D + j], t);
            out[i * D + j] = t;
        }
}

// Linv := L^{-1} (lower-triangular, packed) given invd
template <int D>
MFGM_DEV void tri_inverse(const double (&L)[MFGM_NTRI(D)], const double (&invd)[D], double (&X)[MFGM_NTRI(D)]) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
        X[tix(c, c)] = invd[c];
#pragma unroll
        for (int i = c + 1; i < D; ++i) {
            double t = 0.0;
#pragma unroll
            for (int k = c; k < i; ++k) t = __builtin_fma(-L[tix(i, k)], X[tix(k, c)], t);
            X[tix(i, c)] = t * invd[i];
        }
    }
}

// out(full) := G X, X lower-triangular packed:  out[r][c] = sum_{k>=c} G[r][k] X[k][c]
template <int D>
MFGM_DEV void gemm_full_tri(const double (&G)[D * D], const double (&X)[MFGM_NTRI(D)], double (&out)[D * D]) {
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double t = 0.0;
#pragma unroll
            for (int k = c; k < D; ++k) t = __builtin_fma(G[r * D + k], X[tix(k, c)], t);
            out[r * D + c] = t;
        }
}

// out(full) := S H, S symmetric packed
template <int D>
MFGM_DEV void gemm_sym_full(const double (&S)[MFGM_NTRI(D)], const double (&H)[D * D], double (&out)[D * D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(S[six(i, k)], H[k * D + j], t);
            out[i * D + j] = t;
        }
}

// C(sym) := X^T X, X lower-triangular packed: C[i][j] = sum_{k>=max(i,j)} X[k][i] X[k][j]
template <int D>
MFGM_DEV void tri_t_tri(const double (&X)[MFGM_NTRI(D)], double (&C)[MFGM_NTRI(D)]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = i; k < D; ++k) t = __builtin_fma(X[tix(k, i)], X[tix(k, j)], t);
            C[tix(i, j)] = t;
        }
}

// C(sym) += sign * A^T B restricted to the lower triangle (caller guarantees the product is symmetric)
template <int D>
MFGM_DEV void gemm_tn_sym_acc(const double (&A)[D * D], const double (&B)[D * D], double sign, double (&C)[MFGM_NTRI(D)]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(A[k * D + i], B[k * D + j], t);
            C[tix(i, j)] = __builtin_fma(sign, t, C[tix(i, j)]);
        }
}


// X := X L^{-1}   (each row x solves a L = x by backward substitution over the columns)
template <int D>
MFGM_DEV void trsm_right_lower(const double (&L)[MFGM_NTRI(D)], const double (&invd)[D], double (&X)[D * D]) {
#pragma unroll
    for (int r = 0; r < D; ++r) {
#pragma unroll
        for (int j = D - 1; j >= 0; --j) {
            double t = X[r * D + j];
#pragma unroll
            for (int k = j + 1; k < D; ++k) t = __builtin_fma(-X[r * D + k], L[tix(k, j)], t);
            X[r * D + j] = t * invd[j];
        }
    }
}

// log of a product of positive numbers (used for log-determinants of small Cholesky factors)
template <int D>
MFGM_DEV double log_diag_prod(const double (&L)[MFGM_NTRI(D)]) {
    double p = 1.0;
#pragma unroll
    for (int j = 0; j < D; ++j) p *= L[tix(j, j)];
    return log(p);
}

// Running log-determinant accumulator without a log per pivot: keeps a mantissa product and an
// integer exponent sum; log taken once at the end.
struct LogAcc {
    double mant;
    int expo;
    MFGM_DEV void init() { mant = 1.0; expo = 0; }
    MFGM_DEV void mul(double x) { mant *= x; }
    MFGM_DEV void renorm() {
        int e = __builtin_amdgcn_frexp_exp(mant);
        mant = __builtin_amdgcn_frexp_mant(mant);
        expo += e;
    }
    MFGM_DEV double value() const { return log(mant) + (double)expo * 0.69314718055994530942; }
};

}  // namespace mfgm
