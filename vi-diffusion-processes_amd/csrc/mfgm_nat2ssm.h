// naturals_to_ssm_params, the per-step part (ssm_gaussian_transformations.py:459-511), as ONE pass over the selected-inverse output
// in the packed layout (state_dim <= 8):
//   A_k   = Sigma_{k+1,k} Sigma_kk^{-1}                                   (:459-462)
//   C_k   = P_kk + A_k^T P_{k+1,k}, symmetrised  (= Q_{k-1}^{-1}, C_0 = P_0^{-1}; the block diagonal of :473-484)
//   chol_k = chol(C_k^{-1})                       (:493-495: chol, cholesky_solve(I), chol)
//   b_k   = mu_{k+1} - A_k mu_k                   (what :497-511 solves for, read off the marginal means)
// with P = -2 theta_diag / -theta_sub the precision blocks.  The reference runs these as separate batched factorisations and
// solves over [B, T, d, d] tensors; here every block is read once and the SSM parameters come out packed (A FULL at node k, offsets
// VEC: mu_0 at node 0 and b_{k-1} at node k, Cholesky factors TRI: chol P_0 at node 0 and chol Q_{k-1} at node k).
#pragma once
#include "mfgm_local.h"

namespace mfgm {

template <int D>
static __global__ __launch_bounds__(64) void k_naturals_to_ssm(LevelDesc lv, const double* __restrict__ Sigg,
                                                              const double* __restrict__ Subg, const double* __restrict__ mug,
                                                              const double* __restrict__ tdg, const double* __restrict__ tsg,
                                                              double* __restrict__ Ag, double* __restrict__ offg,
                                                              double* __restrict__ cholg, int* __restrict__ info) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int p = lane % P;
    const int len = min(R, n - p * R);
    int bad = 0;
    double mc[D];
    ld_node<D>(mug, R, 0, me, mc);
    if (p == 0) st_node<D>(offg, R, 0, me, mc);       // offsets at node 0: the initial mean
    for (int s = 0; s < len; ++s) {
        const bool has_next = (p * R + s + 1 < n);
        double Cp[ET];
        ld_node<ET>(tdg, R, s, me, Cp);
#pragma unroll
        for (int e = 0; e < ET; ++e) Cp[e] *= -2.0;                 // P_kk
        if (has_next) {
            double Sg[ET], A[EF], Ps[EF], mn[D], invd[D];
            ld_node<ET>(Sigg, R, s, me, Sg);
            ld_node<EF>(Subg, R, s, me, A);
            ld_node<EF>(tsg, R, s, me, Ps);
            ld_next<D>(mug, R, s, len, lane, me, mn);
            chol_inplace<D>(Sg, invd, bad);
            trsm_right_lower_t<D>(Sg, invd, A);                     // Sigma_{k+1,k} L^{-T}
            trsm_right_lower<D>(Sg, invd, A);                       // ... L^{-1}  = Sigma_{k+1,k} Sigma_kk^{-1}
            st_node<EF>(Ag, R, s, me, A);
            // C += sym(A^T P_{k+1,k}),  P_{k+1,k} = -theta_sub
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double t = 0.0;
#pragma unroll
                    for (int m = 0; m < D; ++m) t += A[m * D + i] * Ps[m * D + j] + A[m * D + j] * Ps[m * D + i];
                    Cp[tix(i, j)] -= 0.5 * t;
                }
            // b_k = mu_{k+1} - A_k mu_k, stored with node k+1
            double bk[D];
            gemv<D>(A, mc, bk);
#pragma unroll
            for (int i = 0; i < D; ++i) bk[i] = mn[i] - bk[i];
            if (s + 1 < len) st_node<D>(offg, R, s + 1, me, bk);
            else st_node<D>(offg, R, 0, LaneRef::of(lane + 1), bk);
#pragma unroll
            for (int i = 0; i < D; ++i) mc[i] = mn[i];
        } else {
            st_node_zero<EF>(Ag, R, s, me);
        }
        // chol(C^{-1}): C = Lc Lc^T, C^{-1} = X^T X with X = Lc^{-1}, then its Cholesky factor
        double invd[D], X[ET], Q[ET];
        chol_inplace<D>(Cp, invd, bad);
        tri_inverse<D>(Cp, invd, X);
        tri_t_tri<D>(X, Q);
        chol_inplace<D>(Q, invd, bad);
        st_node<ET>(cholg, R, s, me, Q);
    }
    if (bad) atomicMax(info, 1);
}

// ---- block-tri-diagonal matrix times vector, natural layout, any d <= 32 (block_tri_diag.py:175-199 -> product_band_mat) ---------------
// diag [B, T, d, d] (symmetric: lower triangles read), sub [B, T-1, d, d] or null, x / out [B, T, d].
// symmetric: out = M x with M = M^T;  otherwise M is lower block-bidiagonal and transpose selects M^T x.
static __global__ __launch_bounds__(256) void k_btd_matvec(int B, int T, int d, const double* __restrict__ diag,
                                                          const double* __restrict__ sub, const double* __restrict__ x,
                                                          double* __restrict__ out, int symmetric, int transpose) {
    const size_t total = (size_t)B * T * d;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t node = e / d;
        const int i = (int)(e - node * d);
        const size_t b = node / T, t = node - b * T;
        const double* Dn = diag + node * (size_t)(d * d);
        const double* xn = x + node * (size_t)d;
        double acc = 0.0;
        for (int j = 0; j < d; ++j) {
            double v;
            if (symmetric) v = (j <= i) ? Dn[i * d + j] : Dn[j * d + i];
            else v = transpose ? Dn[j * d + i] : Dn[i * d + j];
            acc += v * xn[j];
        }
        if (sub) {
            if ((symmetric || !transpose) && t > 0) {          // lower part: S_{t-1} x_{t-1}
                const double* S = sub + (b * (T - 1) + (t - 1)) * (size_t)(d * d);
                const double* xp = xn - d;
                for (int j = 0; j < d; ++j) acc += S[i * d + j] * xp[j];
            }
            if ((symmetric || transpose) && t + 1 < (size_t)T) {   // upper part: S_t^T x_{t+1}
                const double* S = sub + (b * (T - 1) + t) * (size_t)(d * d);
                const double* xq = xn + d;
                for (int j = 0; j < d; ++j) acc += S[j * d + i] * xq[j];
            }
        }
        out[e] = acc;
    }
}

}  // namespace mfgm
