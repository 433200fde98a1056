// Local kernels of the sparse-precision Kalman filter with Gaussian sites and a TIME-INVARIANT emission matrix H [O, D]
// (kalman_filter.py:86-107, 184-271, 417-500; the path of KalmanFilterWithSites / CVIGaussianProcess.elbo / predict_f at the data):
//   k_kf_assemble : posterior precision blocks  D_t = K^{-1}_tt + H^T R_t^{-1} H  and the right-hand side, straight into the packed
//                   layout, with the per-chain sums of the observation terms of the log-likelihood
//   k_kf_project  : f-marginals  H mu_t,  diag(H Sigma_t H^T)  from the packed posterior marginals
// The reference does each of these with einsums over materialised [B, T, o, d] emission tensors plus a re-layout per access.
#pragma once
#include "mfgm_sweeps.h"

namespace mfgm {

struct KfArgs {
    double H[4 * 8];        // emission matrix, row-major [O][D]
    const double* nat1;     // sites, natural layout [Bs, T, O] (Bs = 1: shared by all chains, or B)
    const double* nat2;     // [Bs, T, O, O]
    const double* Hmu;      // H mu_prior [B, T, O] (natural) or null for a zero-mean prior
    int site_batch;         // Bs
    int mode;               // 0: rhs = H^T R^{-1} (y - H mu_p) and the log-likelihood sums; 1: rhs = K^{-1} mu_p + H^T nat1
};

// R^{-1} = -2 nat2, y = R nat1 (site mean), for O = 1, 2
template <int O>
MFGM_DEV void kf_site(const double* __restrict__ n1, const double* __restrict__ n2, double (&Ri)[O * O], double (&y)[O], double& logdet) {
    if constexpr (O == 1) {
        Ri[0] = -2.0 * n2[0];
        y[0] = n1[0] / Ri[0];
        logdet = log(Ri[0]);
    } else {
        static_assert(O == 2, "sites of output dimension 1 or 2");
        Ri[0] = -2.0 * n2[0]; Ri[1] = -2.0 * n2[1]; Ri[2] = -2.0 * n2[2]; Ri[3] = -2.0 * n2[3];
        const double det = Ri[0] * Ri[3] - Ri[1] * Ri[2];
        y[0] = (Ri[3] * n1[0] - Ri[1] * n1[1]) / det;
        y[1] = (Ri[0] * n1[1] - Ri[2] * n1[0]) / det;
        logdet = log(det);
    }
}

template <int D, int O>
static __global__ __launch_bounds__(64) void k_kf_assemble(LevelDesc lv, int T, KfArgs k, const double* __restrict__ Pd,
                                                          const double* __restrict__ plin, double* __restrict__ Dg,
                                                          double* __restrict__ rg, double* __restrict__ part) {
    constexpr int ET = MFGM_NTRI(D);
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    double t1 = 0.0, ld = 0.0;
    for (int s = 0; s < len; ++s) {
        const size_t t = (size_t)p * R + s;
        const size_t sr = (size_t)(k.site_batch == 1 ? 0 : b) * T + t;
        double Ri[O * O], y[O], logdet;
        kf_site<O>(k.nat1 + sr * O, k.nat2 + sr * O * O, Ri, y, logdet);
        double dd[ET], r[D], hm[O];
        ld_node<ET>(Pd, R, s, me, dd);
#pragma unroll
        for (int a = 0; a < O; ++a) hm[a] = k.Hmu ? k.Hmu[((size_t)b * T + t) * O + a] : 0.0;
        // D += H^T R^{-1} H
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int a = 0; a < O; ++a)
#pragma unroll
                    for (int c = 0; c < O; ++c) acc += k.H[a * D + i] * Ri[a * O + c] * k.H[c * D + j];
                dd[tix(i, j)] += acc;
            }
        // v = R^{-1} (y - H mu_p) = nat1 - R^{-1} H mu_p   (mode 0);   nat1 (mode 1)
        double v[O];
#pragma unroll
        for (int a = 0; a < O; ++a) {
            double acc = k.nat1[sr * O + a];
            if (k.mode == 0) {
#pragma unroll
                for (int c = 0; c < O; ++c) acc -= Ri[a * O + c] * hm[c];
            }
            v[a] = acc;
        }
        if (k.mode == 1 && plin) ld_node<D>(plin, R, s, me, r);
        else {
#pragma unroll
            for (int i = 0; i < D; ++i) r[i] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int a = 0; a < O; ++a) r[i] += k.H[a * D + i] * v[a];
        st_node<ET>(Dg, R, s, me, dd);
        st_node<D>(rg, R, s, me, r);
        // (y - H mu_p)^T R^{-1} (y - H mu_p)
#pragma unroll
        for (int a = 0; a < O; ++a) t1 += (y[a] - hm[a]) * v[a];
        ld += logdet;
    }
    if (part) {
        part[lane] = t1;
        part[lv.Lpad + lane] = ld;
    }
}

template <int D, int O>
static __global__ __launch_bounds__(64) void k_kf_project(LevelDesc lv, int T, KfArgs k, const double* __restrict__ mu,
                                                         const double* __restrict__ Sig, double* __restrict__ Fmu,
                                                         double* __restrict__ Fvar) {
    constexpr int ET = MFGM_NTRI(D);
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    for (int s = 0; s < len; ++s) {
        const size_t t = (size_t)p * R + s;
        double m[D], S[ET];
        ld_node<D>(mu, R, s, me, m);
        ld_node<ET>(Sig, R, s, me, S);
#pragma unroll
        for (int a = 0; a < O; ++a) {
            double fm = 0.0, fv = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                fm += k.H[a * D + i] * m[i];
#pragma unroll
                for (int j = 0; j < D; ++j) fv += k.H[a * D + i] * k.H[a * D + j] * S[six(i, j)];
            }
            Fmu[((size_t)b * T + t) * O + a] = fm;
            Fvar[((size_t)b * T + t) * O + a] = fv;
        }
    }
}

}  // namespace mfgm
