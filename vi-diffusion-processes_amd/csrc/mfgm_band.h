// Band of  X = Sigma dP Sigma  for the covariance Sigma of a Gauss-Markov chain given by its band and a symmetric block-tri-diagonal dP:
// the covariance half of the exact derivative of the marginals with respect to the natural parameters, d Sigma = -Sigma dP Sigma,
// which the reference obtains from a GradientTape through banded_matrices' registered gradients of cholesky_band /
// inverse_from_cholesky_band (ssm_natgrad.py:154-201 on naturals_to_ssm_params; variational_cvi_sde.py:508-518).
//
// Every block of Sigma factors through its band:  Sigma_{t+1,s} = A_t Sigma_{t,s} (s <= t),  A_t = C_t Sigma_t^-1,  C_t = Sigma_{t+1,t};
// Sigma_{t,s} = J_t Sigma_{t+1,s} (s > t),  J_t = C_t^T Sigma_{t+1}^-1.  With loc_t = Sigma_t dP_tt Sigma_t,
//   L_t = A_{t-1} L_{t-1} A_{t-1}^T + QL_t,   QL_t = loc_t + sym(Sigma_t dP_{t,t-1} C_{t-1}^T)        (pairs a, b <= t)
//   R_t = J_t R_{t+1} J_t^T + QR_t,           QR_t = loc_t + sym(C_t^T dP_{t+1,t} Sigma_t)            (pairs a, b >= t)
//   X_tt = L_t + R_t - loc_t,     X_{t+1,t} = A_t L_t + R_{t+1} J_t^T + C_t dP_{t+1,t}^T C_t + Sigma_{t+1} dP_{t+1,t} Sigma_t
// (sym(M) = M + M^T).  k_band_prepare forms the maps and offsets of the two recurrences node by node (one Cholesky of Sigma_t per node
// serves A_t and J_{t-1}), the recurrences run on k_congruence_scan (mfgm_vdp.h) -- the descending one on arrays written in reversed
// node order -- and k_band_finish assembles X.  All arrays packed (lane-per-segment plans, d <= 8).
#pragma once
#include "mfgm_sweeps.h"

namespace mfgm {

struct BandArgs {
    const double* Sig;    // SYM  Sigma_tt
    const double* Sub;    // FULL Sigma_{t+1,t} at node t
    const double* dPd;    // SYM  dP_tt (lower triangle)
    const double* dPs;    // FULL dP_{t+1,t} at node t
    double* PhiL; double* QL;    // FULL / SYM, natural node order
    double* PhiR; double* QR;    // FULL / SYM, REVERSED node order (node t of a chain at position n-1-t)
    double* loc;                 // SYM
    const double* Lr; const double* Rr;   // scan results (SYM): L natural order, R reversed order
    double* Xd; double* Xs;      // outputs: SYM X_tt, FULL X_{t+1,t}
};

// (lane, step) of node t of chain b
struct NodeAt { LaneRef w; int s; };
MFGM_DEV NodeAt node_at(const LevelDesc& lv, int b, int t) {
    const int p = t / lv.R;
    return NodeAt{LaneRef::of(b * lv.P + p), t - p * lv.R};
}

// out(sym) = S M S for symmetric S, M (packed)
template <int D>
MFGM_DEV void sym_congruence_sym(const double (&S)[MFGM_NTRI(D)], const double (&M)[MFGM_NTRI(D)], double (&out)[MFGM_NTRI(D)]) {
    double tmp[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(M[six(i, k)], S[six(k, j)], t);
            tmp[i * D + j] = t;
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(S[six(i, k)], tmp[k * D + j], t);
            out[tix(i, j)] = t;
        }
}

template <int D>
__global__ __launch_bounds__(64) void k_band_prepare(LevelDesc lv, BandArgs a) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    for (int s = 0; s < len; ++s) {
        const int t = p * R + s;
        const NodeAt rev = node_at(lv, b, n - 1 - t);
        double S[ET], loc[ET], Lc[ET], invd[D];
        {
            double M[ET];
            ld_node<ET>(a.Sig, R, s, me, S);
            ld_node<ET>(a.dPd, R, s, me, M);
            sym_congruence_sym<D>(S, M, loc);
        }
        st_node<ET>(a.loc, R, s, me, loc);
#pragma unroll
        for (int e = 0; e < ET; ++e) Lc[e] = S[e];
        int bad = 0;
        chol_inplace<D>(Lc, invd, bad);
        // edge (t, t+1): QR_t and A_t
        {
            double QR[ET];
#pragma unroll
            for (int e = 0; e < ET; ++e) QR[e] = loc[e];
            if (t + 1 < n) {
                double C[EF], tmp[EF];
                {
                    double dS[EF];
                    ld_node<EF>(a.dPs, R, s, me, dS);
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            double v = 0.0;
#pragma unroll
                            for (int k = 0; k < D; ++k) v = __builtin_fma(dS[i * D + k], S[six(k, j)], v);
                            tmp[i * D + j] = v;                       // dP_{t+1,t} Sigma_t
                        }
                }
                ld_node<EF>(a.Sub, R, s, me, C);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        double mij = 0.0, mji = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            mij = __builtin_fma(C[k * D + i], tmp[k * D + j], mij);      // (C^T tmp)_ij
                            mji = __builtin_fma(C[k * D + j], tmp[k * D + i], mji);
                        }
                        QR[tix(i, j)] += mij + mji;
                    }
                trsm_right_lower_t<D>(Lc, invd, C);
                trsm_right_lower<D>(Lc, invd, C);                     // A_t = C_t Sigma_t^-1
                const NodeAt nx = node_at(lv, b, t + 1);
                st_node<EF>(a.PhiL, R, nx.s, nx.w, C);
            } else {
                st_node_zero<EF>(a.PhiR, R, rev.s, rev.w);            // J_{n-1} = 0: nothing to the right of the last node
            }
            st_node<ET>(a.QR, R, rev.s, rev.w, QR);
        }
        // edge (t-1, t): QL_t and J_{t-1}
        {
            double QL[ET];
#pragma unroll
            for (int e = 0; e < ET; ++e) QL[e] = loc[e];
            if (t > 0) {
                const NodeAt pv = node_at(lv, b, t - 1);
                double Ct[EF], m1[EF];
                {
                    double Cp[EF], dSp[EF], tmp[EF];
                    ld_node<EF>(a.Sub, R, pv.s, pv.w, Cp);
                    ld_node<EF>(a.dPs, R, pv.s, pv.w, dSp);
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            double v = 0.0;
#pragma unroll
                            for (int k = 0; k < D; ++k) v = __builtin_fma(dSp[i * D + k], Cp[j * D + k], v);
                            tmp[i * D + j] = v;                       // dP_{t,t-1} C_{t-1}^T
                            Ct[i * D + j] = Cp[j * D + i];
                        }
                    gemm_sym_full<D>(S, tmp, m1);                     // Sigma_t dP_{t,t-1} C_{t-1}^T
                }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) QL[tix(i, j)] += m1[i * D + j] + m1[j * D + i];
                trsm_right_lower_t<D>(Lc, invd, Ct);
                trsm_right_lower<D>(Lc, invd, Ct);                    // J_{t-1} = C_{t-1}^T Sigma_t^-1
                const NodeAt rp = node_at(lv, b, n - t);              // reversed position of node t-1
                st_node<EF>(a.PhiR, R, rp.s, rp.w, Ct);
            } else {
                st_node_zero<EF>(a.PhiL, R, s, me);                   // A_{-1} = 0: nothing to the left of the first node
            }
            st_node<ET>(a.QL, R, s, me, QL);
        }
    }
}

template <int D>
__global__ __launch_bounds__(64) void k_band_finish(LevelDesc lv, BandArgs a) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    for (int s = 0; s < len; ++s) {
        const int t = p * R + s;
        const NodeAt rev = node_at(lv, b, n - 1 - t);
        double Lt[ET];
        ld_node<ET>(a.Lr, R, s, me, Lt);
        {
            double Rt[ET], loc[ET], Xd[ET];
            ld_node<ET>(a.Rr, R, rev.s, rev.w, Rt);
            ld_node<ET>(a.loc, R, s, me, loc);
#pragma unroll
            for (int e = 0; e < ET; ++e) Xd[e] = Lt[e] + Rt[e] - loc[e];
            st_node<ET>(a.Xd, R, s, me, Xd);
        }
        if (t + 1 < n) {
            const NodeAt nx = node_at(lv, b, t + 1), rn = node_at(lv, b, n - 2 - t);
            double Xs[EF];
            {
                double A[EF];
                ld_node<EF>(a.PhiL, R, nx.s, nx.w, A);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) v = __builtin_fma(A[i * D + k], Lt[six(k, j)], v);
                        Xs[i * D + j] = v;                            // A_t L_t
                    }
            }
            {
                double J[EF], Rn[ET];
                ld_node<EF>(a.PhiR, R, rev.s, rev.w, J);
                ld_node<ET>(a.Rr, R, rn.s, rn.w, Rn);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double v = Xs[i * D + j];
#pragma unroll
                        for (int k = 0; k < D; ++k) v = __builtin_fma(Rn[six(i, k)], J[j * D + k], v);
                        Xs[i * D + j] = v;                            // + R_{t+1} J_t^T
                    }
            }
            double dS[EF];
            ld_node<EF>(a.dPs, R, s, me, dS);
            {
                double C[EF], tmp[EF];
                ld_node<EF>(a.Sub, R, s, me, C);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) v = __builtin_fma(dS[k * D + i], C[k * D + j], v);
                        tmp[i * D + j] = v;                           // dP_{t+1,t}^T C_t
                    }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double v = Xs[i * D + j];
#pragma unroll
                        for (int k = 0; k < D; ++k) v = __builtin_fma(C[i * D + k], tmp[k * D + j], v);
                        Xs[i * D + j] = v;                            // + C_t dP_{t+1,t}^T C_t
                    }
            }
            {
                double S[ET], Sn[ET], tmp[EF];
                ld_node<ET>(a.Sig, R, s, me, S);
                ld_node<ET>(a.Sig, R, nx.s, nx.w, Sn);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) v = __builtin_fma(dS[i * D + k], S[six(k, j)], v);
                        tmp[i * D + j] = v;                           // dP_{t+1,t} Sigma_t
                    }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double v = Xs[i * D + j];
#pragma unroll
                        for (int k = 0; k < D; ++k) v = __builtin_fma(Sn[six(i, k)], tmp[k * D + j], v);
                        Xs[i * D + j] = v;                            // + Sigma_{t+1} dP_{t+1,t} Sigma_t
                    }
            }
            st_node<EF>(a.Xs, R, s, me, Xs);
        }
    }
}

}  // namespace mfgm
