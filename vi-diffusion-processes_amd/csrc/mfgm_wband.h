// Band of  X = Sigma dP Sigma  for block sizes up to 32 on the MFMA (natural-layout arrays, one wavefront per node / per segment): the
// covariance half of the exact derivative of the marginals with respect to the natural parameters, d Sigma = -Sigma dP Sigma, which the
// reference obtains from a GradientTape through banded_matrices' registered gradients of cholesky_band / inverse_from_cholesky_band
// (ssm_natgrad.py:154-201 on naturals_to_ssm_params).  Same mathematics as mfgm_band.h (d <= 8, packed arrays):
//   A_t = C_t Sigma_t^-1,  J_t = C_t^T Sigma_{t+1}^-1   (C_t = Sigma_{t+1,t}),   loc_t = Sigma_t dP_tt Sigma_t,
//   L_t = A_{t-1} L_{t-1} A_{t-1}^T + QL_t,   QL_t = loc_t + sym(Sigma_t dP_{t,t-1} C_{t-1}^T)        (ascending)
//   R_t = J_t R_{t+1} J_t^T + QR_t,           QR_t = loc_t + sym(C_t^T dP_{t+1,t} Sigma_t)            (descending)
//   X_tt = L_t + R_t - loc_t,     X_{t+1,t} = A_t L_t + R_{t+1} J_t^T + C_t dP_{t+1,t}^T C_t + Sigma_{t+1} dP_{t+1,t} Sigma_t
// written with Gram products gram(X, Y) = X^T Y of accumulator-layout tiles only (mfgm_mfma.h): blocks are loaded from memory in the
// orientation a product needs, the maps of the two recurrences are STORED TRANSPOSED (PhiL_t = A_{t-1}^T, PhiR_t = J_t^T), so that
//   Phi X Phi^T = gram(PhiT, gram(X, PhiT))      (X symmetric)
// and the composition of two steps of a recurrence is gram(PhiT_2, Phi_1) / gram(PhiT_2, gram(Q_1, PhiT_2)) + Q_2: no transposes through
// LDS anywhere.  Sigma_t^-1 comes from the 4 x 4-pivot block sweeps of mfgm_mfma_inv.h (sweep_inv), once per node.
// The two recurrences share their launches.  A recurrence over T nodes runs as three passes over segments of ~sqrt(T / 2.5) nodes
// (maps of the segments, one sequential pass over the segment maps of a chain, final sweep), like mfgm_congruence_scan does for the
// packed plans.
#pragma once
#include "mfgm_mfma_inv.h"

namespace mfgm {

struct WBandArgs {
    int B, T, d;
    const double* Sig;    // [B, T, d, d]     Sigma_tt
    const double* Sub;    // [B, T-1, d, d]   Sigma_{t+1,t}
    const double* dPd;    // [B, T, d, d]     dP_tt (symmetric, full)
    const double* dPs;    // [B, T-1, d, d]   dP_{t+1,t}
    double* PhiL; double* QL;     // [B, T, d, d]:  PhiL_t = A_{t-1}^T (zero at t = 0)
    double* PhiR; double* QR;     //                PhiR_t = J_t^T     (zero at t = T-1)
    double* loc;
    double* Lr; double* Rr;       // the two recurrences
    double* Xd; double* Xs;       // outputs [B, T, d, d], [B, T-1, d, d]
    int* info;
};

struct WScanArgs {
    int B, T, d, R, P;            // segments of R positions, P per chain
    int reverse;                  // position j of the recurrence is node T-1-j
    const double* PhiT; const double* Q;
    double* X;
    double* segT; double* segQ; double* segX;     // [B, P, d, d] each
};

MFGM_DEV const double* wb_blk(const double* base, int b, int n, int t, int EF) { return base + ((size_t)b * n + t) * EF; }
MFGM_DEV double* wb_blk(double* base, int b, int n, int t, int EF) { return base + ((size_t)b * n + t) * EF; }

// per node: Sigma_t^-1, the maps towards both neighbours, loc, QL, QR
template <int NT>
static __global__ __launch_bounds__(64) void kwb_prepare(WBandArgs a) {
    __shared__ double lds[16];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d, T = a.T;
    const int b = blockIdx.x / T, t = blockIdx.x - b * T;
    int bad = 0;
    const Mat<NT> S = ld_mat<NT, false, false>(wb_blk(a.Sig, b, T, t, EF), d, L, 1.0);
    const Mat<NT> Dm = ld_mat<NT, false, false>(wb_blk(a.dPd, b, T, t, EF), d, L, 1.0);
    Mat<NT> K = ld_mat<NT, false, true>(wb_blk(a.Sig, b, T, t, EF), d, L, 1.0);
    LogAcc la;
    la.init();
    sweep_inv<NT>(K, L, la, bad, lds);
    const Mat<NT> loc = gram<NT>(S, gram<NT>(Dm, S));
    st_mat<NT, false>(wb_blk(a.loc, b, T, t, EF), d, L, loc);
    if (t > 0) {
        const double* cp = wb_blk(a.Sub, b, T - 1, t - 1, EF);
        const double* ep = wb_blk(a.dPs, b, T - 1, t - 1, EF);
        const Mat<NT> C = ld_mat<NT, false, false>(cp, d, L, 1.0), CT = ld_mat<NT, true, false>(cp, d, L, 1.0);
        const Mat<NT> E = ld_mat<NT, false, false>(ep, d, L, 1.0), ET = ld_mat<NT, true, false>(ep, d, L, 1.0);
        // m = Sigma_t dP_{t,t-1} C_{t-1}^T = gram(S, gram(ET, CT)),  m^T = gram(CT, gram(E, S))
        Mat<NT> QL = gram<NT>(S, gram<NT>(ET, CT), loc);
        QL = gram<NT>(CT, gram<NT>(E, S), QL);
        st_mat<NT, false>(wb_blk(a.QL, b, T, t, EF), d, L, QL);
        st_mat<NT, false>(wb_blk(a.PhiR, b, T, t - 1, EF), d, L, gram<NT>(K, C));        // J_{t-1}^T = Sigma_t^-1 C_{t-1}
    } else {
        st_mat<NT, false>(wb_blk(a.QL, b, T, t, EF), d, L, loc);
        st_mat<NT, false>(wb_blk(a.PhiL, b, T, 0, EF), d, L, mat_zero<NT>());
    }
    if (t + 1 < T) {
        const double* cp = wb_blk(a.Sub, b, T - 1, t, EF);
        const double* ep = wb_blk(a.dPs, b, T - 1, t, EF);
        const Mat<NT> C = ld_mat<NT, false, false>(cp, d, L, 1.0), CT = ld_mat<NT, true, false>(cp, d, L, 1.0);
        const Mat<NT> E = ld_mat<NT, false, false>(ep, d, L, 1.0), ET = ld_mat<NT, true, false>(ep, d, L, 1.0);
        // m = C_t^T dP_{t+1,t} Sigma_t = gram(C, gram(ET, S)),  m^T = gram(S, gram(E, C))
        Mat<NT> QR = gram<NT>(C, gram<NT>(ET, S), loc);
        QR = gram<NT>(S, gram<NT>(E, C), QR);
        st_mat<NT, false>(wb_blk(a.QR, b, T, t, EF), d, L, QR);
        st_mat<NT, false>(wb_blk(a.PhiL, b, T, t + 1, EF), d, L, gram<NT>(K, CT));       // A_t^T = Sigma_t^-1 C_t^T
    } else {
        st_mat<NT, false>(wb_blk(a.QR, b, T, t, EF), d, L, loc);
        st_mat<NT, false>(wb_blk(a.PhiR, b, T, T - 1, EF), d, L, mat_zero<NT>());
    }
    if (bad && L.lane == 0) flag_not_pd(a.info, 0, blockIdx.x);
}

// the ascending and the descending recurrence are independent: both ride in one launch (blockIdx.y picks the recurrence)
struct WScanPair { WScanArgs s[2]; };

MFGM_DEV int wb_node(const WScanArgs& a, int j) { return a.reverse ? a.T - 1 - j : j; }

// pass 1: the composite map (Phi, Q) of every segment; Phi is stored transposed
template <int NT>
static __global__ __launch_bounds__(64) void kwb_scan_maps(WScanPair w) {
    const WScanArgs& a = w.s[blockIdx.y];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int b = blockIdx.x / a.P, p = blockIdx.x - b * a.P;
    const int j0 = p * a.R, j1 = min(a.T, j0 + a.R);
    Mat<NT> Phi = ld_mat<NT, true, false>(wb_blk(a.PhiT, b, a.T, wb_node(a, j0), EF), d, L, 1.0);
    Mat<NT> Q = ld_mat<NT, false, false>(wb_blk(a.Q, b, a.T, wb_node(a, j0), EF), d, L, 1.0);
    Mat<NT> PTn = mat_zero<NT>(), Qn = mat_zero<NT>();
    if (j0 + 1 < j1) {
        PTn = ld_mat<NT, false, false>(wb_blk(a.PhiT, b, a.T, wb_node(a, j0 + 1), EF), d, L, 1.0);
        Qn = ld_mat<NT, false, false>(wb_blk(a.Q, b, a.T, wb_node(a, j0 + 1), EF), d, L, 1.0);
    }
    for (int j = j0 + 1; j < j1; ++j) {
        const Mat<NT> PT = PTn, Qj = Qn;
        if (j + 1 < j1) {
            PTn = ld_mat<NT, false, false>(wb_blk(a.PhiT, b, a.T, wb_node(a, j + 1), EF), d, L, 1.0);
            Qn = ld_mat<NT, false, false>(wb_blk(a.Q, b, a.T, wb_node(a, j + 1), EF), d, L, 1.0);
        }
        Phi = gram<NT>(PT, Phi);
        Q = gram<NT>(PT, gram<NT>(Q, PT), Qj);
    }
    st_mat<NT, true>(wb_blk(a.segT, b, a.P, p, EF), d, L, Phi);
    st_mat<NT, false>(wb_blk(a.segQ, b, a.P, p, EF), d, L, Q);
}

// pass 2: X at the last position of every segment, one wavefront per chain
template <int NT>
static __global__ __launch_bounds__(64) void kwb_scan_tops(WScanPair w) {
    const WScanArgs& a = w.s[blockIdx.y];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d, b = blockIdx.x;
    Mat<NT> X = mat_zero<NT>();
    Mat<NT> PTn = ld_mat<NT, false, false>(wb_blk(a.segT, b, a.P, 0, EF), d, L, 1.0);
    Mat<NT> Qn = ld_mat<NT, false, false>(wb_blk(a.segQ, b, a.P, 0, EF), d, L, 1.0);
    for (int p = 0; p < a.P; ++p) {
        const Mat<NT> PT = PTn, Qp = Qn;
        if (p + 1 < a.P) {
            PTn = ld_mat<NT, false, false>(wb_blk(a.segT, b, a.P, p + 1, EF), d, L, 1.0);
            Qn = ld_mat<NT, false, false>(wb_blk(a.segQ, b, a.P, p + 1, EF), d, L, 1.0);
        }
        X = gram<NT>(PT, gram<NT>(X, PT), Qp);
        st_mat<NT, false>(wb_blk(a.segX, b, a.P, p, EF), d, L, X);
    }
}

// pass 3: the recurrence inside every segment from the value at its left end
template <int NT>
static __global__ __launch_bounds__(64) void kwb_scan_sweep(WScanPair w) {
    const WScanArgs& a = w.s[blockIdx.y];
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d;
    const int b = blockIdx.x / a.P, p = blockIdx.x - b * a.P;
    const int j0 = p * a.R, j1 = min(a.T, j0 + a.R);
    Mat<NT> X = (p > 0) ? ld_mat<NT, false, false>(wb_blk(a.segX, b, a.P, p - 1, EF), d, L, 1.0) : mat_zero<NT>();
    Mat<NT> PTn = ld_mat<NT, false, false>(wb_blk(a.PhiT, b, a.T, wb_node(a, j0), EF), d, L, 1.0);
    Mat<NT> Qn = ld_mat<NT, false, false>(wb_blk(a.Q, b, a.T, wb_node(a, j0), EF), d, L, 1.0);
    for (int j = j0; j < j1; ++j) {
        const Mat<NT> PT = PTn, Qj = Qn;
        if (j + 1 < j1) {
            PTn = ld_mat<NT, false, false>(wb_blk(a.PhiT, b, a.T, wb_node(a, j + 1), EF), d, L, 1.0);
            Qn = ld_mat<NT, false, false>(wb_blk(a.Q, b, a.T, wb_node(a, j + 1), EF), d, L, 1.0);
        }
        X = gram<NT>(PT, gram<NT>(X, PT), Qj);
        st_mat<NT, false>(wb_blk(a.X, b, a.T, wb_node(a, j), EF), d, L, X);
    }
}

// per node: X_tt and X_{t+1,t}
template <int NT>
static __global__ __launch_bounds__(64) void kwb_finish(WBandArgs a) {
    const LaneId L{(int)threadIdx.x, (int)threadIdx.x >> 4, (int)threadIdx.x & 15};
    const int d = a.d, EF = d * d, T = a.T;
    const int b = blockIdx.x / T, t = blockIdx.x - b * T;
    const Mat<NT> Lm = ld_mat<NT, false, false>(wb_blk(a.Lr, b, T, t, EF), d, L, 1.0);
    {
        const Mat<NT> Rm = ld_mat<NT, false, false>(wb_blk(a.Rr, b, T, t, EF), d, L, 1.0);
        const Mat<NT> loc = ld_mat<NT, false, false>(wb_blk(a.loc, b, T, t, EF), d, L, 1.0);
        st_mat<NT, false>(wb_blk(a.Xd, b, T, t, EF), d, L, mat_sub<NT>(mat_add<NT>(Lm, Rm), loc));
    }
    if (t + 1 < T) {
        const double* cp = wb_blk(a.Sub, b, T - 1, t, EF);
        const double* ep = wb_blk(a.dPs, b, T - 1, t, EF);
        const Mat<NT> C = ld_mat<NT, false, false>(cp, d, L, 1.0), CT = ld_mat<NT, true, false>(cp, d, L, 1.0);
        const Mat<NT> E = ld_mat<NT, false, false>(ep, d, L, 1.0), ET = ld_mat<NT, true, false>(ep, d, L, 1.0);
        const Mat<NT> S0 = ld_mat<NT, false, false>(wb_blk(a.Sig, b, T, t, EF), d, L, 1.0);
        const Mat<NT> S1 = ld_mat<NT, false, false>(wb_blk(a.Sig, b, T, t + 1, EF), d, L, 1.0);
        const Mat<NT> AT = ld_mat<NT, false, false>(wb_blk(a.PhiL, b, T, t + 1, EF), d, L, 1.0);
        const Mat<NT> JT = ld_mat<NT, false, false>(wb_blk(a.PhiR, b, T, t, EF), d, L, 1.0);
        const Mat<NT> R1 = ld_mat<NT, false, false>(wb_blk(a.Rr, b, T, t + 1, EF), d, L, 1.0);
        Mat<NT> Xs = gram<NT>(AT, Lm);                               // A_t L_t
        Xs = gram<NT>(R1, JT, Xs);                                   // R_{t+1} J_t^T   (R symmetric)
        Xs = gram<NT>(CT, gram<NT>(E, C), Xs);                       // C dP_{t+1,t}^T C
        Xs = gram<NT>(S1, gram<NT>(ET, S0), Xs);                     // Sigma_{t+1} dP_{t+1,t} Sigma_t
        st_mat<NT, false>(wb_blk(a.Xs, b, T - 1, t, EF), d, L, Xs);
    }
}

}  // namespace mfgm
