// Tensor-product Gauss-Hermite kernels for drifts that couple the state dimensions or have no polynomial form, and for a full
// (non-diagonal) diffusion matrix: the reference's own formulation of the local CVI-DP / VDP quantities, natively.
//
//   linearisation   A_k = I + dt E_q[df/dx], b_k = dt (E_q f - E_q[df/dx] m)                  markovflow/sde/sde.py:92-131, 484-518 (10 points
//                                                                                             per dimension), sde_utils.py:119-179
//   Girsanov KL     1/2 sum_t { E_{q(x_t)} |x + dt f(x) - A_t x - b_t|^2_{Qp^-1} - d - logdet Qq_t + logdet Qp + tr(Qp^-1 Qq_t) }
//                   + KL[q(x0) || p(x0)]                                                      sde_utils.py:262-359 (20 points per dimension)
//   its gradient with respect to the expectation parameters (eta_lin, eta_diag, eta_sub)      sde_utils.py:473-547: the reference tapes
//                   through expectations_to_ssm_params and gpflow's mvnquad; here the same chain rule is written out -- the
//                   quadrature rule is differentiated AS A FORMULA (nodes X_i = m + sqrt(2) L xi_i, L = chol S: d X_i / d m = I,
//                   d X_i / d L_ab = sqrt(2) xi_ib e_a, then the Cholesky's reverse-mode rule), which is what a tape does and what
//                   matters for a ReLU drift, whose quadrature error is not small
//   and with respect to the drift parameters                                                   variational_cvi_sde.py:495-506
//   E_sde and its gradients (VDP)                                                             vi_sde.py:205-287
//
// State dimension d <= 3 (the tensor grid has 20^d points per time step; the reference's own runs are d <= 2).  Arrays are in the
// NATURAL layout ([B, T, d], [B, T, d, d]): these are small models, one thread works one time step.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mfgm.h"
#include "mfgm_math.h"

namespace mfgm {

constexpr int kQD = 3;                 // largest state dimension
constexpr int kQP = MFGM_QUAD_NTHETA;  // drift parameters

__device__ __constant__ const double kGH10x[5] = {0.3429013272237046, 1.0366108297895136, 1.7566836492998816, 2.5327316742327897,
                                                  3.4361591188377374};
__device__ __constant__ const double kGH10w[5] = {0.34464233493201907, 0.13548370298026777, 0.01911158050077031, 0.0007580709343122176,
                                                  4.310652630718299e-06};
__device__ __constant__ const double kGH20x[10] = {0.24534070830090124, 0.7374737285453944, 1.234076215395323, 1.7385377121165861,
                                                   2.2549740020892757, 2.7888060584281305, 3.3478545673832163, 3.944764040115625,
                                                   4.603682449550744, 5.387480890011233};
__device__ __constant__ const double kGH20w[10] = {0.2607930634495549, 0.16173933398399998, 0.0615063720639769, 0.013997837447101022,
                                                   0.00183010313108049, 0.00012882627996192928, 4.402121090230851e-06,
                                                   6.127490259982928e-08, 2.4820623623151755e-10, 1.2578006724379234e-13};
// node k (0 .. H-1, ascending) and weight (w / sqrt(pi)) of numpy.polynomial.hermite.hermgauss(H); the rule is symmetric
MFGM_DEV void gh_node(int H, int k, double& xi, double& w) {
    const int half = H / 2, kk = (k < half) ? (half - 1 - k) : (k - half);
    const double xa = (H == 10) ? kGH10x[kk] : kGH20x[kk];
    w = (H == 10) ? kGH10w[kk] : kGH20w[kk];
    xi = (k < half) ? -xa : xa;
}

// ---- drifts: f [d], J = df/dx [d, d] (row-major), and d f / d theta_p [np, d] at a point ------------------------------------------
//   kind 10  Van der Pol (sde.py:432-482), d = 2:  f = tau (a (x1 - x1^3 / 3 - x2), x1 / a),  theta = (a, tau)
//   kind 11  ReLU network 1 -> nh -> 1 applied to every state dimension (sde.py:359-429),  theta = (W1 [nh], b1 [nh], W2 [nh], b2)
//   kind 12  per-dimension cubic f_i = c1 x_i - c3 x_i^3 (Ornstein-Uhlenbeck: (-decay, 0); double well scale x (c - x^2): (scale c, scale);
//            sde.py:134-224) -- on this route when the diffusion matrix is not diagonal,  theta = (c1, c3)
//   kind 13 / 14 / 15  theta tanh x / sin(x - theta) / sqrt(theta |x|) per dimension (sde.py:227-356),  theta = (theta, -)
MFGM_DEV int quad_nparam(const mfgm_quad_drift& q) { return q.kind == 11 ? 3 * q.nh + 1 : 2; }

template <bool PGRAD>
MFGM_DEV void quad_drift(const mfgm_quad_drift& q, const double* x, double* f, double* J, double* fp /* [np][d] */) {
    const int d = q.d;
    if (q.kind == 10) {
        const double a = q.theta[0], tau = q.theta[1], x1 = x[0], x2 = x[1];
        const double g = x1 - x1 * x1 * x1 * (1.0 / 3.0) - x2;
        f[0] = tau * a * g;
        f[1] = tau * x1 / a;
        J[0] = tau * a * (1.0 - x1 * x1); J[1] = -tau * a;
        J[2] = tau / a; J[3] = 0.0;
        if (PGRAD) {
            fp[0] = tau * g; fp[1] = -tau * x1 / (a * a);            // d f / d a
            fp[2] = a * g; fp[3] = x1 / a;                           // d f / d tau
        }
    } else if (q.kind == 11) {
        const int nh = q.nh;
        const double *W1 = q.theta, *b1 = q.theta + nh, *W2 = q.theta + 2 * nh, b2 = q.theta[3 * nh];
        for (int e = 0; e < d * d; ++e) J[e] = 0.0;
        if (PGRAD)
            for (int e = 0; e < (3 * nh + 1) * d; ++e) fp[e] = 0.0;
        for (int i = 0; i < d; ++i) {
            double acc = b2, jac = 0.0;
            for (int k = 0; k < nh; ++k) {
                const double z = __builtin_fma(W1[k], x[i], b1[k]);
                const bool on = z > 0.0;                             // relu'(0) = 0, as TensorFlow's ReluGrad
                const double h = on ? z : 0.0;
                acc = __builtin_fma(W2[k], h, acc);
                if (on) jac = __builtin_fma(W2[k], W1[k], jac);
                if (PGRAD) {
                    fp[k * d + i] = on ? W2[k] * x[i] : 0.0;         // d / d W1_k
                    fp[(nh + k) * d + i] = on ? W2[k] : 0.0;         // d / d b1_k
                    fp[(2 * nh + k) * d + i] = h;                    // d / d W2_k
                }
            }
            if (PGRAD) fp[(3 * nh) * d + i] = 1.0;                   // d / d b2
            f[i] = acc;
            J[i * d + i] = jac;
        }
    } else if (q.kind >= 13) {
        // per-dimension non-polynomial drifts (sde.py:227-356): 13 theta tanh x, 14 sin(x - theta), 15 sqrt(theta |x|); one parameter
        const double th = q.theta[0];
        for (int e = 0; e < d * d; ++e) J[e] = 0.0;
        for (int i = 0; i < d; ++i) {
            const double xi = x[i];
            double fv, f1, ft;
            if (q.kind == 13) {
                const double t = tanh(xi);
                fv = th * t; f1 = th * (1.0 - t * t); ft = t;
            } else if (q.kind == 14) {
                fv = sin(xi - th); f1 = cos(xi - th); ft = -f1;
            } else {
                const double a = fabs(xi), r = sqrt(th * a);
                fv = r; f1 = (xi < 0.0 ? -0.5 : 0.5) * r / a; ft = 0.5 * r / th;
            }
            f[i] = fv;
            J[i * d + i] = f1;
            if (PGRAD) { fp[i] = ft; fp[d + i] = 0.0; }
        }
    } else {
        const double c1 = q.theta[0], c3 = q.theta[1];
        for (int e = 0; e < d * d; ++e) J[e] = 0.0;
        for (int i = 0; i < d; ++i) {
            const double xi = x[i], x2 = xi * xi;
            f[i] = xi * (c1 - c3 * x2);
            J[i * d + i] = c1 - 3.0 * c3 * x2;
            if (PGRAD) { fp[i] = xi; fp[d + i] = -xi * x2; }
        }
    }
}

// ---- small dense helpers, d <= 3, row-major full storage ---------------------------------------------------------------------------
MFGM_DEV bool q_chol(int d, const double* S, double* L) {          // lower Cholesky factor; false: not positive definite
    bool ok = true;
    for (int e = 0; e < d * d; ++e) L[e] = 0.0;
    for (int j = 0; j < d; ++j) {
        double s = S[j * d + j];
        for (int k = 0; k < j; ++k) s -= L[j * d + k] * L[j * d + k];
        if (!(s > 0.0)) { ok = false; s = 1.0; }
        const double r = sqrt(s);
        L[j * d + j] = r;
        for (int i = j + 1; i < d; ++i) {
            double t = S[i * d + j];
            for (int k = 0; k < j; ++k) t -= L[i * d + k] * L[j * d + k];
            L[i * d + j] = t / r;
        }
    }
    return ok;
}
MFGM_DEV void q_tri_inv(int d, const double* L, double* X) {       // X = L^{-1} (lower)
    for (int e = 0; e < d * d; ++e) X[e] = 0.0;
    for (int j = 0; j < d; ++j) {
        X[j * d + j] = 1.0 / L[j * d + j];
        for (int i = j + 1; i < d; ++i) {
            double t = 0.0;
            for (int k = j; k < i; ++k) t -= L[i * d + k] * X[k * d + j];
            X[i * d + j] = t / L[i * d + i];
        }
    }
}
MFGM_DEV void q_spd_inv(int d, const double* L, double* P) {       // (L L^T)^{-1} = X^T X
    double X[kQD * kQD];
    q_tri_inv(d, L, X);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double t = 0.0;
            for (int k = 0; k < d; ++k) t += X[k * d + i] * X[k * d + j];
            P[i * d + j] = t;
        }
}
MFGM_DEV void q_mm(int d, const double* A, const double* B, double* C, bool tA = false, bool tB = false) {
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double t = 0.0;
            for (int k = 0; k < d; ++k) t += (tA ? A[k * d + i] : A[i * d + k]) * (tB ? B[j * d + k] : B[k * d + j]);
            C[i * d + j] = t;
        }
}
MFGM_DEV double q_logdet_chol(int d, const double* L) {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += log(L[i * d + i]);
    return 2.0 * s;
}
MFGM_DEV double q_sym(const double* Wp, int i, int j) { return Wp[six(i, j)]; }      // packed lower triangle, symmetric access

// E_q h over the H^d tensor rule on N(m, L L^T): calls body(X [d], xi [d], w) for every node
template <class Body>
MFGM_DEV void quad_loop(int d, int H, const double* m, const double* L, Body body) {
    const double r2 = 1.4142135623730951;
    int idx[kQD] = {0, 0, 0};
    int total = 1;
    for (int i = 0; i < d; ++i) total *= H;
    for (int n = 0; n < total; ++n) {
        double xi[kQD], X[kQD], w = 1.0;
        for (int i = 0; i < d; ++i) {
            double wi;
            gh_node(H, idx[i], xi[i], wi);
            w *= wi;
        }
        for (int i = 0; i < d; ++i) {
            double t = m[i];
            for (int k = 0; k <= i; ++k) t = __builtin_fma(r2 * L[i * d + k], xi[k], t);
            X[i] = t;
        }
        body(X, xi, w);
        for (int i = d - 1; i >= 0; --i) {
            if (++idx[i] < H) break;
            idx[i] = 0;
        }
    }
}

// S-gradient of a function given through its gradient GL with respect to the lower Cholesky factor L of S (Murray 2016):
//   GS = 1/2 (P + P^T),  P = L^{-T} Phi(L^T GL) L^{-1},  Phi = lower triangle with the diagonal halved
MFGM_DEV void q_chol_backward(int d, const double* L, const double* GL, double* GS) {
    double M[kQD * kQD], X[kQD * kQD], T1[kQD * kQD], P[kQD * kQD];
    q_mm(d, L, GL, M, true, false);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) M[i * d + j] = (i > j) ? M[i * d + j] : ((i == j) ? 0.5 * M[i * d + j] : 0.0);
    q_tri_inv(d, L, X);
    q_mm(d, X, M, T1, true, false);          // L^{-T} Phi
    q_mm(d, T1, X, P);                       // ... L^{-1}
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) GS[i * d + j] = 0.5 * (P[i * d + j] + P[j * d + i]);
}

// ---- linearisation: one thread per (chain, transition); transition k is linearised on the marginal handed in for it -------------------
static __global__ void k_quad_linearize(mfgm_quad_drift q, int N, const double* __restrict__ mean, const double* __restrict__ cov,
                                        double* __restrict__ A, double* __restrict__ b, int* info) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int d = q.d;
    double m[kQD], L[kQD * kQD], Ef[kQD], EJ[kQD * kQD];
    for (int i = 0; i < d; ++i) m[i] = mean[(size_t)n * d + i];
    if (!q_chol(d, cov + (size_t)n * d * d, L)) atomicMax(info, 1);
    for (int i = 0; i < d; ++i) Ef[i] = 0.0;
    for (int e = 0; e < d * d; ++e) EJ[e] = 0.0;
    quad_loop(d, 10, m, L, [&](const double* X, const double*, double w) {
        double f[kQD], J[kQD * kQD];
        quad_drift<false>(q, X, f, J, nullptr);
        for (int i = 0; i < d; ++i) Ef[i] = __builtin_fma(w, f[i], Ef[i]);
        for (int e = 0; e < d * d; ++e) EJ[e] = __builtin_fma(w, J[e], EJ[e]);
    });
    const bool clip = q.clip_lo < q.clip_hi;
    for (int i = 0; i < d; ++i) {
        double t = Ef[i];
        for (int j = 0; j < d; ++j) t -= EJ[i * d + j] * m[j];
        double bv = q.dt * t;
        if (clip) bv = fmin(fmax(bv, q.clip_lo), q.clip_hi);
        b[(size_t)n * d + i] = bv;
        for (int j = 0; j < d; ++j) {
            double av = (i == j ? 1.0 : 0.0) + q.dt * EJ[i * d + j];
            if (clip) av = fmin(fmax(av, q.clip_lo), q.clip_hi);
            A[((size_t)n * d + i) * d + j] = av;
        }
    }
}

// ---- one transition of the Girsanov KL and its gradient pieces ----------------------------------------------------------------------
// (m, S) = marginal at t, C = Cov(x_{t+1}, x_t), (mn, Sn) = marginal at t + 1.  Returns the transition's KL; pieces (all [d] / [d, d]
// row-major, symmetric ones symmetrised):  Gm, GS, GC = d KL_t / d (m, S, C),  Gn, GSn = d KL_t / d (mn, Sn);  gth [np] += d KL_t / d theta.
template <bool GRAD>
MFGM_DEV double quad_transition(const mfgm_quad_drift& q, const double* m, const double* S, const double* C, const double* mn,
                                const double* Sn, double* Gm, double* GS, double* GC, double* Gn, double* GSn, double* gth, bool& ok) {
    const int d = q.d, np = quad_nparam(q);
    double L[kQD * kQD], Sinv[kQD * kQD], A[kQD * kQD], bq[kQD], Qq[kQD * kQD], Lq[kQD * kQD];
    ok = q_chol(d, S, L) && ok;
    q_spd_inv(d, L, Sinv);
    q_mm(d, C, Sinv, A);                                     // A = C S^{-1}
    for (int i = 0; i < d; ++i) {
        double t = mn[i];
        for (int j = 0; j < d; ++j) t -= A[i * d + j] * m[j];
        bq[i] = t;
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double t = Sn[i * d + j];
            for (int k = 0; k < d; ++k) t -= A[i * d + k] * C[j * d + k];
            Qq[i * d + j] = t;
        }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < i; ++j) Qq[i * d + j] = Qq[j * d + i] = 0.5 * (Qq[i * d + j] + Qq[j * d + i]);
    ok = q_chol(d, Qq, Lq) && ok;
    // g = E_q r^T W r,  r = x + dt f(x) - A x - b, and the gradients of the rule with respect to (b, A, X-nodes, theta)
    double g = 0.0, gb[kQD], gA[kQD * kQD], gm[kQD], gL[kQD * kQD];
    for (int i = 0; i < d; ++i) { gb[i] = 0.0; gm[i] = 0.0; }
    for (int e = 0; e < d * d; ++e) { gA[e] = 0.0; gL[e] = 0.0; }
    quad_loop(d, 20, m, L, [&](const double* X, const double* xi, double w) {
        double f[kQD], J[kQD * kQD], fp[kQP * kQD], r[kQD], Wr[kQD];
        quad_drift<GRAD>(q, X, f, J, fp);
        for (int i = 0; i < d; ++i) {
            double t = X[i] + q.dt * f[i] - bq[i];
            for (int j = 0; j < d; ++j) t -= A[i * d + j] * X[j];
            r[i] = t;
        }
        double h = 0.0;
        for (int i = 0; i < d; ++i) {
            double t = 0.0;
            for (int j = 0; j < d; ++j) t += q_sym(q.W, i, j) * r[j];
            Wr[i] = t;
            h += t * r[i];
        }
        g = __builtin_fma(w, h, g);
        if (GRAD) {
            for (int i = 0; i < d; ++i) {
                gb[i] -= 2.0 * w * Wr[i];
                for (int j = 0; j < d; ++j) gA[i * d + j] -= 2.0 * w * Wr[i] * X[j];
            }
            // d h / d X = 2 (I + dt J - A)^T W r
            for (int a = 0; a < d; ++a) {
                double t = 0.0;
                for (int i = 0; i < d; ++i) t += ((i == a ? 1.0 : 0.0) + q.dt * J[i * d + a] - A[i * d + a]) * Wr[i];
                const double gx = 2.0 * w * t;
                gm[a] += gx;
                for (int c = 0; c <= a; ++c) gL[a * d + c] += 1.4142135623730951 * gx * xi[c];
            }
            for (int p = 0; p < np; ++p) {
                double t = 0.0;
                for (int i = 0; i < d; ++i) t += Wr[i] * fp[p * d + i];
                gth[p] += w * q.dt * t;                      // 1/2 * 2 w r^T W dt df/dtheta
            }
        }
    });
    // KL_t = 1/2 { g - d - logdet Qq + logdet Qp + tr(W Qq) }
    double trWQ = 0.0;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) trWQ += q_sym(q.W, i, j) * Qq[i * d + j];
    const double val = 0.5 * (g - (double)d - q_logdet_chol(d, Lq) + q.logdetQp + trWQ);
    if (GRAD) {
        double P[kQD * kQD], GQ[kQD * kQD], GAt[kQD * kQD], T1[kQD * kQD], T2[kQD * kQD], GSl[kQD * kQD];
        q_spd_inv(d, Lq, P);
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) GQ[i * d + j] = q_sym(q.W, i, j) - P[i * d + j];          // d h / d Qq
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) GAt[i * d + j] = gA[i * d + j] - gb[i] * m[j];             // total d g / d A (b = mn - A m)
        // GC = 1/2 [ GAt S^{-1} - 2 GQ A ]
        q_mm(d, GAt, Sinv, T1);
        q_mm(d, GQ, A, T2);
        for (int e = 0; e < d * d; ++e) GC[e] = 0.5 * T1[e] - T2[e];
        // GS = 1/2 [ chol-backward(gL) + sym(-A^T GAt S^{-1}) + A^T GQ A ]
        q_chol_backward(d, L, gL, GSl);
        double U[kQD * kQD], V[kQD * kQD];
        q_mm(d, A, T1, U, true, false);                      // A^T GAt S^{-1}
        q_mm(d, A, T2, V, true, false);                      // A^T GQ A
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j)
                GS[i * d + j] = 0.5 * (GSl[i * d + j] - 0.5 * (U[i * d + j] + U[j * d + i]) + 0.5 * (V[i * d + j] + V[j * d + i]));
        for (int i = 0; i < d; ++i) {
            double t = gm[i];
            for (int k = 0; k < d; ++k) t -= A[k * d + i] * gb[k];
            Gm[i] = 0.5 * t;
            Gn[i] = 0.5 * gb[i];
        }
        for (int e = 0; e < d * d; ++e) GSn[e] = 0.5 * GQ[e];
    }
    return val;
}

// pieces per transition: [Gm d | Gn d | GS d^2 | GC d^2 | GSn d^2]
MFGM_DEV int quad_piece_size(int d) { return 2 * d + 3 * d * d; }

template <bool GRAD>
static __global__ void k_quad_kl_transitions(mfgm_quad_drift q, int B, int T, const double* __restrict__ mu, const double* __restrict__ Sig,
                                             const double* __restrict__ Sub, double* __restrict__ klt, double* __restrict__ pieces,
                                             double* __restrict__ gtheta, int* info) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B * (T - 1)) return;
    const int d = q.d, np = quad_nparam(q), b = n / (T - 1), t = n - b * (T - 1);
    const size_t node = (size_t)b * T + t;
    double Gm[kQD], GS[kQD * kQD], GC[kQD * kQD], Gn[kQD], GSn[kQD * kQD], gth[kQP];
    for (int p = 0; p < np; ++p) gth[p] = 0.0;
    bool ok = true;
    const double v = quad_transition<GRAD>(q, mu + node * d, Sig + node * d * d, Sub + ((size_t)b * (T - 1) + t) * d * d, mu + (node + 1) * d,
                                           Sig + (node + 1) * d * d, Gm, GS, GC, Gn, GSn, gth, ok);
    if (!ok) atomicMax(info, 1);
    klt[n] = v;
    if (GRAD) {
        double* o = pieces + (size_t)n * quad_piece_size(d);
        for (int i = 0; i < d; ++i) { o[i] = Gm[i]; o[d + i] = Gn[i]; }
        for (int e = 0; e < d * d; ++e) { o[2 * d + e] = GS[e]; o[2 * d + d * d + e] = GC[e]; o[2 * d + 2 * d * d + e] = GSn[e]; }
        for (int p = 0; p < np; ++p) gtheta[(size_t)n * np + p] = gth[p];
    }
}

// d KL / d (eta_lin, eta_diag, eta_sub) of node (b, t) from the pieces of the transitions t (own) and t - 1 (entering) and, at t = 0,
// KL[q(x0) || p(x0)]:  with eta_lin = m, eta_diag = S + m m^T, eta_sub_t = C_t + m_{t+1} m_t^T,
//   d/d eta_lin_t = dF/dm_t - 2 (dF/dS_t) m_t - (dF/dC_t)^T m_{t+1} - (dF/dC_{t-1}) m_{t-1},  d/d eta_diag = dF/dS,  d/d eta_sub = dF/dC
static __global__ void k_quad_kl_assemble(mfgm_quad_drift q, int B, int T, const double* __restrict__ mu, const double* __restrict__ Sig,
                                          const double* __restrict__ pieces, double* __restrict__ kl0, double* __restrict__ g1,
                                          double* __restrict__ gd, double* __restrict__ gs, int* info) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B * T) return;
    const int d = q.d, b = n / T, t = n - b * T, ps = quad_piece_size(d);
    double dm[kQD], dS[kQD * kQD];
    for (int i = 0; i < d; ++i) dm[i] = 0.0;
    for (int e = 0; e < d * d; ++e) dS[e] = 0.0;
    const double* m = mu + (size_t)n * d;
    if (t == 0) {
        double L[kQD * kQD], Si[kQD * kQD], diff[kQD];
        if (!q_chol(d, Sig + (size_t)n * d * d, L)) atomicMax(info, 1);
        q_spd_inv(d, L, Si);
        double tr = 0.0, mh = 0.0;
        for (int i = 0; i < d; ++i) diff[i] = m[i] - q.mu0[i];
        for (int i = 0; i < d; ++i) {
            double tt = 0.0;
            for (int j = 0; j < d; ++j) {
                tt += q_sym(q.P0inv, i, j) * diff[j];
                tr += q_sym(q.P0inv, i, j) * Sig[((size_t)n * d + i) * d + j];
                dS[i * d + j] = 0.5 * (q_sym(q.P0inv, i, j) - Si[i * d + j]);
            }
            dm[i] = tt;
            mh += tt * diff[i];
        }
        kl0[b] = 0.5 * (tr + mh - (double)d + q.logdetP0 - q_logdet_chol(d, L));
    }
    const double* own = (t < T - 1) ? pieces + ((size_t)b * (T - 1) + t) * ps : nullptr;
    const double* ent = (t > 0) ? pieces + ((size_t)b * (T - 1) + t - 1) * ps : nullptr;
    if (own) {
        for (int i = 0; i < d; ++i) dm[i] += own[i];
        for (int e = 0; e < d * d; ++e) dS[e] += own[2 * d + e];
    }
    if (ent) {
        for (int i = 0; i < d; ++i) dm[i] += ent[d + i];
        for (int e = 0; e < d * d; ++e) dS[e] += ent[2 * d + 2 * d * d + e];
    }
    for (int i = 0; i < d; ++i) {
        double tt = dm[i];
        for (int j = 0; j < d; ++j) tt -= 2.0 * dS[i * d + j] * m[j];
        if (own) {
            const double* GC = own + 2 * d + d * d;
            const double* mn = mu + (size_t)(n + 1) * d;
            for (int j = 0; j < d; ++j) tt -= GC[j * d + i] * mn[j];
        }
        if (ent) {
            const double* GC = ent + 2 * d + d * d;
            const double* mp = mu + (size_t)(n - 1) * d;
            for (int j = 0; j < d; ++j) tt -= GC[i * d + j] * mp[j];
        }
        g1[(size_t)n * d + i] = tt;
    }
    for (int e = 0; e < d * d; ++e) gd[(size_t)n * d * d + e] = dS[e];
    if (own) {
        const double* GC = own + 2 * d + d * d;
        for (int e = 0; e < d * d; ++e) gs[((size_t)b * (T - 1) + t) * d * d + e] = GC[e];
    }
}

// KL[q(x0) || p(x0)] alone (value-only calls)
static __global__ void k_quad_kl0(mfgm_quad_drift q, int B, int T, const double* __restrict__ mu, const double* __restrict__ Sig,
                                  double* __restrict__ kl0, int* info) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int d = q.d;
    const size_t n = (size_t)b * T;
    double L[kQD * kQD];
    if (!q_chol(d, Sig + n * d * d, L)) atomicMax(info, 1);
    double tr = 0.0, mh = 0.0;
    for (int i = 0; i < d; ++i) {
        double tt = 0.0;
        for (int j = 0; j < d; ++j) {
            tt += q_sym(q.P0inv, i, j) * (mu[n * d + j] - q.mu0[j]);
            tr += q_sym(q.P0inv, i, j) * Sig[(n * d + i) * d + j];
        }
        mh += tt * (mu[n * d + i] - q.mu0[i]);
    }
    kl0[b] = 0.5 * (tr + mh - (double)d + q.logdetP0 - q_logdet_chol(d, L));
}

// out[b, c] = add[b] (c == 0) + sum_t x[b, t, c]: the per-chain KL and parameter-gradient sums (one block per chain; fixed order)
static __global__ __launch_bounds__(256) void k_quad_sum(int n_t, int nc, const double* __restrict__ x, const double* __restrict__ add,
                                                         double* __restrict__ out) {
    __shared__ double sh[256];
    const int b = blockIdx.x;
    for (int c = 0; c < nc; ++c) {
        double acc = 0.0;
        for (int t = threadIdx.x; t < n_t; t += 256) acc += x[((size_t)b * n_t + t) * nc + c];
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[(size_t)b * nc + c] = sh[0] + ((add && c == 0) ? add[b] : 0.0);
        __syncthreads();
    }
}

// ---- VDP: E_sde of one node and its gradients (vi_sde.py:205-287) ----------------------------------------------------------------------
//   E_sde_t = 1/2 E_{N(m, S)} |f(x) - (-A x + b)|^2_{q^-1}   (the reference's q(x) drift is  -A_t x + b_t, vi_sde.py:60-75),
//   Wq = q^{-1} (packed, q.W holds (dt q)^{-1}: Wq = dt q.W).  Outputs dE/dm [d], dE/dS [d, d] (symmetric), dE/dA, dE/db, dE/dtheta.
static __global__ void k_quad_esde(mfgm_quad_drift q, int N, const double* __restrict__ mean, const double* __restrict__ cov,
                                   const double* __restrict__ Aq, const double* __restrict__ bq, double* __restrict__ E,
                                   double* __restrict__ dEdm, double* __restrict__ dEdS, double* __restrict__ dEdA, double* __restrict__ dEdb,
                                   double* __restrict__ gtheta, int* info) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int d = q.d, np = quad_nparam(q);
    double m[kQD], L[kQD * kQD], A[kQD * kQD], b[kQD];
    for (int i = 0; i < d; ++i) { m[i] = mean[(size_t)n * d + i]; b[i] = bq[(size_t)n * d + i]; }
    for (int e = 0; e < d * d; ++e) A[e] = Aq[(size_t)n * d * d + e];
    if (!q_chol(d, cov + (size_t)n * d * d, L)) atomicMax(info, 1);
    double g = 0.0, gb[kQD], gA[kQD * kQD], gm[kQD], gL[kQD * kQD], gth[kQP];
    for (int i = 0; i < d; ++i) { gb[i] = 0.0; gm[i] = 0.0; }
    for (int e = 0; e < d * d; ++e) { gA[e] = 0.0; gL[e] = 0.0; }
    for (int p = 0; p < np; ++p) gth[p] = 0.0;
    quad_loop(d, 20, m, L, [&](const double* X, const double* xi, double w) {
        double f[kQD], J[kQD * kQD], fp[kQP * kQD], r[kQD], Wr[kQD];
        quad_drift<true>(q, X, f, J, fp);
        for (int i = 0; i < d; ++i) {
            double t = f[i] - b[i];
            for (int j = 0; j < d; ++j) t += A[i * d + j] * X[j];
            r[i] = t;
        }
        double h = 0.0;
        for (int i = 0; i < d; ++i) {
            double t = 0.0;
            for (int j = 0; j < d; ++j) t += q.dt * q_sym(q.W, i, j) * r[j];
            Wr[i] = t;
            h += t * r[i];
        }
        g = __builtin_fma(w, h, g);
        for (int i = 0; i < d; ++i) {
            gb[i] -= 2.0 * w * Wr[i];
            for (int j = 0; j < d; ++j) gA[i * d + j] += 2.0 * w * Wr[i] * X[j];
        }
        for (int a = 0; a < d; ++a) {
            double t = 0.0;
            for (int i = 0; i < d; ++i) t += (J[i * d + a] + A[i * d + a]) * Wr[i];
            const double gx = 2.0 * w * t;
            gm[a] += gx;
            for (int c = 0; c <= a; ++c) gL[a * d + c] += 1.4142135623730951 * gx * xi[c];
        }
        for (int p = 0; p < np; ++p) {
            double t = 0.0;
            for (int i = 0; i < d; ++i) t += Wr[i] * fp[p * d + i];
            gth[p] += w * t;
        }
    });
    E[n] = 0.5 * g;
    double GS[kQD * kQD];
    q_chol_backward(d, L, gL, GS);
    for (int i = 0; i < d; ++i) {
        if (dEdm) dEdm[(size_t)n * d + i] = 0.5 * gm[i];
        if (dEdb) dEdb[(size_t)n * d + i] = 0.5 * gb[i];
    }
    for (int e = 0; e < d * d; ++e) {
        if (dEdS) dEdS[(size_t)n * d * d + e] = 0.5 * GS[e];
        if (dEdA) dEdA[(size_t)n * d * d + e] = 0.5 * gA[e];
    }
    if (gtheta)
        for (int p = 0; p < np; ++p) gtheta[(size_t)n * np + p] = gth[p];
}

// ---- VDP: the Lagrange sweep with jump conditions (vi_sde.py:289-347), one thread per chain ---------------------------------------------
// psi [B, N, d, d], lam [B, N, d] (N = transitions): row N - 1 is (1e-10 I, 0), then for t = N - 1 .. 1
//   psi_{t-1} = psi_t - dt (psi_t A_t + psi_t A_t - dEdS_t) - dobsS_t,   lam_{t-1} = lam_t - dt (A_t lam_t - dEdm_t) - dobsm_t
// exactly as the reference's loop writes it (`psi @ A + psi @ A`).  clip > 0: stabilize_system -- NaN -> 1e-8 and clipping to [-clip, clip]
// of the four gradient arrays on load (vi_sde.py:312-323).  These are the small models of the quadrature route (T ~ 500).
MFGM_DEV double quad_stab(double x, double c) {
    if (c <= 0.0) return x;
    if (x != x) x = 1e-8;
    return fmin(fmax(x, -c), c);
}
static __global__ void k_quad_vdp_lagrange(int B, int N, int d, double dt, double clip, const double* __restrict__ A,
                                           const double* __restrict__ dEdm, const double* __restrict__ dEdS,
                                           const double* __restrict__ dobsm, const double* __restrict__ dobsS, double* __restrict__ psi,
                                           double* __restrict__ lam) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double P[kQD * kQD], l[kQD];
    for (int i = 0; i < d; ++i) {
        l[i] = 0.0;
        for (int j = 0; j < d; ++j) P[i * d + j] = (i == j) ? 1e-10 : 0.0;
    }
    for (int t = N - 1; t >= 0; --t) {
        const size_t n = (size_t)b * N + t;
        for (int i = 0; i < d; ++i) lam[n * d + i] = l[i];
        for (int e = 0; e < d * d; ++e) psi[n * d * d + e] = P[e];
        if (t == 0) break;
        const double* At = A + n * d * d;
        const size_t no = (size_t)b * (N + 1) + t;
        double PA[kQD * kQD], Al[kQD];
        q_mm(d, P, At, PA);
        for (int i = 0; i < d; ++i) {
            double tt = 0.0;
            for (int j = 0; j < d; ++j) tt += At[i * d + j] * l[j];
            Al[i] = tt;
        }
        for (int e = 0; e < d * d; ++e)
            P[e] = P[e] - dt * (PA[e] + PA[e] - quad_stab(dEdS[n * d * d + e], clip)) - quad_stab(dobsS[no * d * d + e], clip);
        for (int i = 0; i < d; ++i) l[i] = l[i] - dt * (Al[i] - quad_stab(dEdm[n * d + i], clip)) - quad_stab(dobsm[no * d + i], clip);
    }
}

}  // namespace mfgm
