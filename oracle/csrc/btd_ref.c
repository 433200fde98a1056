/*
 * CPU restatement ("port") of the reference's block-tri-diagonal Gauss-Markov path in plain C.
 *
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/__init__.py): used by tests/ to cross-check the NumPy
 * oracle and by bench.py's `cpu_baseline` leg.  Never linked into or called from the product library.
 *
 * Natural row-major layout, one chain at a time: diag [T,d,d], sub [T-1,d,d], vec [T,d].
 * Algorithms follow the reference's sequential banded route:
 *   ref_btd_cholesky        SymmetricBlockTriDiagonal.cholesky   (markovflow/block_tri_diag.py:428-440 -> cholesky_band)
 *   ref_btd_solve           LowerTriangularBlockTriDiagonal.solve (block_tri_diag.py:339-351 -> solve_triang_mat)
 *   ref_btd_inverse_blocks  inverse_from_cholesky_band            (block_tri_diag.py:318-337; ssm_gaussian_transformations.py:443-458)
 *   ref_ssm_to_naturals     ssm_to_naturals                       (ssm_gaussian_transformations.py:182-253)
 *   ref_kl_terms            StateSpaceModel.kl_divergence         (state_space_model.py:557-593)
 *   ref_cvi_step            one CVISitesSSM iteration: update_data_sites, update_girsanov_sites, classic_elbo
 *                           (variational_cvi_sde.py:161-192, 279-352; docs/diffusion_processes/cvi_dp_trainer.py:72-75)
 *   ref_vdp_forward / ref_vdp_step   VariationalMarkovGP: forward_pass, update_lagrange + update_param, elbo
 *                           (markovflow/models/vi_sde.py:171-204, 289-414, 436-455; the loop body of
 *                           docs/diffusion_processes/vi_markov_gp_trainer.py:50-75), closed-form cubic-drift moments
 * banded-matrices (the native dependency that holds these loops in the reference) is not vendored in
 * /root/reference; its published algorithms are restated here in block form.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(i, j, d) ((i) * (d) + (j))

/* in-place lower Cholesky of a d x d SPD matrix (lower triangle read); returns 1 if not PD */
static int chol_d(double* a, int d) {
    for (int j = 0; j < d; ++j) {
        double s = a[IDX(j, j, d)];
        for (int k = 0; k < j; ++k) s -= a[IDX(j, k, d)] * a[IDX(j, k, d)];
        if (!(s > 0.0)) return 1;
        double l = sqrt(s);
        a[IDX(j, j, d)] = l;
        for (int i = j + 1; i < d; ++i) {
            double t = a[IDX(i, j, d)];
            for (int k = 0; k < j; ++k) t -= a[IDX(i, k, d)] * a[IDX(j, k, d)];
            a[IDX(i, j, d)] = t / l;
        }
        for (int i = 0; i < j; ++i) a[IDX(i, j, d)] = 0.0;
    }
    return 0;
}
/* x := L^{-1} x (n right-hand sides stored as columns of the d x n row-major matrix x) */
static void trsm_l(const double* L, double* x, int d, int n) {
    for (int c = 0; c < n; ++c)
        for (int i = 0; i < d; ++i) {
            double t = x[i * n + c];
            for (int k = 0; k < i; ++k) t -= L[IDX(i, k, d)] * x[k * n + c];
            x[i * n + c] = t / L[IDX(i, i, d)];
        }
}
/* x := L^{-T} x */
static void trsm_lt(const double* L, double* x, int d, int n) {
    for (int c = 0; c < n; ++c)
        for (int i = d - 1; i >= 0; --i) {
            double t = x[i * n + c];
            for (int k = i + 1; k < d; ++k) t -= L[IDX(k, i, d)] * x[k * n + c];
            x[i * n + c] = t / L[IDX(i, i, d)];
        }
}
/* C (+)= op(A) op(B): plain triple loops; ta/tb transpose flags; beta in {0,1}; alpha scalar */
static void gemm_d(double alpha, const double* A, int ta, const double* B, int tb, double beta, double* C, int d) {
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double t = 0.0;
            for (int k = 0; k < d; ++k) t += (ta ? A[IDX(k, i, d)] : A[IDX(i, k, d)]) * (tb ? B[IDX(j, k, d)] : B[IDX(k, j, d)]);
            C[IDX(i, j, d)] = alpha * t + (beta != 0.0 ? C[IDX(i, j, d)] : 0.0);
        }
}

int ref_btd_cholesky(const double* diag, const double* sub, double* Ld, double* Ls, int T, int d) {
    const int dd = d * d;
    double* carry = (double*)calloc(dd, sizeof(double));
    double* tmp = (double*)malloc(dd * sizeof(double));
    int bad = 0;
    for (int k = 0; k < T && !bad; ++k) {
        double* L = Ld + (size_t)k * dd;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) L[IDX(i, j, d)] = diag[(size_t)k * dd + IDX(i, j, d)] - carry[IDX(i, j, d)];
        bad = chol_d(L, d);
        if (k < T - 1) {
            /* L_{k+1,k} = S_k L^{-T}: solve L X^T = S^T */
            const double* S = sub + (size_t)k * dd;
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) tmp[IDX(i, j, d)] = S[IDX(j, i, d)];
            trsm_l(L, tmp, d, d);
            double* G = Ls + (size_t)k * dd;
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) G[IDX(i, j, d)] = tmp[IDX(j, i, d)];
            gemm_d(1.0, G, 0, G, 1, 0.0, carry, d);
        }
    }
    free(carry);
    free(tmp);
    return bad;
}

void ref_btd_solve(const double* Ld, const double* Ls, const double* rhs, double* out, int T, int d, int transpose) {
    const int dd = d * d;
    if (!transpose) {
        for (int k = 0; k < T; ++k) {
            double* y = out + (size_t)k * d;
            for (int i = 0; i < d; ++i) {
                double t = rhs[(size_t)k * d + i];
                if (k > 0)
                    for (int j = 0; j < d; ++j) t -= Ls[(size_t)(k - 1) * dd + IDX(i, j, d)] * out[(size_t)(k - 1) * d + j];
                y[i] = t;
            }
            trsm_l(Ld + (size_t)k * dd, y, d, 1);
        }
    } else {
        for (int k = T - 1; k >= 0; --k) {
            double* y = out + (size_t)k * d;
            for (int i = 0; i < d; ++i) {
                double t = rhs[(size_t)k * d + i];
                if (k < T - 1)
                    for (int j = 0; j < d; ++j) t -= Ls[(size_t)k * dd + IDX(j, i, d)] * out[(size_t)(k + 1) * d + j];
                y[i] = t;
            }
            trsm_lt(Ld + (size_t)k * dd, y, d, 1);
        }
    }
}

double ref_btd_logdet(const double* Ld, int T, int d) {
    double s = 0.0;
    for (int k = 0; k < T; ++k)
        for (int i = 0; i < d; ++i) s += log(Ld[(size_t)k * d * d + IDX(i, i, d)]);
    return s;
}

void ref_btd_inverse_blocks(const double* Ld, const double* Ls, double* Sd, double* Ss, int T, int d) {
    const int dd = d * d;
    double* Linv = (double*)malloc(dd * sizeof(double));
    double* H = (double*)malloc(dd * sizeof(double));
    for (int k = T - 1; k >= 0; --k) {
        const double* L = Ld + (size_t)k * dd;
        memset(Linv, 0, dd * sizeof(double));
        for (int i = 0; i < d; ++i) Linv[IDX(i, i, d)] = 1.0;
        trsm_l(L, Linv, d, d);
        double* S = Sd + (size_t)k * dd;
        gemm_d(1.0, Linv, 1, Linv, 0, 0.0, S, d);
        if (k < T - 1) {
            gemm_d(1.0, Ls + (size_t)k * dd, 0, Linv, 0, 0.0, H, d);            /* H = L_{k+1,k} L_kk^{-1} */
            double* Sb = Ss + (size_t)k * dd;
            gemm_d(-1.0, Sd + (size_t)(k + 1) * dd, 0, H, 0, 0.0, Sb, d);       /* S_{k+1,k} = -S_{k+1,k+1} H */
            gemm_d(-1.0, Sb, 1, H, 0, 1.0, S, d);                               /* S_kk -= S_{k+1,k}^T H */
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < i; ++j) {
                double v = 0.5 * (S[IDX(i, j, d)] + S[IDX(j, i, d)]);
                S[IDX(i, j, d)] = S[IDX(j, i, d)] = v;
            }
    }
    free(Linv);
    free(H);
}

/* A [T-1,d,d], off [T,d] (mu0 then b), chol [T,d,d] (chol P0 then chol Q) -> naturals */
void ref_ssm_to_naturals(const double* A, const double* off, const double* chol, double* lin, double* diag, double* sub,
                         int T, int d) {
    const int dd = d * d;
    double* X = (double*)malloc(dd * sizeof(double));
    double* M = (double*)malloc(dd * sizeof(double));
    double* z = (double*)malloc((size_t)T * d * sizeof(double));
    for (int k = 0; k < T; ++k) {
        const double* C = chol + (size_t)k * dd;
        memset(X, 0, dd * sizeof(double));
        for (int i = 0; i < d; ++i) X[IDX(i, i, d)] = 1.0;
        trsm_l(C, X, d, d);
        gemm_d(-0.5, X, 1, X, 0, 0.0, diag + (size_t)k * dd, d);   /* -1/2 Q^{-1} */
        double* zk = z + (size_t)k * d;
        memcpy(zk, off + (size_t)k * d, d * sizeof(double));
        trsm_l(C, zk, d, 1);
        trsm_lt(C, zk, d, 1);
        memcpy(lin + (size_t)k * d, zk, d * sizeof(double));
        if (k > 0) {
            const double* Ak = A + (size_t)(k - 1) * dd;
            gemm_d(1.0, X, 0, Ak, 0, 0.0, M, d);                                  /* chol^{-1} A */
            gemm_d(1.0, X, 1, M, 0, 0.0, sub + (size_t)(k - 1) * dd, d);          /* Q^{-1} A */
            gemm_d(-0.5, M, 1, M, 0, 1.0, diag + (size_t)(k - 1) * dd, d);        /* -1/2 A^T Q^{-1} A */
            for (int i = 0; i < d; ++i) {
                double t = 0.0;
                for (int j = 0; j < d; ++j) t += Ak[IDX(j, i, d)] * zk[j];
                lin[(size_t)(k - 1) * d + i] -= t;
            }
        }
    }
    free(X);
    free(M);
    free(z);
}

/* trace and Mahalanobis terms of KL(q||p): q blocks (Sig, Sub, mu), p precision (Pd, Ps) and means mup */
void ref_kl_terms(const double* Sig, const double* Sub, const double* mu, const double* Pd, const double* Ps, const double* mup,
                  int T, int d, double* trace, double* maha) {
    const int dd = d * d;
    double tr = 0.0, mh = 0.0;
    for (int k = 0; k < T; ++k) {
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double pij = Pd[(size_t)k * dd + IDX(i, j, d)];
                tr += pij * Sig[(size_t)k * dd + IDX(i, j, d)];
                mh += pij * (mup[(size_t)k * d + i] - mu[(size_t)k * d + i]) * (mup[(size_t)k * d + j] - mu[(size_t)k * d + j]);
            }
        if (k < T - 1)
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double pij = Ps[(size_t)k * dd + IDX(i, j, d)];
                    tr += 2.0 * pij * Sub[(size_t)k * dd + IDX(i, j, d)];
                    mh += 2.0 * pij * (mup[(size_t)(k + 1) * d + i] - mu[(size_t)(k + 1) * d + i]) * (mup[(size_t)k * d + j] - mu[(size_t)k * d + j]);
                }
    }
    *trace = tr;
    *maha = mh;
}

/*
 * One CVISitesSSM iteration for B chains (OpenMP over chains), Gaussian likelihood with precision Rinv [d,d]:
 *   update_data_sites(lr_d); update_girsanov_sites(lr_g); classic_elbo()
 * State (updated in place): girsanov sites g1 [B,T,d], g2d [B,T,d,d], g2s [B,T-1,d,d]; data sites d1 [B,n,d], d2 [B,n,d,d].
 * Prior naturals p1, pd, ps, prior means pmu, prior sum-log-chol pslc[B]; observations y [B,n,d] at grid indices idx[n].
 * Returns the sum over chains of the ELBO; elbo_out[B] per chain.  work: caller-provided scratch of
 * ref_cvi_step_work_doubles(T,d) doubles PER CHAIN.
 */
size_t ref_cvi_step_work_doubles(int T, int d) { return (size_t)T * (7 * d * d + 4 * d) + 16 * d * d; }

double ref_cvi_step(int B, int T, int d, int n, const int* idx, const double* y, const double* Rinv, double logdetR,
                    const double* p1, const double* pd, const double* ps, const double* pmu, const double* pslc,
                    double* g1, double* g2d, double* g2s, double* d1, double* d2, double lr_d, double lr_g,
                    double* work, double* elbo_out) {
    const int dd = d * d;
    const size_t W = ref_cvi_step_work_doubles(T, d);
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int b = 0; b < B; ++b) {
        double* w = work + (size_t)b * W;
        double* q1 = w; w += (size_t)T * d;
        double* qd = w; w += (size_t)T * dd;
        double* qs = w; w += (size_t)T * dd;
        double* Ld = w; w += (size_t)T * dd;
        double* Ls = w; w += (size_t)T * dd;
        double* Sd = w; w += (size_t)T * dd;
        double* Ss = w; w += (size_t)T * dd;
        double* yv = w; w += (size_t)T * d;
        double* mu = w; w += (size_t)T * d;
        double* Pd = w; w += (size_t)T * dd;   /* precision = -2 qd, -qs (scratch) */
        double* tmpv = w; w += (size_t)T * d;
        const double* P1 = p1 + (size_t)b * T * d; const double* PD = pd + (size_t)b * T * dd; const double* PS = ps + (size_t)b * (T - 1) * dd;
        double* G1 = g1 + (size_t)b * T * d; double* G2D = g2d + (size_t)b * T * dd; double* G2S = g2s + (size_t)b * (T - 1) * dd;
        double* D1 = d1 + (size_t)b * n * d; double* D2 = d2 + (size_t)b * n * dd;
        const double* Y = y + (size_t)b * n * d;
        double logdetL = 0.0;
        /* ---- update_data_sites: Gaussian likelihood => target (Rinv y, -1/2 Rinv), then refresh ---- */
        for (int i = 0; i < n; ++i) {
            for (int r = 0; r < d; ++r) {
                double t = 0.0;
                for (int c = 0; c < d; ++c) t += Rinv[IDX(r, c, d)] * Y[(size_t)i * d + c];
                D1[(size_t)i * d + r] = (1 - lr_d) * D1[(size_t)i * d + r] + lr_d * t;
                for (int c = 0; c < d; ++c)
                    D2[(size_t)i * dd + IDX(r, c, d)] = (1 - lr_d) * D2[(size_t)i * dd + IDX(r, c, d)] - 0.5 * lr_d * Rinv[IDX(r, c, d)];
            }
        }
        for (int pass = 0; pass < 2; ++pass) {
            /* full_sites: theta_q = theta_p + g + scatter(data) */
            for (size_t i = 0; i < (size_t)T * d; ++i) q1[i] = P1[i] + G1[i];
            for (size_t i = 0; i < (size_t)T * dd; ++i) qd[i] = PD[i] + G2D[i];
            for (size_t i = 0; i < (size_t)(T - 1) * dd; ++i) qs[i] = PS[i] + G2S[i];
            for (int i = 0; i < n; ++i) {
                for (int r = 0; r < d; ++r) q1[(size_t)idx[i] * d + r] += D1[(size_t)i * d + r];
                for (int r = 0; r < dd; ++r) qd[(size_t)idx[i] * dd + r] += D2[(size_t)i * dd + r];
            }
            if (pass == 0) {
                /* update_girsanov_sites: g += lr (scatter(data) - (theta_q - theta_p)), then refresh with the new sites */
                for (size_t i = 0; i < (size_t)T * d; ++i) G1[i] -= lr_g * (q1[i] - P1[i]);
                for (size_t i = 0; i < (size_t)T * dd; ++i) G2D[i] -= lr_g * (qd[i] - PD[i]);
                for (size_t i = 0; i < (size_t)(T - 1) * dd; ++i) G2S[i] -= lr_g * (qs[i] - PS[i]);
                for (int i = 0; i < n; ++i) {
                    for (int r = 0; r < d; ++r) G1[(size_t)idx[i] * d + r] += lr_g * D1[(size_t)i * d + r];
                    for (int r = 0; r < dd; ++r) G2D[(size_t)idx[i] * dd + r] += lr_g * D2[(size_t)i * dd + r];
                }
                /* the reference refreshes the marginals after the data-site update as well: do the sweep */
            }
            for (size_t i = 0; i < (size_t)T * dd; ++i) Pd[i] = -2.0 * qd[i];
            for (size_t i = 0; i < (size_t)(T - 1) * dd; ++i) qs[i] = -qs[i];
            ref_btd_cholesky(Pd, qs, Ld, Ls, T, d);
            ref_btd_solve(Ld, Ls, q1, yv, T, d, 0);
            ref_btd_solve(Ld, Ls, yv, mu, T, d, 1);
            ref_btd_inverse_blocks(Ld, Ls, Sd, Ss, T, d);
            logdetL = ref_btd_logdet(Ld, T, d);
        }
        /* classic_elbo at the final q: VE at the observations - KL(q || p) */
        double ve = 0.0;
        for (int i = 0; i < n; ++i) {
            const double* m = mu + (size_t)idx[i] * d;
            const double* S = Sd + (size_t)idx[i] * dd;
            double quad = 0.0, trc = 0.0;
            for (int r = 0; r < d; ++r)
                for (int c = 0; c < d; ++c) {
                    quad += (Y[(size_t)i * d + r] - m[r]) * Rinv[IDX(r, c, d)] * (Y[(size_t)i * d + c] - m[c]);
                    trc += Rinv[IDX(r, c, d)] * S[IDX(r, c, d)];
                }
            ve += -0.5 * trc - 0.5 * quad - 0.5 * logdetR - 0.5 * d * log(2.0 * M_PI);
        }
        for (size_t i = 0; i < (size_t)T * dd; ++i) Pd[i] = -2.0 * PD[i];
        for (size_t i = 0; i < (size_t)(T - 1) * dd; ++i) qs[i] = -PS[i];
        double tr, mh;
        ref_kl_terms(Sd, Ss, mu, Pd, qs, pmu + (size_t)b * T * d, T, d, &tr, &mh);
        double kl = 0.5 * (tr + mh - (double)T * d + 2.0 * pslc[b] + 2.0 * logdetL);
        (void)tmpv;
        elbo_out[b] = ve - kl;
        total += ve - kl;
    }
    return total;
}


/* ------------------------------------------------------------------------------------------------------------
 * CVI-DP (CVISitesSDE) with a per-dimension cubic Euler map u(x) = alpha x - beta x^3 and diagonal diffusion:
 * closed-form Girsanov KL and its gradient with respect to the expectation parameters, restating
 * oracle/np_sde.py::sde_ssm_kl_closed_form (itself pinned to the reference's quadrature formulation,
 * sde_utils.py:262-359, 473-547, by tests/test_oracle_sde.py).
 * mu [T,d], Sig [T,d,d], Sub [T-1,d,d]; outputs g1 [T,d], gd [T,d,d], gs [T-1,d,d] when want_grads.
 * ------------------------------------------------------------------------------------------------------------ */
static void inv_spd(const double* S, double* out, double* logdet, int d, double* tmp) {
    memcpy(tmp, S, (size_t)d * d * sizeof(double));
    chol_d(tmp, d);
    double ld = 0.0;
    for (int i = 0; i < d; ++i) ld += log(tmp[IDX(i, i, d)]);
    *logdet = 2.0 * ld;
    memset(out, 0, (size_t)d * d * sizeof(double));
    for (int i = 0; i < d; ++i) out[IDX(i, i, d)] = 1.0;
    trsm_l(tmp, out, d, d);
    trsm_lt(tmp, out, d, d);
}

double ref_sde_kl(const double* mu, const double* Sig, const double* Sub, const double* alpha, const double* beta,
                  const double* qdiag, double dt, const double* init_mu, const double* init_cov, int T, int d, int want_grads,
                  double* g1, double* gd, double* gs) {
    const int dd = d * d;
    double* buf = (double*)calloc((size_t)12 * dd + 16 * d, sizeof(double));
    double *P0inv = buf, *tmp = buf + dd, *Sinv = buf + 2 * dd, *A = buf + 3 * dd, *Qq = buf + 4 * dd, *P = buf + 5 * dd,
           *PA = buf + 6 * dd, *AtPA = buf + 7 * dd, *S0inv = buf + 8 * dd;
    double *W = buf + 12 * dd, *ubar = W + d, *J = ubar + d, *V = J + d, *ub_v = V + d, *J_m = ub_v + d, *J_v = J_m + d,
           *V_m = J_v + d, *V_v = V_m + d, *We = V_v + d, *kv = We + d;
    double logdetQp = 0.0, ld;
    for (int i = 0; i < d; ++i) { W[i] = 1.0 / (dt * qdiag[i]); logdetQp += log(dt * qdiag[i]); }
    double* Gm = NULL;
    if (want_grads) {
        Gm = (double*)calloc((size_t)T * d, sizeof(double));
        memset(gd, 0, (size_t)T * dd * sizeof(double));
        memset(gs, 0, (size_t)(T - 1) * dd * sizeof(double));
    }
    double ldP0;
    inv_spd(init_cov, P0inv, &ldP0, d, tmp);
    inv_spd(Sig, S0inv, &ld, d, tmp);
    double kl = 0.0, tr = 0.0, mh = 0.0;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            tr += P0inv[IDX(i, j, d)] * Sig[IDX(i, j, d)];
            mh += (mu[i] - init_mu[i]) * P0inv[IDX(i, j, d)] * (mu[j] - init_mu[j]);
        }
    kl = 0.5 * (tr + mh - d + ldP0 - ld);
    if (want_grads) {
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                Gm[i] += P0inv[IDX(i, j, d)] * (mu[j] - init_mu[j]);
                gd[IDX(i, j, d)] += 0.5 * (P0inv[IDX(i, j, d)] - S0inv[IDX(i, j, d)]);
            }
    }
    for (int t = 0; t < T - 1; ++t) {
        const double *m = mu + (size_t)t * d, *S = Sig + (size_t)t * dd, *C = Sub + (size_t)t * dd, *mn = mu + (size_t)(t + 1) * d,
                     *Sn = Sig + (size_t)(t + 1) * dd;
        for (int i = 0; i < d; ++i) {
            const double al = alpha[i], be = beta[i], mi = m[i], v = S[IDX(i, i, d)], m2 = mi * mi, a = m2 + v;
            ubar[i] = al * mi - be * mi * (m2 + 3 * v);
            J[i] = al - 3 * be * a;
            V[i] = al * al * v - 6 * al * be * v * a + be * be * v * (9 * m2 * m2 + 36 * m2 * v + 15 * v * v);
            ub_v[i] = -3 * be * mi; J_m[i] = -6 * be * mi; J_v[i] = -3 * be;
            V_m[i] = -12 * al * be * mi * v + be * be * mi * v * (36 * m2 + 72 * v);
            V_v[i] = al * al - 6 * al * be * (m2 + 2 * v) + be * be * (9 * m2 * m2 + 72 * m2 * v + 45 * v * v);
        }
        inv_spd(S, Sinv, &ld, d, tmp);
        gemm_d(1.0, C, 0, Sinv, 0, 0.0, A, d);                 /* A = C S^{-1} */
        gemm_d(-1.0, A, 0, C, 1, 0.0, Qq, d);                  /* -A C^T */
        for (int i = 0; i < dd; ++i) Qq[i] += Sn[i];
        double ldQ;
        inv_spd(Qq, P, &ldQ, d, tmp);
        double val = -ldQ + logdetQp - d;
        for (int i = 0; i < d; ++i) {
            double e = ubar[i] - mn[i];
            We[i] = W[i] * e;
            kv[i] = W[i] * C[IDX(i, i, d)];
            val += W[i] * V[i] - 2.0 * J[i] * kv[i] + W[i] * Sn[IDX(i, i, d)] + We[i] * e;
        }
        kl += 0.5 * val;
        if (want_grads) {
            gemm_d(1.0, P, 0, A, 0, 0.0, PA, d);
            gemm_d(1.0, A, 1, PA, 0, 0.0, AtPA, d);
            double* GC = gs + (size_t)t * dd;
            for (int i = 0; i < dd; ++i) GC[i] = PA[i];
            for (int i = 0; i < d; ++i) GC[IDX(i, i, d)] -= W[i] * J[i];
            double* GS = gd + (size_t)t * dd;
            double* GSn = gd + (size_t)(t + 1) * dd;
            for (int i = 0; i < dd; ++i) { GS[i] += -0.5 * AtPA[i]; GSn[i] += -0.5 * P[i]; }
            for (int i = 0; i < d; ++i) {
                GS[IDX(i, i, d)] += 0.5 * W[i] * V_v[i] - kv[i] * J_v[i] + We[i] * ub_v[i];
                GSn[IDX(i, i, d)] += 0.5 * W[i];
                Gm[(size_t)t * d + i] += 0.5 * W[i] * V_m[i] - kv[i] * J_m[i] + We[i] * J[i];
                Gm[(size_t)(t + 1) * d + i] += -We[i];
            }
        }
    }
    if (want_grads) {
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < d; ++i) {
                double v = Gm[(size_t)t * d + i];
                for (int j = 0; j < d; ++j) v -= 2.0 * gd[(size_t)t * dd + IDX(i, j, d)] * mu[(size_t)t * d + j];
                if (t < T - 1)
                    for (int j = 0; j < d; ++j) v -= gs[(size_t)t * dd + IDX(j, i, d)] * mu[(size_t)(t + 1) * d + j];
                if (t > 0)
                    for (int j = 0; j < d; ++j) v -= gs[(size_t)(t - 1) * dd + IDX(i, j, d)] * mu[(size_t)(t - 1) * d + j];
                g1[(size_t)t * d + i] = v;
            }
        free(Gm);
    }
    free(buf);
    return kl;
}

/*
 * One CVISitesSDE (CVI-DP) iteration for B chains under a fixed linearised prior (naturals p1, pd, ps):
 *   update_data_sites(lr_d); update_girsanov_sites(lr_g); classic_elbo()      (cvi_dp_trainer.py:72-75)
 * Gaussian likelihood with precision Rinv.  State updated in place as in ref_cvi_step.
 */
size_t ref_cvi_dp_step_work_doubles(int T, int d) { return (size_t)T * (10 * d * d + 6 * d) + 16 * d * d; }

double ref_cvi_dp_step(int B, int T, int d, int n, const int* idx, const double* y, const double* Rinv, double logdetR,
                       const double* p1, const double* pd, const double* ps, const double* alpha, const double* beta,
                       const double* qdiag, double dt, const double* init_mu, const double* init_cov, double* g1, double* g2d,
                       double* g2s, double* d1, double* d2, double lr_d, double lr_g, double* work, double* elbo_out) {
    const int dd = d * d;
    const size_t W = ref_cvi_dp_step_work_doubles(T, d);
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int b = 0; b < B; ++b) {
        double* w = work + (size_t)b * W;
        double* q1 = w; w += (size_t)T * d;
        double* qd = w; w += (size_t)T * dd;
        double* qs = w; w += (size_t)T * dd;
        double* Ld = w; w += (size_t)T * dd;
        double* Ls = w; w += (size_t)T * dd;
        double* Sd = w; w += (size_t)T * dd;
        double* Ss = w; w += (size_t)T * dd;
        double* yv = w; w += (size_t)T * d;
        double* mu = w; w += (size_t)T * d;
        double* Pd = w; w += (size_t)T * dd;
        double* k1 = w; w += (size_t)T * d;
        double* kd = w; w += (size_t)T * dd;
        double* ks = w; w += (size_t)T * dd;
        const double* P1 = p1 + (size_t)b * T * d; const double* PD = pd + (size_t)b * T * dd; const double* PS = ps + (size_t)b * (T - 1) * dd;
        double* G1 = g1 + (size_t)b * T * d; double* G2D = g2d + (size_t)b * T * dd; double* G2S = g2s + (size_t)b * (T - 1) * dd;
        double* D1 = d1 + (size_t)b * n * d; double* D2 = d2 + (size_t)b * n * dd;
        const double* Y = y + (size_t)b * n * d;
        for (int i = 0; i < n; ++i)
            for (int r = 0; r < d; ++r) {
                double t = 0.0;
                for (int c = 0; c < d; ++c) t += Rinv[IDX(r, c, d)] * Y[(size_t)i * d + c];
                D1[(size_t)i * d + r] = (1 - lr_d) * D1[(size_t)i * d + r] + lr_d * t;
                for (int c = 0; c < d; ++c)
                    D2[(size_t)i * dd + IDX(r, c, d)] = (1 - lr_d) * D2[(size_t)i * dd + IDX(r, c, d)] - 0.5 * lr_d * Rinv[IDX(r, c, d)];
            }
        for (int pass = 0; pass < 2; ++pass) {
            for (size_t i = 0; i < (size_t)T * d; ++i) q1[i] = P1[i] + G1[i];
            for (size_t i = 0; i < (size_t)T * dd; ++i) qd[i] = PD[i] + G2D[i];
            for (size_t i = 0; i < (size_t)(T - 1) * dd; ++i) qs[i] = PS[i] + G2S[i];
            for (int i = 0; i < n; ++i) {
                for (int r = 0; r < d; ++r) q1[(size_t)idx[i] * d + r] += D1[(size_t)i * d + r];
                for (int r = 0; r < dd; ++r) qd[(size_t)idx[i] * dd + r] += D2[(size_t)i * dd + r];
            }
            for (size_t i = 0; i < (size_t)T * dd; ++i) Pd[i] = -2.0 * qd[i];
            for (size_t i = 0; i < (size_t)(T - 1) * dd; ++i) qs[i] = -qs[i];
            ref_btd_cholesky(Pd, qs, Ld, Ls, T, d);
            ref_btd_solve(Ld, Ls, q1, yv, T, d, 0);
            ref_btd_solve(Ld, Ls, yv, mu, T, d, 1);
            ref_btd_inverse_blocks(Ld, Ls, Sd, Ss, T, d);
            if (pass == 0) {
                /* update_girsanov_sites at the refreshed posterior */
                ref_sde_kl(mu, Sd, Ss, alpha, beta, qdiag, dt, init_mu, init_cov, T, d, 1, k1, kd, ks);
                for (size_t i = 0; i < (size_t)T * d; ++i) G1[i] -= lr_g * k1[i];
                for (size_t i = 0; i < (size_t)T * dd; ++i) G2D[i] -= lr_g * kd[i];
                for (size_t i = 0; i < (size_t)(T - 1) * dd; ++i) G2S[i] -= lr_g * ks[i];
                for (int i = 0; i < n; ++i) {
                    for (int r = 0; r < d; ++r) G1[(size_t)idx[i] * d + r] += lr_g * D1[(size_t)i * d + r];
                    for (int r = 0; r < dd; ++r) G2D[(size_t)idx[i] * dd + r] += lr_g * D2[(size_t)i * dd + r];
                }
            }
        }
        double ve = 0.0;
        for (int i = 0; i < n; ++i) {
            const double* m = mu + (size_t)idx[i] * d;
            const double* S = Sd + (size_t)idx[i] * dd;
            double quad = 0.0, trc = 0.0;
            for (int r = 0; r < d; ++r)
                for (int c = 0; c < d; ++c) {
                    quad += (Y[(size_t)i * d + r] - m[r]) * Rinv[IDX(r, c, d)] * (Y[(size_t)i * d + c] - m[c]);
                    trc += Rinv[IDX(r, c, d)] * S[IDX(r, c, d)];
                }
            ve += -0.5 * trc - 0.5 * quad - 0.5 * logdetR - 0.5 * d * log(2.0 * M_PI);
        }
        double kl = ref_sde_kl(mu, Sd, Ss, alpha, beta, qdiag, dt, init_mu, init_cov, T, d, 0, NULL, NULL, NULL);
        elbo_out[b] = ve - kl;
        total += ve - kl;
    }
    return total;
}


/* ---- VDP (VariationalMarkovGP, markovflow/models/vi_sde.py) for a per-dimension cubic drift f_i(x) = af x_i - bf x_i^3 and diagonal q ----
 * Restates oracle/np_models.VariationalMarkovGP (closed_form = True), which restates the reference line by line.
 * Layout per trajectory: A [N,d,d], b [N,d] (the posterior drift is -A x + b), m [T,d], S [T,d,d] with T = N + 1. */
static double vdp_fix(double x, double lo, double hi) {
    if (isnan(x)) x = 1e-8;
    return x < lo ? lo : (x > hi ? hi : x);
}
/* E u, E u', Var u of u(x) = af x - bf x^3 under N(m, v) and the partials the gradients need (oracle/np_sde.cubic_moments) */
static void vdp_cubic(double af, double bf, double m, double v, double* ubar, double* J, double* V, double* ubar_v, double* J_m,
                      double* J_v, double* V_m, double* V_v) {
    const double a = m * m + v;
    *ubar = af * m - bf * (m * m * m + 3 * m * v);
    *J = af - 3 * bf * a;
    *V = af * af * v - 6 * af * bf * v * a + bf * bf * (9 * m * m * m * m * v + 36 * m * m * v * v + 15 * v * v * v);
    *ubar_v = -3 * bf * m;
    *J_m = -6 * bf * m;
    *J_v = -3 * bf;
    *V_m = -12 * af * bf * m * v + bf * bf * (36 * m * m * m * v + 72 * m * v * v);
    *V_v = af * af - 6 * af * bf * (m * m + 2 * v) + bf * bf * (9 * m * m * m * m + 72 * m * m * v + 45 * v * v);
}
/* forward_pass (vi_sde.py:171-204): marginals of x_{k+1} = (I - A_k dt) x_k + b_k dt + N(0, q dt), by the moment recursion;
 * stabilize: NaN -> 1e-8 and clipping of the transition parameters to [-1, 1] (vi_sde.py:186-200) */
static void vdp_forward_one(int T, int d, const double* A, const double* b, const double* qdiag, double dt, const double* q0_mu,
                            const double* q0_chol, int stabilize, double* m, double* S, double* tmp) {
    const int dd = d * d, N = T - 1;
    double *At = tmp, *AS = tmp + dd;
    memcpy(m, q0_mu, (size_t)d * sizeof(double));
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double t = 0.0;
            for (int k = 0; k < d; ++k) t += q0_chol[IDX(i, k, d)] * q0_chol[IDX(j, k, d)];
            S[IDX(i, j, d)] = t;
        }
    for (int k = 0; k < N; ++k) {
        const double *Ak = A + (size_t)k * dd, *bk = b + (size_t)k * d, *mk = m + (size_t)k * d, *Sk = S + (size_t)k * dd;
        double *mn = m + (size_t)(k + 1) * d, *Sn = S + (size_t)(k + 1) * dd;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double a = -Ak[IDX(i, j, d)] * dt + (i == j ? 1.0 : 0.0);
                At[IDX(i, j, d)] = stabilize ? vdp_fix(a, -1.0, 1.0) : a;
            }
        for (int i = 0; i < d; ++i) {
            double bt = bk[i] * dt;
            if (stabilize) bt = vdp_fix(bt, -1.0, 1.0);
            double t = bt;
            for (int j = 0; j < d; ++j) t += At[IDX(i, j, d)] * mk[j];
            mn[i] = t;
        }
        gemm_d(1.0, At, 0, Sk, 0, 0.0, AS, d);
        gemm_d(1.0, AS, 0, At, 1, 0.0, Sn, d);
        for (int i = 0; i < d; ++i) Sn[IDX(i, i, d)] += qdiag[i] * dt;
    }
}
/* E_sde (vi_sde.py:416-434 via oracle/np_sde.e_sde_closed_form) and, when dm / dS are given, its gradients divided by dt */
static double vdp_esde_one(int N, int d, double af, double bf, const double* qdiag, double dt, const double* A, const double* b,
                           const double* m, const double* S, double* dEdm, double* dEdS, double* tmp) {
    const int dd = d * d;
    double* LS = tmp;
    double E = 0.0;
    for (int n = 0; n < N; ++n) {
        const double *An = A + (size_t)n * dd, *bn = b + (size_t)n * d, *mn = m + (size_t)n * d, *Sn = S + (size_t)n * dd;
        double* gm = dEdm ? dEdm + (size_t)n * d : NULL;
        double* gS = dEdS ? dEdS + (size_t)n * dd : NULL;
        if (gm) memset(gm, 0, (size_t)d * sizeof(double));
        if (gS) memset(gS, 0, (size_t)dd * sizeof(double));
        for (int i = 0; i < d; ++i)                                  /* L = -A */
            for (int j = 0; j < d; ++j) {
                double t = 0.0;
                for (int k = 0; k < d; ++k) t -= An[IDX(i, k, d)] * Sn[IDX(k, j, d)];
                LS[IDX(i, j, d)] = t;
            }
        for (int i = 0; i < d; ++i) {
            double El = bn[i], LSL = 0.0;
            for (int k = 0; k < d; ++k) {
                El -= An[IDX(i, k, d)] * mn[k];
                LSL -= LS[IDX(i, k, d)] * An[IDX(i, k, d)];
            }
            const double LSii = LS[IDX(i, i, d)];
            double Ef, Jf, Vf, ubar_v, J_m, J_v, V_m, V_v;
            vdp_cubic(af, bf, mn[i], Sn[IDX(i, i, d)], &Ef, &Jf, &Vf, &ubar_v, &J_m, &J_v, &V_m, &V_v);
            const double r = El - Ef, w = 1.0 / qdiag[i];
            E += 0.5 * dt * w * (LSL - 2.0 * LSii * Jf + Vf + r * r);
            if (gm) {
                for (int k = 0; k < d; ++k) gm[k] += 0.5 * w * 2.0 * r * (-An[IDX(i, k, d)] - (k == i ? Jf : 0.0));
                gm[i] += 0.5 * w * (-2.0 * LSii * J_m + V_m);
                for (int j = 0; j < d; ++j)
                    for (int k = 0; k < d; ++k) {
                        const double lj = -An[IDX(i, j, d)], lk = -An[IDX(i, k, d)];
                        gS[IDX(j, k, d)] += 0.5 * w * (lj * lk - Jf * (lj * (k == i ? 1.0 : 0.0) + (j == i ? 1.0 : 0.0) * lk));
                    }
                gS[IDX(i, i, d)] += 0.5 * w * (-2.0 * LSii * J_v + V_v - 2.0 * r * ubar_v);
            }
        }
    }
    return E;
}

size_t ref_vdp_work_doubles(int T, int d) { return (size_t)T * (2 * d * d + 2 * d) + 8 * d * d; }

/* marginals of the current (A, b) for B trajectories */
void ref_vdp_forward(int B, int T, int d, const double* A, const double* b, const double* qdiag, double dt, const double* q0_mu,
                     const double* q0_chol, int stabilize, double* m, double* S) {
    const int dd = d * d, N = T - 1;
#pragma omp parallel for schedule(static)
    for (int bb = 0; bb < B; ++bb) {
        double tmp[2 * 64];
        double* t = (2 * dd <= 128) ? tmp : (double*)malloc((size_t)2 * dd * sizeof(double));
        vdp_forward_one(T, d, A + (size_t)bb * N * dd, b + (size_t)bb * N * d, qdiag, dt, q0_mu, q0_chol, stabilize, m + (size_t)bb * T * d,
                        S + (size_t)bb * T * dd, t);
        if (t != tmp) free(t);
    }
}

/*
 * One iteration of VIMarkovGPTrainer.perform_inference's loop body (vi_markov_gp_trainer.py:55-75) for B trajectories:
 *   update_lagrange(m, S); update_param(m, S, lr); (m, S) = forward_pass(); elbo()
 * m, S: on entry the marginals of the current (A, b) (ref_vdp_forward), on return those of the updated ones.  Gaussian likelihood with
 * precision Rinv at the grid indices idx [n] (sorted).  q(x0) = N(q0_mu, q0_chol q0_chol^T) is kept fixed, p(x0) = N(p0_mu, p0_cov).
 */
double ref_vdp_step(int B, int T, int d, int n, const int* idx, const double* y, const double* Rinv, double logdetR, double af,
                    double bf, const double* qdiag, double dt, const double* q0_mu, const double* q0_chol, const double* p0_mu,
                    const double* p0_cov, double* A, double* b, double* m, double* S, double lr, int stabilize, double* work,
                    double* elbo_out) {
    const int dd = d * d, N = T - 1;
    const size_t W = ref_vdp_work_doubles(T, d);
    const double CLO = -5000.0, CHI = 5000.0;                     /* vi_sde.py:59-60 */
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int bb = 0; bb < B; ++bb) {
        double* w = work + (size_t)bb * W;
        double* dEdm = w; w += (size_t)T * d;
        double* dEdS = w; w += (size_t)T * dd;
        double* lam = w; w += (size_t)T * d;
        double* psi = w; w += (size_t)T * dd;
        double* tmp = w;
        double *Ab = A + (size_t)bb * N * dd, *bv = b + (size_t)bb * N * d, *mb = m + (size_t)bb * T * d, *Sb = S + (size_t)bb * T * dd;
        const double* Y = y + (size_t)bb * n * d;
        /* update_lagrange (vi_sde.py:289-347) */
        vdp_esde_one(N, d, af, bf, qdiag, dt, Ab, bv, mb, Sb, dEdm, dEdS, tmp);
        if (stabilize) {
            for (size_t i = 0; i < (size_t)N * d; ++i) dEdm[i] = vdp_fix(dEdm[i], CLO, CHI);
            for (size_t i = 0; i < (size_t)N * dd; ++i) dEdS[i] = vdp_fix(dEdS[i], CLO, CHI);
        }
        memset(lam, 0, (size_t)N * d * sizeof(double));
        memset(psi, 0, (size_t)N * dd * sizeof(double));
        for (int t = 0; t < N; ++t)
            for (int i = 0; i < d; ++i) psi[(size_t)t * dd + IDX(i, i, d)] = 1e-10;
        int io = n - 1;
        for (int t = N - 1; t >= 1; --t) {
            const double *At = Ab + (size_t)t * dd, *pt = psi + (size_t)t * dd, *lt = lam + (size_t)t * d;
            double *pp = psi + (size_t)(t - 1) * dd, *lp = lam + (size_t)(t - 1) * d;
            while (io >= 0 && idx[io] > t) --io;
            const int has = (io >= 0 && idx[io] == t);
            for (int i = 0; i < d; ++i) {
                double dl = -dEdm[(size_t)t * d + i];
                for (int k = 0; k < d; ++k) dl += At[IDX(i, k, d)] * lt[k];
                double om = 0.0;
                if (has) {
                    for (int c = 0; c < d; ++c) om += Rinv[IDX(i, c, d)] * (Y[(size_t)io * d + c] - mb[(size_t)t * d + c]);
                    if (stabilize) om = vdp_fix(om, CLO, CHI);
                }
                lp[i] = lt[i] - dt * dl - om;
                for (int j = 0; j < d; ++j) {
                    double dp = -dEdS[(size_t)t * dd + IDX(i, j, d)];
                    for (int k = 0; k < d; ++k) dp += 2.0 * pt[IDX(i, k, d)] * At[IDX(k, j, d)];       /* psi A + psi A, as written there */
                    double oS = has ? -0.5 * Rinv[IDX(i, j, d)] : 0.0;
                    if (has && stabilize) oS = vdp_fix(oS, CLO, CHI);
                    pp[IDX(i, j, d)] = pt[IDX(i, j, d)] - dt * dp - oS;
                }
            }
        }
        /* update_param (vi_sde.py:349-414) */
        for (int t = 0; t < N; ++t) {
            double *At = Ab + (size_t)t * dd, *bt = bv + (size_t)t * d;
            const double *mt = mb + (size_t)t * d, *St = Sb + (size_t)t * dd;
            double *pt = psi + (size_t)t * dd, *lt = lam + (size_t)t * d;
            double Atil[64], Ef[8];
            double* At_ = (dd <= 64) ? Atil : tmp;
            for (int i = 0; i < d; ++i) {
                double Jf, Vf, u1, u2, u3, u4, u5, ef;
                vdp_cubic(af, bf, mt[i], St[IDX(i, i, d)], &ef, &Jf, &Vf, &u1, &u2, &u3, &u4, &u5);
                if (d <= 8) Ef[i] = ef; else tmp[dd + i] = ef;
                for (int j = 0; j < d; ++j) {
                    double p = pt[IDX(i, j, d)];
                    if (stabilize) p = vdp_fix(p, CLO, CHI);
                    At_[IDX(i, j, d)] = (i == j ? -Jf : 0.0) + 2.0 * qdiag[i] * p;
                }
            }
            for (int i = 0; i < d; ++i) {
                double l = lt[i];
                if (stabilize) l = vdp_fix(l, CLO, CHI);
                double bt_ = ((d <= 8) ? Ef[i] : tmp[dd + i]) - qdiag[i] * l;
                for (int j = 0; j < d; ++j) bt_ += At_[IDX(i, j, d)] * mt[j];
                bt[i] = (1.0 - lr) * bt[i] + lr * bt_;
            }
            for (int e = 0; e < dd; ++e) At[e] = (1.0 - lr) * At[e] + lr * At_[e];
        }
        /* forward_pass and elbo (vi_sde.py:171-204, 436-455) */
        vdp_forward_one(T, d, Ab, bv, qdiag, dt, q0_mu, q0_chol, stabilize, mb, Sb, tmp);
        double ve = 0.0;
        for (int i = 0; i < n; ++i) {
            const double* mm = mb + (size_t)idx[i] * d;
            const double* SS = Sb + (size_t)idx[i] * dd;
            double quad = 0.0, trc = 0.0;
            for (int r = 0; r < d; ++r)
                for (int c = 0; c < d; ++c) {
                    quad += (Y[(size_t)i * d + r] - mm[r]) * Rinv[IDX(r, c, d)] * (Y[(size_t)i * d + c] - mm[c]);
                    trc += Rinv[IDX(r, c, d)] * SS[IDX(r, c, d)];
                }
            ve += -0.5 * trc - 0.5 * quad - 0.5 * logdetR - 0.5 * d * log(2.0 * M_PI);
        }
        const double esde = vdp_esde_one(N, d, af, bf, qdiag, dt, Ab, bv, mb, Sb, NULL, NULL, tmp);
        /* KL[q(x0) || p(x0)] */
        double *S0 = tmp, *Pinv = tmp + dd, *t2 = tmp + 2 * dd;
        double ld0 = 0.0, ldp = 0.0;
        for (int i = 0; i < d; ++i) {
            ld0 += 2.0 * log(fabs(q0_chol[IDX(i, i, d)]));
            for (int j = 0; j < d; ++j) {
                double t = 0.0;
                for (int k = 0; k < d; ++k) t += q0_chol[IDX(i, k, d)] * q0_chol[IDX(j, k, d)];
                S0[IDX(i, j, d)] = t;
            }
        }
        inv_spd(p0_cov, Pinv, &ldp, d, t2);
        double tr0 = 0.0, mah = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                tr0 += Pinv[IDX(i, j, d)] * S0[IDX(j, i, d)];
                mah += (p0_mu[i] - q0_mu[i]) * Pinv[IDX(i, j, d)] * (p0_mu[j] - q0_mu[j]);
            }
        const double kl0 = 0.5 * (tr0 + mah - d + ldp - ld0);
        elbo_out[bb] = ve - esde - kl0;
        total += elbo_out[bb];
    }
    return total;
}

/* number of OpenMP threads later calls may use (the baseline is timed with all host cores and with one) */
/* ------------------------------------------------------------------------------------------------------------
 * CVI for GP regression with a state-space kernel (CVIGaussianProcess, variational_cvi.py:225-421), one chain, scalar output
 * f_t = h . s_t, zero-mean prior, scalar Gaussian likelihood of variance s2: one iteration of
 *   update_sites(); elbo()
 * as oracle/np_models.CVIGaussianProcess does them (predict_f from the posterior of the current sites -- factorisation, two
 * substitutions, selected inverse --, the likelihood's gradients with respect to the expectation parameters, the damped site update;
 * then KalmanFilterWithSites.log_likelihood, kalman_filter.py:184-255, with the new sites: a second factorisation).
 * Pd [T,d,d], Ps [T-1,d,d]: prior precision blocks; half_logdet_prior = 1/2 log|K_prior^-1|; nat1, nat2 [T]: the sites, in place.
 * work: ref_cvigp_step_work_doubles(T, d) doubles.  Returns the ELBO.
 * ------------------------------------------------------------------------------------------------------------ */
size_t ref_cvigp_step_work_doubles(int T, int d) { return (size_t)T * (5 * d * d + 3 * d); }

double ref_cvigp_step(int T, int d, const double* Pd, const double* Ps, double half_logdet_prior, const double* h, const double* y,
                      double s2, double lr, double* nat1, double* nat2, double* work) {
    const int dd = d * d;
    double* D = work;
    double* Ld = D + (size_t)T * dd;
    double* Ls = Ld + (size_t)T * dd;
    double* Sd = Ls + (size_t)T * dd;
    double* Ss = Sd + (size_t)T * dd;
    double* lin = Ss + (size_t)T * dd;
    double* yv = lin + (size_t)T * d;
    double* mu = yv + (size_t)T * d;
    /* update_sites: posterior of the current sites, marginals of f at the data */
    for (int t = 0; t < T; ++t) {
        const double rinv = -2.0 * nat2[t];
        for (int r = 0; r < d; ++r) {
            lin[(size_t)t * d + r] = h[r] * nat1[t];
            for (int c = 0; c < d; ++c) D[(size_t)t * dd + IDX(r, c, d)] = Pd[(size_t)t * dd + IDX(r, c, d)] + rinv * h[r] * h[c];
        }
    }
    ref_btd_cholesky(D, Ps, Ld, Ls, T, d);
    ref_btd_solve(Ld, Ls, lin, yv, T, d, 0);
    ref_btd_solve(Ld, Ls, yv, mu, T, d, 1);
    ref_btd_inverse_blocks(Ld, Ls, Sd, Ss, T, d);
    for (int t = 0; t < T; ++t) {
        double fm = 0.0, fv = 0.0;
        for (int r = 0; r < d; ++r) {
            fm += h[r] * mu[(size_t)t * d + r];
            for (int c = 0; c < d; ++c) fv += h[r] * Sd[(size_t)t * dd + IDX(r, c, d)] * h[c];
        }
        /* gradients of the variational expectations wrt (mu, var), then wrt the expectation parameters [mu, var + mu^2] */
        const double dmu = (y[t] - fm) / s2, dvar = -0.5 / s2 + 0.0 * fv;
        const double g1 = dmu - 2.0 * dvar * fm, g2 = dvar;
        nat1[t] = (1.0 - lr) * nat1[t] + lr * g1;
        nat2[t] = (1.0 - lr) * nat2[t] + lr * g2;
    }
    /* elbo: the log marginal likelihood of the model whose likelihood terms are the (new) Gaussian sites */
    double term1 = 0.0, logdet_r = 0.0;
    for (int t = 0; t < T; ++t) {
        const double rinv = -2.0 * nat2[t], obs = -0.5 * nat1[t] / nat2[t];
        term1 += rinv * obs * obs;
        logdet_r += log(rinv);
        for (int r = 0; r < d; ++r) {
            lin[(size_t)t * d + r] = h[r] * rinv * obs;
            for (int c = 0; c < d; ++c) D[(size_t)t * dd + IDX(r, c, d)] = Pd[(size_t)t * dd + IDX(r, c, d)] + rinv * h[r] * h[c];
        }
    }
    ref_btd_cholesky(D, Ps, Ld, Ls, T, d);
    ref_btd_solve(Ld, Ls, lin, yv, T, d, 0);
    double term2 = 0.0;
    for (size_t i = 0; i < (size_t)T * d; ++i) term2 += yv[i] * yv[i];
    return -0.5 * log(2.0 * M_PI) * T - 0.5 * term1 + 0.5 * term2 + half_logdet_prior - ref_btd_logdet(Ld, T, d) + 0.5 * logdet_r;
}

/* ------------------------------------------------------------------------------------------------------------
 * Sparse / inducing-state CVI (SparseCVIGaussianProcess, sparse_variational_cvi.py:38-292), one chain of M inducing states of
 * dimension d, scalar output, Gaussian likelihood of variance s2, zero-mean prior: one iteration of
 *   update_sites(data); classic_elbo(data)
 * as oracle/np_conditionals.SparseCVIGaussianProcess does them.  The sites live on the M + 1 intervals between consecutive inducing
 * states (nat1 [M+1, 2d], nat2 [M+1, 2d, 2d]; interval j couples states j-1 and j, the two end intervals have one real state and a
 * virtual one with the prior's stationary covariance Pinf).  Data point n lies in interval idx[n]; w [N, 2d] = h^T P_n and c [N] =
 * h^T T_n h are the conditional statistics of f(t_n) given the two states (conditionals.py:207-256), computed once by the caller.
 * Pd [M,d,d], Ps [M-1,d,d]: prior precision blocks; pslc: the prior's sum of log Cholesky diagonals (-1/2 log|K^-1|).
 * work: ref_sparse_cvi_step_work_doubles(M, d) doubles.  Returns the ELBO.
 * ------------------------------------------------------------------------------------------------------------ */
size_t ref_sparse_cvi_step_work_doubles(int M, int d) { return (size_t)M * (7 * d * d + 3 * d) + (size_t)(M + 1) * (2 * d + 4 * d * d); }

static void sparse_posterior(int M, int d, const double* Pd, const double* Ps, const double* nat1, const double* nat2, double* D,
                             double* S, double* lin, double* Ld, double* Ls, double* yv, double* mu, double* Sd, double* Ss,
                             double* logdetL) {
    const int dd = d * d, d2 = 2 * d;
    /* precision = prior precision - 2 (overlap-added nat2), linear part = overlap-added nat1 */
    for (int k = 0; k < M; ++k) {
        const double* a1 = nat1 + (size_t)(k + 1) * d2;          /* interval k+1: its left half belongs to state k */
        const double* b1 = nat1 + (size_t)k * d2 + d;            /* interval k: its right half belongs to state k */
        const double* a2 = nat2 + (size_t)(k + 1) * d2 * d2;
        const double* b2 = nat2 + (size_t)k * d2 * d2;
        for (int r = 0; r < d; ++r) {
            lin[(size_t)k * d + r] = a1[r] + b1[r];
            for (int c = 0; c < d; ++c)
                D[(size_t)k * dd + IDX(r, c, d)] = Pd[(size_t)k * dd + IDX(r, c, d)] - 2.0 * (a2[(size_t)r * d2 + c] + b2[(size_t)(d + r) * d2 + d + c]);
        }
        if (k < M - 1)
            for (int r = 0; r < d; ++r)
                for (int c = 0; c < d; ++c)                      /* block (state k+1, state k) of interval k+1: rows d.., columns ..d */
                    S[(size_t)k * dd + IDX(r, c, d)] = Ps[(size_t)k * dd + IDX(r, c, d)] - 2.0 * a2[(size_t)(d + r) * d2 + c];
    }
    ref_btd_cholesky(D, S, Ld, Ls, M, d);
    ref_btd_solve(Ld, Ls, lin, yv, M, d, 0);
    ref_btd_solve(Ld, Ls, yv, mu, M, d, 1);
    ref_btd_inverse_blocks(Ld, Ls, Sd, Ss, M, d);
    *logdetL = ref_btd_logdet(Ld, M, d);
}

/* mean and variance of f at data point n from the marginals of the two states of its interval */
static void sparse_predict(int M, int d, int j, const double* w, double c, const double* mu, const double* Sd, const double* Ss,
                           const double* Pinf, double* fm, double* fv) {
    const int dd = d * d;
    const double* wl = w;
    const double* wr = w + d;
    const double* SL = (j > 0) ? Sd + (size_t)(j - 1) * dd : Pinf;
    const double* SR = (j < M) ? Sd + (size_t)j * dd : Pinf;
    double m = 0.0, v = c;
    for (int r = 0; r < d; ++r) {
        if (j > 0) m += wl[r] * mu[(size_t)(j - 1) * d + r];
        if (j < M) m += wr[r] * mu[(size_t)j * d + r];
        for (int q = 0; q < d; ++q) {
            v += wl[r] * SL[IDX(r, q, d)] * wl[q] + wr[r] * SR[IDX(r, q, d)] * wr[q];
            if (j > 0 && j < M) v += 2.0 * wr[r] * Ss[(size_t)(j - 1) * dd + IDX(r, q, d)] * wl[q];      /* Sigma_{j,j-1} */
        }
    }
    *fm = m;
    *fv = v;
}

double ref_sparse_cvi_step(int M, int d, int N, const int* idx, const double* w, const double* c, const double* y, double s2, double lr,
                           const double* Pd, const double* Ps, double pslc, const double* Pinf, double* nat1, double* nat2, double* work) {
    const int dd = d * d, d2 = 2 * d;
    double* D = work;
    double* S = D + (size_t)M * dd;
    double* Ld = S + (size_t)M * dd;
    double* Ls = Ld + (size_t)M * dd;
    double* Sd = Ls + (size_t)M * dd;
    double* Ss = Sd + (size_t)M * dd;
    double* Pp = Ss + (size_t)M * dd;                 /* scratch for the KL terms */
    double* lin = Pp + (size_t)M * dd;
    double* yv = lin + (size_t)M * d;
    double* mu = yv + (size_t)M * d;
    double* s1 = mu + (size_t)M * d;
    double* sq = s1 + (size_t)(M + 1) * d2;
    double logdetL;
    /* update_sites */
    sparse_posterior(M, d, Pd, Ps, nat1, nat2, D, S, lin, Ld, Ls, yv, mu, Sd, Ss, &logdetL);
    memset(s1, 0, (size_t)(M + 1) * d2 * sizeof(double));
    memset(sq, 0, (size_t)(M + 1) * d2 * d2 * sizeof(double));
    for (int n = 0; n < N; ++n) {
        const int j = idx[n];
        const double* wn = w + (size_t)n * d2;
        double fm, fv;
        sparse_predict(M, d, j, wn, c[n], mu, Sd, Ss, Pinf, &fm, &fv);
        const double dmu = (y[n] - fm) / s2, dvar = -0.5 / s2 + 0.0 * fv;
        const double g1 = dmu - 2.0 * dvar * fm, g2 = dvar;
        for (int r = 0; r < d2; ++r) {
            s1[(size_t)j * d2 + r] += wn[r] * g1;
            for (int q = 0; q < d2; ++q) sq[(size_t)j * d2 * d2 + (size_t)r * d2 + q] += g2 * wn[r] * wn[q];
        }
    }
    for (size_t i = 0; i < (size_t)(M + 1) * d2; ++i) nat1[i] = (1.0 - lr) * nat1[i] + lr * s1[i];
    for (size_t i = 0; i < (size_t)(M + 1) * d2 * d2; ++i) nat2[i] = (1.0 - lr) * nat2[i] + lr * sq[i];
    /* classic_elbo with the new sites */
    sparse_posterior(M, d, Pd, Ps, nat1, nat2, D, S, lin, Ld, Ls, yv, mu, Sd, Ss, &logdetL);
    double ve = 0.0;
    for (int n = 0; n < N; ++n) {
        double fm, fv;
        sparse_predict(M, d, idx[n], w + (size_t)n * d2, c[n], mu, Sd, Ss, Pinf, &fm, &fv);
        ve += -0.5 * log(2.0 * M_PI) - 0.5 * log(s2) - 0.5 * ((y[n] - fm) * (y[n] - fm) + fv) / s2;
    }
    memset(lin, 0, (size_t)M * d * sizeof(double));     /* prior means: zero */
    double tr, mh;
    ref_kl_terms(Sd, Ss, mu, Pd, Ps, lin, M, d, &tr, &mh);
    (void)Pp;
    return ve - 0.5 * (tr + mh - (double)M * d + 2.0 * pslc + 2.0 * logdetL);
}

void ref_set_num_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
