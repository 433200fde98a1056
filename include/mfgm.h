/*
 * mfgm -- MI355X-native block-tri-diagonal Gauss-Markov kernels (C ABI).
 *
 * Drop-in boundary for the native layer under the reference's block_tri_diag / state_space_model /
 * kalman_filter / ssm_gaussian_transformations modules, i.e. for the TF custom ops of
 * banded-matrices==0.0.6 that AaltoML/vi-diffusion-processes binds at
 *     markovflow/block_tri_diag.py:22-31        (cholesky_band, solve_triang_mat, product_band_mat,
 *                                                inverse_from_cholesky_band, block_to_band, band_to_block)
 *     markovflow/ssm_gaussian_transformations.py:23 (inverse_from_cholesky_band, solve_triang_band)
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (fp64, contiguous); `stream` is a
 *     hipStream_t passed as void*; calls are asynchronous on that stream and re-entrant across streams
 *     as long as each in-flight call has its own workspace;
 *   - no hidden allocation: scratch comes from the caller (`ws`, mfgm_plan_workspace_bytes());
 *   - return value: 0 ok, 1 bad argument / unsupported shape, 3 HIP runtime error.  A block that is not
 *     positive definite does not abort the sweep: the optional device word `info` becomes non-zero
 *     (the Python wrapper turns that into ArithmeticError, the stand-in for TF's Cholesky failure);
 *   - "natural" layout = the reference's row-major [B, T, d, d] / [B, T-1, d, d] / [B, T, d] tensors;
 *     "packed" layout = the segment-interleaved layout the sweeps run on (csrc/mfgm_layout.h):
 *     element (chain b, node t = p*R + s, e) of a per-node quantity with E doubles is at
 *     (((lane/64)*R + s)*E + e)*64 + lane%64 with lane = b*P + p.  Symmetric and lower-triangular blocks are stored as packed lower
 *     triangles (E = d(d+1)/2) in the packed layout.
 */
#ifndef MFGM_H
#define MFGM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfgm_plan mfgm_plan;

/* kinds of per-node quantities */
#define MFGM_VEC 0  /* [d]                                   */
#define MFGM_FULL 1 /* [d, d] general block                  */
#define MFGM_SYM 2  /* [d, d] symmetric: lower triangle read, both triangles written */
#define MFGM_TRI 3  /* [d, d] lower-triangular: upper triangle written as zero       */

/* Partition plan for B chains of T nodes with d x d blocks.  R0 = nodes per lane segment at the finest
 * level (0 = choose so that about 64 Ki lanes exist), Rup = segment length of the coarser levels (0 = 4).
 * Supported d: 1..32.  d <= 8 runs the lane-per-segment kernels on the packed wave-tiled layout; 8 < d <= 32 runs the
 * wavefront-per-segment ("wide") kernels, for which the "packed" arrays are simply the natural [B, T, d*d] / [B, T, d]
 * arrays (mfgm_pack / mfgm_unpack then only symmetrise / zero-fill) and R0 = 0 chooses about 8 Ki segments.  The wide
 * path covers pack/unpack, factor, selinv, lincomb, node_io, ssm_to_naturals and kl_terms; the SDE / VDP / stationary-
 * kernel entry points return 1 for d > 8. */
int mfgm_plan_create(int B, int T, int d, int R0, int Rup, mfgm_plan** out);
void mfgm_plan_destroy(mfgm_plan* plan);
/* out[0..5] = nlevels, R (level 0), P (level 0), Lpad (level 0), B, T */
int mfgm_plan_describe(const mfgm_plan* plan, int* out6);
size_t mfgm_plan_workspace_bytes(const mfgm_plan* plan);
/* number of doubles of a level-0 packed array of the given kind */
size_t mfgm_packed_doubles(const mfgm_plan* plan, int kind);

/* natural -> packed and back.  n_nodes = T for per-state tensors, T-1 for per-transition tensors
 * (block_tri_diag.py:206-237 `_convert_to_band`, :553-596 `_banded_to_block_tri` are the reference's
 * re-layout steps these replace). */
int mfgm_pack(const mfgm_plan* plan, int kind, const double* natural, int n_nodes, double* packed, void* stream);
int mfgm_unpack(const mfgm_plan* plan, int kind, const double* packed, double* natural, int n_nodes, void* stream);

/* Block Cholesky of the symmetric block-tri-diagonal matrix with diagonal blocks aD*D_t and
 * sub-diagonal blocks aS*S_t (packed), plus y = L^{-1} (aR*r) when r != NULL:
 *   replaces SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:428-440 -> cholesky_band),
 *   LowerTriangularBlockTriDiagonal.solve(transpose_left=False) (:339-351 -> solve_triang_mat) and
 *   abs_log_det (:353-366).
 * Outputs (packed): L (TRI), G = L_{t+1,t} at node t (FULL), y (VEC, may be NULL iff r is NULL),
 * logdet[B] = sum log diag(L) and quad[B] = |y|^2 (either may be NULL). */
int mfgm_packed_factor(const mfgm_plan* plan, const double* D, const double* S, const double* r, double aD,
                       double aS, double aR, double* L, double* G, double* y, double* logdet, double* quad,
                       void* ws, int* info, void* stream);

/* Diagonal and sub-diagonal blocks of (L L^T)^{-1} and x = L^{-T} y from the factor above:
 *   replaces block_diagonal_of_inverse (block_tri_diag.py:318-337 -> inverse_from_cholesky_band),
 *   the sub-diagonal read at ssm_gaussian_transformations.py:443-458, and solve(transpose_left=True).
 * Must follow mfgm_packed_factor with the same plan and workspace (coarse-level factors live in ws).
 * Outputs (packed): Sig (SYM), Sub = Sigma_{t+1,t} at node t (FULL, may be NULL), x (VEC, NULL iff y NULL). */
int mfgm_packed_selinv(const mfgm_plan* plan, const double* L, const double* G, const double* y, double* Sig,
                       double* Sub, double* x, void* ws, void* stream);

/* The same pair for callers that only want what the selected inverse delivers (marginal blocks, means, log-determinant, quadratic
 * form) and never read the factor itself.  form 0: exactly mfgm_packed_factor / mfgm_packed_selinv.  form 1 (plans with 8 < d <= 32
 * only, otherwise error 1): INVERSE FORM -- the arrays L, G, y receive (F_t^{-1}, S_t F_t^{-1}, F_t^{-1} h_t) for the pivot blocks F_t of
 * the elimination instead of (L_tt, L_{t+1,t}, y_t), logdet / quad are unchanged, and the arrays are only meaningful as inputs of
 * mfgm_packed_selinv_form(form 1) of the same plan: the backward pass then needs no factorisation at all and the forward passes
 * invert F_t with MFMA block sweeps (csrc/mfgm_mfma_inv.h). */
int mfgm_packed_factor_form(const mfgm_plan* plan, int form, const double* D, const double* S, const double* r, double aD, double aS,
                            double aR, double* L, double* G, double* y, double* logdet, double* quad, void* ws, int* info,
                            void* stream);
int mfgm_packed_selinv_form(const mfgm_plan* plan, int form, const double* L, const double* G, const double* y, double* Sig,
                            double* Sub, double* x, void* ws, void* stream);

/* The scalar assembly of a bound from its per-chain terms in one launch (kalman_filter.py:229-255: cst + term1 + term2 + term3;
 * state_space_model.py:557-593): out[i] = c + ce * extra[i] + sum_k w[k] * terms[k][i] for i < n chains (terms device [n_terms, n],
 * n_terms <= 8, w HOST [n_terms], extra device [n] or NULL), total = sum_i out[i] (out or total may be NULL).  info (device, may be
 * NULL): when non-zero -- a pivot block of the factorisation behind the terms was not positive definite -- every value is NaN, so
 * that a hot loop needs no host synchronisation to notice (mfgm_packed_factor conventions). */
int mfgm_combine_terms(int n_terms, int n, const double* terms, const double* w, double c, const double* extra, double ce, const int* info,
                       double* out, double* total, void* stream);

/* out = a*x + b*y + c*z over n doubles (y and z may be NULL): the element-wise site / natural-parameter
 * arithmetic of the CVI updates (variational_cvi_sde.py:161-174, 279-317) on packed arrays. */
int mfgm_lincomb(size_t n, double* out, double a, const double* x, double b, const double* y, double c,
                 const double* z, void* stream);

/* The damped site update of CVIGaussianProcess.update_sites, theta <- (1 - rho) theta + rho g assigned to BOTH site variables
 * (variational_cvi.py:364-368, two tf.Variable.assign), in place and in one launch: nat1[i] += w (g1[i] - nat1[i]) for i < n1,
 * nat2[i] += w (g2[i] - nat2[i]) for i < n2 (flat device arrays). */
int mfgm_site_lerp(double* nat1, const double* g1, size_t n1, double* nat2, const double* g2, size_t n2, double w, void* stream);
/* The same blend out of place: out1[i] = nat1[i] + w (g1[i] - nat1[i]), out2 likewise (out may be the input).  update_data_sites of
 * the CVI-DP model (variational_cvi_sde.py:301-317) on its two site arrays, and the same blend made one step ahead into spare buffers
 * by the pipelined loop (mfgm_cq_factor_pipelined). */
int mfgm_site_lerp_to(double* out1, const double* nat1, const double* g1, size_t n1, double* out2, const double* nat2, const double* g2,
                      size_t n2, double w, void* stream);
/* classic_elbo of the CVI-DP model on the structured state (variational_cvi_sde.py:339-352, 446-486) from the pieces the sweeps leave:
 * elbo[b] = sum_j ve_part[b, j] - (kl_part[b] + logdet[b] + c)  with ve_part [B, nblk] from mfgm_mvn_ve_compact, kl_part [B] from
 * mfgm_cq_selinv_kl, logdet [B] = log|L_q| from mfgm_cq_factor and c = -T d / 2; total = sum_b elbo[b] (either output may be NULL);
 * NaN when *info != 0 (a pivot block was not positive definite).  One launch instead of seven element-wise / reduction kernels. */
int mfgm_cq_elbo(int B, int nblk, const double* ve_part, const double* kl_part, const double* logdet, double c, const int* info,
                 double* elbo, double* total, void* stream);

/* Sparse node lists (observation times on the grid): node_ids[i] = b*T + t (int64, device), values natural
 * [n, d] / [n, d, d].  mode 0: gather packed -> values; 1: scatter values -> packed (overwrite);
 * 2: packed += scale*values, and packed2 += scale*values when packed2 != NULL.  Replaces tf.scatter_nd / tf.gather_nd at
 * variational_cvi_sde.py:167-172, 303-304 and kalman_filter.py:577. */
int mfgm_node_io(const mfgm_plan* plan, int kind, double* packed, double* packed2, const long long* node_ids, int n,
                 double* values, int mode, double scale, void* stream);
/* The same for a vector array and a symmetric array at the same nodes in one launch (the CVI site updates always move the
 * linear and the diagonal-block part together, variational_cvi_sde.py:167-172, 301-317); d <= 8. */
int mfgm_node_io_pair(const mfgm_plan* plan, double* packed_vec, double* packed_sym, const long long* node_ids, int n,
                      double* values_vec, double* values_sym, int mode, double scale, void* stream);

/* update_data_sites of the CVI models at the listed nodes (variational_cvi_sde.py:301-317) in one pass: sites <- (1 - lr) sites + lr g
 * for the linear part (sites_vec, g_vec: natural [n, d]) and the diagonal-block part (sites_sym, g_sym: natural [n, d, d], symmetric),
 * and packed_vec / packed_sym (the posterior naturals theta_lin VEC, theta_diag SYM) += new - old at those nodes.  d <= 8. */
int mfgm_site_update_pair(const mfgm_plan* plan, double* packed_vec, double* packed_sym, const long long* node_ids, int n,
                          double* sites_vec, double* sites_sym, const double* g_vec, const double* g_sym, double lr, void* stream);

/* Variational expectations of a multivariate Gaussian likelihood N(y; f, R) at the observation nodes, per trajectory
 * (multivariate_gaussian.py:80-115; variational_cvi_sde.py:319-337): sum_i -1/2 tr(Sinv Sigma_i) - 1/2 |y_i - mu_i|^2_Sinv + cst, as
 * nb = ceil(n_per / 256) partial sums per trajectory: ve [B, nb] (the caller adds them).
 * node_ids / y ([B * n_per], [B * n_per, d]) are trajectory-major; Sinv [d, d] = R^{-1} and cst = -log|chol R| - d/2 log(2 pi) on the
 * device / by value; the gathered marginals go to out_mu [B n_per, d], out_cov [B n_per, d, d] when not NULL.  d <= 8. */
int mfgm_mvn_obs_ve(const mfgm_plan* plan, const double* mu, const double* Sig, const long long* node_ids, int n_per, const double* y,
                    const double* Sinv, double cst, double* out_mu, double* out_cov, double* ve, void* stream);

/* SSM parameters -> natural parameters (cD=-0.5, cS=1; ssm_gaussian_transformations.py:182-253 `ssm_to_naturals`)
 * or precision blocks (cD=1, cS=-1; state_space_model.py:431-483 `_build_precision`), all packed:
 *   A (FULL, transition t->t+1 at node t), off (VEC: mu0 then b_k), chol (TRI: chol P0 then chol Q_k)
 *   -> lin (VEC, may be NULL together with off), diag (SYM), sub (FULL),
 *      sumlogchol[B] = sum_t log|chol_t| (may be NULL) i.e. -1/2 log det of the precision
 *      (state_space_model.py:343-373 `log_det_precision`). */
int mfgm_packed_ssm_to_naturals(const mfgm_plan* plan, const double* A, const double* off, const double* chol, double cD,
                                double cS, double* lin, double* diag, double* sub, double* sumlogchol, void* ws,
                                void* stream);

/* The per-step part of naturals_to_ssm_params (ssm_gaussian_transformations.py:459-511) in one pass over the selected inverse of
 * the precision (-2 theta_diag, -theta_sub): packed marginals Sig (SYM), Sub (FULL: Sigma_{t+1,t} at node t), mu (VEC) and the packed
 * naturals -> packed SSM parameters A (FULL at node t: t -> t+1), off (VEC: mu_0 then b_k), chol (TRI: chol P_0 then chol Q_k).
 * d <= 8; a non-positive pivot sets *info. */
int mfgm_packed_naturals_to_ssm(const mfgm_plan* plan, const double* Sig, const double* Sub, const double* mu, const double* theta_diag,
                                const double* theta_sub, double* A, double* off, double* chol, int* info, void* stream);

/* Block-tri-diagonal matrix times vector on natural-layout arrays (BlockTriDiagonal.dense_mult, block_tri_diag.py:175-199 ->
 * product_band_mat): diag [B, T, d, d], sub [B, T-1, d, d] or NULL, x / out [B, T, d] (out != x), any d.  symmetric != 0: the matrix is
 * symmetric with the lower triangles of diag given; otherwise it is lower block-bidiagonal and transpose != 0 applies its transpose. */
int mfgm_btd_matvec(int B, int T, int d, const double* diag, const double* sub, const double* x, double* out, int symmetric,
                    int transpose, void* stream);

/* L x = r (transpose == 0) or L^T x = r for a lower block-bidiagonal factor on natural-layout arrays
 * (LowerTriangularBlockTriDiagonal.solve, block_tri_diag.py:339-351 -> solve_triang_mat): Ld [B, T, d, d] (lower triangles read),
 * Ls [B, T-1, d, d] = L_{t+1,t}, r / x [B, T, d], d <= 32; scratch of mfgm_bidiag_scratch_doubles(B, T, d) doubles.  The plain
 * substitution, parallelised exactly over segments as an affine recurrence (csrc/mfgm_bidiag.h): no Gram matrix is formed. */
size_t mfgm_bidiag_scratch_doubles(int B, int T, int d);
int mfgm_bidiag_solve(int B, int T, int d, const double* Ld, const double* Ls, const double* r, double* x, int transpose, double* scratch,
                      void* stream);

/* Trace and Mahalanobis terms of KL(q || p) (state_space_model.py:557-593): q given by marginal blocks
 * Sig (SYM), Sub (FULL), mu (VEC); p by precision blocks aD*Pd (SYM), aS*Ps (FULL) and marginal means mup.
 * trace[B], maha[B]. */
int mfgm_packed_kl_terms(const mfgm_plan* plan, const double* Sig, const double* Sub, const double* mu, const double* Pd,
                         const double* Ps, double aD, double aS, const double* mup, double* trace, double* maha, void* ws,
                         void* stream);

/* Parameters of an SDE prior whose Euler map is a per-dimension cubic u_i(x) = x + dt f_i(x) = alpha_i x - beta_i x^3
 * (OrnsteinUhlenbeckSDE: alpha = 1 - dt*decay, beta = 0; DoubleWellSDE: alpha = 1 + dt*scale*c, beta = dt*scale;
 * markovflow/sde/sde.py:134-224) with diagonal diffusion q.  Symmetric matrices are packed lower triangles. */
typedef struct mfgm_sde_params {
    double alpha[8], beta[8];
    double W[8];          /* 1 / (dt q_ii) */
    double P0inv[36];     /* inverse of the prior initial covariance */
    double mu0[8];        /* prior initial mean */
    double logdetQp;      /* sum_i log(dt q_ii) */
    double logdetP0;
    double lr;            /* Girsanov-site learning rate (mode 2) */
    double clip_lo, clip_hi; /* clipping of the linearised A, b; lo >= hi disables (variational_cvi_sde.py:417-430) */
    double sq_dtq[8];     /* sqrt(dt q_ii) */
    double cholP0[36];    /* Cholesky of the prior initial covariance */
    double theta[8];      /* parameter of the non-polynomial drifts (kind >= 1) */
    double dt;            /* Euler step, u(x) = x + dt f(x) (kind >= 1) */
    int kind;             /* 0: cubic (closed-form moments); 1: f = theta tanh x (BenesSDE, sde.py:227-268); 2: f = sin(x - theta)
                           * (SineDiffusionSDE, :271-312); 3: f = sqrt(theta |x|) (SqrtDiffusionSDE, :315-356).  Kinds >= 1 use the
                           * reference's Gauss-Hermite rules (10 points for the linearisation, 20 for the KL) per dimension; d <= 4 */
    int pad_;
} mfgm_sde_params;

/* KL[q || p_SDE] of the Gaussian chain q (marginal blocks mu, Sig, Sub packed) from the Euler-discretised SDE prior,
 * in closed form (replaces the quadrature + GradientTape of SSM_KL_along_Gaussian_path / SDE_SSM_KL_with_grads_wrt_exp_params,
 * sde_utils.py:262-359, 473-547):
 *   mode 0: kl[B] only;
 *   mode 1: also d KL / d(eta_lin, eta_diag, eta_sub) written to (o1 VEC, od SYM, os FULL);
 *   mode 2: fused update_girsanov_sites (variational_cvi_sde.py:279-299): (o1,od,os) are the Girsanov sites and
 *           (q1,qd,qs) the posterior naturals; both get  -= lr * dKL/d eta  (the sparse data-site term is added by the caller);
 *   mode 3: as mode 2 but only the posterior naturals move (sites kept implicit as theta_q - theta_prior - data sites).
 * kl may be NULL in modes 1-3. */
int mfgm_packed_sde_kl(const mfgm_plan* plan, int mode, const mfgm_sde_params* prm, const double* mu, const double* Sig,
                       const double* Sub, double* kl, double* o1, double* od, double* os, double* q1, double* qd, double* qs,
                       void* ws, int* info, void* stream);

/* mfgm_packed_selinv that additionally writes the moment array mom [3d per node] = (mu_t, diag Sigma_t, diag Sigma_{t+1,t})
 * consumed by mfgm_packed_sde_lean; Sub may be NULL (then the full cross-covariance blocks are never written).
 * only_level < 0 runs every level; >= 0 launches that level's kernel alone (profiling, see mfgm_packed_selinv_level). */
int mfgm_packed_selinv_mom(const mfgm_plan* plan, int only_level, const double* L, const double* G, const double* y, double* Sig,
                           double* Sub, double* x, double* mom, void* ws, void* stream);

/* The same, for a factorisation made with G = NULL in mfgm_packed_factor (d <= 8): L_{t+1,t} = aS S L^{-T} is never stored; the
 * backward sweep rebuilds what it needs from the input sub-diagonal blocks S (H = aS S (L L^T)^{-1}).  Same bytes read, d^2 doubles
 * per node fewer written by the forward sweep.  S and aS must be those given to mfgm_packed_factor. */
int mfgm_packed_selinv_mom_s(const mfgm_plan* plan, int only_level, const double* L, const double* S, double aS, const double* y,
                             double* Sig, double* x, double* mom /* may be NULL: marginals only */, void* ws, void* stream);

/* Backward sweep of a G = NULL factorisation fused with the Girsanov-site update of CVI-DP (variational_cvi_sde.py:279-299, the
 * update mfgm_packed_sde_lean mode 3 makes from the moment array): instead of the marginals it writes
 *   (n1, nd, ns) = (1 - lr) (q1, qd, S) + lr theta~(mu, diag Sigma, diag Sigma_sub),      lr = prm->lr,
 * out of place (n1 != q1, nd != qd, ns != S).  (qd, S, q1) with scales (-2, aS = -1, 1) must be what mfgm_packed_factor was given.
 * Cubic drifts (prm->kind == 0), d <= 8, plans with at least two levels; returns 1 otherwise (callers then run
 * mfgm_packed_selinv_mom_s + mfgm_packed_sde_lean).  only_level as in mfgm_packed_selinv_mom. */
int mfgm_packed_selinv_girsanov(const mfgm_plan* plan, int only_level, const double* L, const double* S, double aS, const double* y,
                                const mfgm_sde_params* prm, const double* q1, const double* qd, double* n1, double* nd, double* ns,
                                void* ws, void* stream);

/* Backward sweep of a G = NULL factorisation that returns, next to the marginals (Sig, x), what mfgm_packed_sde_lean mode 0 would
 * compute from the moment array: kl_part [B], to which the caller adds log|L_q| - T d / 2 (variational_cvi_sde.py:446-486).  The
 * moment array is never written.  Same restrictions and fallback as mfgm_packed_selinv_girsanov. */
int mfgm_packed_selinv_kl(const mfgm_plan* plan, int only_level, const double* L, const double* S, double aS, const double* y,
                          const mfgm_sde_params* prm, double* Sig, double* x, double* kl_part, void* ws, void* stream);

/* ---- CVI-DP on the structured posterior naturals ("cq" state; csrc/mfgm_cq.h) ----------------------------------------------------
 * For a prior whose drift acts per dimension with diagonal diffusion (prm->kind == 0) and data sites that are one block shared by
 * every observation (Gaussian likelihoods: the site gradient -1/2 R^{-1} does not depend on q, variational_cvi_sde.py:204-220,
 * 301-317), the posterior naturals of CVISitesSDE (variational_cvi_sde.py:161-174) never hold more than
 *   dyn [3d per node] = (theta_lin, diag theta_diag, diag theta_sub) of  theta_prior + Girsanov sites  (data sites NOT included),
 *   one uniform off-diagonal value for theta_diag and one for theta_sub (the initial Girsanov sites, -1e-10 in the reference,
 *   :141-152, times the running product of (1 - lr)), the block p0_off at node 0 (off-diagonal part of -1/2 P0^{-1}),
 *   and the data sites given sparsely: slot[node] = observation index or -1 (packed node order: ((lane / 64) * R + step) * 64 +
 *   lane % 64), site_lin [n, d] = data-site nat1, site_sym [d(d+1)/2] = the data-site nat2 block (packed lower triangle, device).
 * The sweeps below rebuild the dense blocks in registers; results equal the dense entry points (mfgm_packed_factor with scales
 * (-2, -1, 1), mfgm_packed_selinv_girsanov, mfgm_packed_selinv_kl) on the equivalent dense naturals.  d <= 8, >= 2 levels. */
typedef struct mfgm_cq_state {
    const double* dyn;
    double d_off, s_off;
    const double* p0_off;   /* may be NULL */
    const int* slot;        /* may be NULL: no observation sites */
    const double* site_lin;
    const double* site_sym;
} mfgm_cq_state;
size_t mfgm_cq_dyn_doubles(const mfgm_plan* plan);   /* doubles of a dyn array */
size_t mfgm_cq_slot_ints(const mfgm_plan* plan);     /* ints of a slot array */
/* dense packed naturals (lin VEC, diag SYM, sub FULL) -> dyn; range [4 * Lpad] receives per-lane min / max of the off-diagonal
 * entries of diag (node 0 of every chain excluded) and of sub, so that the caller can check they are uniform. */
int mfgm_cq_pack(const mfgm_plan* plan, const double* lin, const double* diag, const double* sub, double* dyn, double* range,
                 void* stream);
/* cq state -> dense packed naturals (p0_off included, observation sites not) */
int mfgm_cq_unpack(const mfgm_plan* plan, const mfgm_cq_state* q, double* lin, double* diag, double* sub, void* stream);
/* slot array of a list of observation nodes (node_ids[i] = b*T + t); *dup becomes non-zero when two observations share a node */
int mfgm_cq_slots(const mfgm_plan* plan, const long long* node_ids, int n, int* slot, int* dup, void* stream);
/* Block Cholesky of the posterior precision + forward substitution (mfgm_packed_factor with G = NULL): L (TRI), y (VEC), logdet /
 * quad [B] (may be NULL). */
int mfgm_cq_factor(const mfgm_plan* plan, const mfgm_cq_state* q, double* L, double* y, double* logdet, double* quad, void* ws,
                   int* info, void* stream);
/* The same factorisation, pipelined across the steps of the CVI-DP loop (cvi_dp_trainer.py:72-75 iterates update_data_sites,
 * update_girsanov_sites, classic_elbo: two factorisations per step).  Under a Gaussian likelihood the data sites of the NEXT step do not
 * depend on q, so the state the first factorisation of step n + 1 will see -- dyn / offsets as they are now, sites (q_next->site_lin,
 * q_next->site_sym) -- is known while the second factorisation of step n runs.
 *   q_next (with observation sites): the level-0 reduce of q_next is made next to this call's bandwidth-bound level-0 forward sweep and
 *     its separator system goes to the second copy of the level-1 input region the workspace of such plans holds.
 *       side_stream == NULL: ONE kernel, two wavefronts per tile -- the forward sweep and the reduce of the same records (read from HBM
 *         once; the reduce's spike and Gram accumulators live in LDS so that both wavefronts fit 256 registers);
 *       side_stream != NULL: the reduce as a kernel of its own on side_stream, started once the coarse levels of this factorisation are
 *         done (the two kernels share the SIMDs: 193 + 302 registers; the records are read twice);
 *   use_ahead != 0: the record the previous pipelined call on this plan made was for EXACTLY the state q (the caller's
 *     responsibility): the level-0 reduce of this call is skipped, `stream` waits for the work queued on side_stream (if given), and the
 *     coarse levels read the record in place (nothing is copied: the two copies of the region swap roles).
 * Results are those of mfgm_cq_factor (the record holds the same numbers the level-0 reduce would write). */
int mfgm_cq_factor_pipelined(const mfgm_plan* plan, const mfgm_cq_state* q, double* L, double* y, double* logdet, double* quad, void* ws,
                             int* info, int use_ahead, const mfgm_cq_state* q_next, void* side_stream, void* stream);
/* one part of it alone (profiling): stage 0 the level-0 reduce, 1 the level-0 forward (as mfgm_packed_factor_stage), 2 the levels above
 * the finest one (their reduces, the fused coarse kernel, their forwards) on whatever the last stage 0 left in the workspace */
int mfgm_cq_factor_stage(const mfgm_plan* plan, int stage, const mfgm_cq_state* q, double* L, double* y, void* ws, int* info,
                         void* stream);
/* backward sweep fused with update_girsanov_sites (variational_cvi_sde.py:279-299): dyn_out = (1 - lr) dyn + lr theta~ (out of place;
 * the caller scales d_off / s_off by (1 - lr)).  only_level 0: the level-0 kernel alone (+ its fix-up), < 0: every level. */
int mfgm_cq_selinv_girsanov(const mfgm_plan* plan, int only_level, const mfgm_cq_state* q, const double* L, const double* y,
                            const mfgm_sde_params* prm, double* dyn_out, void* ws, void* stream);
/* backward sweep with the KL sum (mfgm_packed_selinv_kl): marginals Sig (SYM), x (VEC) -- both NULL: not written, the ELBO needs only
 * the sums and the observation nodes --, kl_part [B]; obs_mu [n, d] / obs_cov [n, d, d] (both or neither) receive the marginals at
 * the observation nodes, in observation order.  only_level 0: the level-0 kernel alone, > 0: the levels above it alone (profiling),
 * < 0: everything. */
int mfgm_cq_selinv_kl(const mfgm_plan* plan, int only_level, const mfgm_cq_state* q, const double* L, const double* y,
                      const mfgm_sde_params* prm, double* Sig, double* x, double* kl_part, double* obs_mu, double* obs_cov, void* ws,
                      void* stream);
/* mfgm_mvn_obs_ve on marginals already gathered in observation order (mu [B n_per, d], cov [B n_per, d, d]): ve [B, ceil(n_per/256)] */
int mfgm_mvn_ve_compact(int B, int n_per, int d, const double* mu, const double* cov, const double* y, const double* Sinv, double cst,
                        double* ve, void* stream);

/* ---- Tensor-product Gauss-Hermite kernels: drifts that couple the state dimensions / have no polynomial form, full diffusion matrix ----
 * The reference's own formulation of the local CVI-DP / VDP quantities (gpflow.quadrature.mvnquad: nodes m + sqrt(2) L xi, 10 points per
 * dimension in the linearisation, 20 in the KL / E_sde), with the GradientTape's chain rule written out (the rule is differentiated as a
 * formula).  Arrays in the reference's NATURAL layout, device pointers; d <= 3; small models (20^d nodes per time step).
 *   kind 10  Van der Pol (markovflow/sde/sde.py:432-518), d = 2: theta = (a, tau)
 *   kind 11  ReLU network 1 -> nh -> 1 on every state dimension (sde.py:359-429): theta = (W1 [nh], b1 [nh], W2 [nh], b2), nh <= 13
 *   kind 12  per-dimension cubic c1 x - c3 x^3 (Ornstein-Uhlenbeck / double well, sde.py:134-224) with a NON-DIAGONAL diffusion
 *            matrix: theta = (c1, c3)
 *   kind 13 / 14 / 15  theta tanh x / sin(x - theta) / sqrt(theta |x|) per dimension (sde.py:227-356): theta = (theta, unused) -- the VDP
 *            model and prior learning with these drifts (the closed-form-moment kernels serve their CVI-DP inference) */
#define MFGM_QUAD_NTHETA 40
typedef struct mfgm_quad_drift {
    int kind, d, nh, pad_;
    double theta[MFGM_QUAD_NTHETA];
    double dt;
    double W[6];           /* (dt q)^{-1}, packed lower triangle */
    double logdetQp;       /* log det (dt q) */
    double mu0[3];         /* p(x0) */
    double P0inv[6];       /* packed lower triangle */
    double logdetP0;
    double clip_lo, clip_hi;   /* clipping of the linearised A, b (variational_cvi_sde.py:420-432); lo >= hi: none */
} mfgm_quad_drift;
/* A [N, d, d] = I + dt E_q[df/dx], b [N, d] = dt (E_q f - E_q[df/dx] m) on N(mean[n], cov[n]) (sde_utils.py:119-179, drift.py:66-117);
 * *info becomes non-zero when a covariance is not positive definite */
int mfgm_quad_linearize(const mfgm_quad_drift* drift, int N, const double* mean, const double* cov, double* A, double* b, int* info,
                        void* stream);
/* KL[q || p_SDE] of B chains along their Gaussian paths (sde_utils.py:262-359): mu [B, T, d], Sig [B, T, d, d], Sub [B, T-1, d, d] =
 * Cov(x_{t+1}, x_t) -> kl [B]; with g1 [B, T, d], gd [B, T, d, d], gs [B, T-1, d, d] (all or none) its gradient with respect to the
 * expectation parameters (sde_utils.py:473-547), and with gtheta [B, np] (np = 2, or 3 nh + 1 for kind 11; may be NULL) the gradient with
 * respect to the drift parameters (variational_cvi_sde.py:495-506).  scratch: mfgm_quad_kl_scratch_doubles(B, T, d, nh) doubles. */
size_t mfgm_quad_kl_scratch_doubles(int B, int T, int d, int nh);
int mfgm_quad_kl(const mfgm_quad_drift* drift, int B, int T, const double* mu, const double* Sig, const double* Sub, double* kl,
                 double* g1, double* gd, double* gs, double* gtheta, double* scratch, int* info, void* stream);
/* VDP: E_sde per node without the Riemann factor dt, E [N] = 1/2 E_{N(mean, cov)} |f(x) + A x - b|^2_{q^-1} (the variational drift is
 * -A x + b, vi_sde.py:422-434, sde_utils.py:182-259), and its gradients with respect to (m, S) (vi_sde.py:205-243), (A, b) and the drift
 * parameters (vi_sde.py:457-470); every output but E may be NULL. */
int mfgm_quad_esde(const mfgm_quad_drift* drift, int N, const double* mean, const double* cov, const double* A, const double* b, double* E,
                   double* dEdm, double* dEdS, double* dEdA, double* dEdb, double* gtheta, int* info, void* stream);

/* VDP Lagrange sweep with jump conditions (vi_sde.py:289-347) on natural-layout arrays, d <= 3: A, dEdS, psi [B, N, d, d], dEdm, lam
 * [B, N, d], dobsm [B, N + 1, d], dobsS [B, N + 1, d, d] (the likelihood's gradients scattered on the grid); clip > 0: stabilize_system
 * (NaN -> 1e-8, clipping to [-clip, clip] of the four gradient arrays).  The reference's loop, sequential per chain. */
int mfgm_quad_vdp_lagrange(int B, int N, int d, double dt, double clip, const double* A, const double* dEdm, const double* dEdS,
                           const double* dobsm, const double* dobsS, double* psi, double* lam, void* stream);

/* ---- Kalman filter with Gaussian sites and a time-invariant emission matrix (kalman_filter.py:86-107, 184-271, 417-500) ----------------
 * The path of KalmanFilterWithSites.log_likelihood / CVIGaussianProcess.elbo and of predict_f at the data points
 * (variational_cvi.py:106-135, 351-379): sites nat1 [Bs, T, o], nat2 [Bs, T, o, o] in natural layout (Bs = 1: shared by all chains,
 * or B), H [o, d] row-major, Hmu = H mu_prior [B, T, o] (NULL: zero-mean prior).  o <= 2, d <= 8.  One call = assembly of the
 * posterior precision + right-hand side in the packed layout, the sweeps, and the reductions / projections around them. */
typedef struct mfgm_kf_sites {
    double H[32];
    int o;
    int site_batch;
    const double* nat1;
    const double* nat2;
    const double* Hmu;
} mfgm_kf_sites;
/* log-likelihood terms per chain: t1 = sum_t (y_t - H mu_p)^T R_t^{-1} (y_t - H mu_p), ldR = sum_t log det R_t^{-1}, logdet = log|L| of
 * the posterior precision, quad = |L^{-1} H^T R^{-1} (y - H mu_p)|^2 (kalman_filter.py:229-251; R^{-1} = -2 nat2, y = site means).
 * Pd (SYM) / Ps (FULL): packed prior precision blocks; D, r, L, y: packed scratch (SYM, VEC, TRI, VEC). */
int mfgm_kf_sites_loglik(const mfgm_plan* plan, const mfgm_kf_sites* sites, const double* Pd, const double* Ps, double* D, double* r,
                         double* L, double* y, double* t1, double* ldR, double* logdet, double* quad, void* ws, int* info, void* stream);
/* mfgm_kf_sites_loglik with the scalar assembly inside (kalman_filter.py:229-255): ll[b] = cst + term1 + term2 + term3 with
 * term3 = -sumlogchol[b] - log|L| + 1/2 ldR (sumlogchol [B]: sum log diag chol of the prior's P0, Q_k; cst: the caller's
 * -1/2 o T log 2 pi), total = sum_b ll[b]; NaN when a pivot block was not positive definite (info).  terms [4, B] (may be NULL) receives
 * (t1, ldR, log|L|, |y|^2).  The four per-chain sums take two launches instead of five. */
int mfgm_kf_sites_elbo(const mfgm_plan* plan, const mfgm_kf_sites* sites, const double* Pd, const double* Ps, double* D, double* r,
                       double* L, double* y, const double* sumlogchol, double cst, double* terms, double* ll, double* total, void* ws,
                       int* info, void* stream);
/* posterior marginals (Sig SYM, x VEC packed) of  precision = prior precision + H^T R^{-1} H,  rhs = plin + H^T nat1  (plin = packed
 * K^{-1} mu_prior or NULL), and their projections Fmu = H x, Fvar = diag(H Sig H^T), natural [B, T, o]. */
int mfgm_kf_sites_predict(const mfgm_plan* plan, const mfgm_kf_sites* sites, const double* Pd, const double* Ps, const double* plin,
                          double* D, double* r, double* L, double* y, double* Sig, double* x, double* Fmu, double* Fvar, void* ws,
                          int* info, void* stream);

/* The same marginals and projections from a factorisation that is already in place: (L, y) as mfgm_kf_sites_loglik /
 * mfgm_kf_sites_predict left them for THESE sites (and the coarse levels of that factorisation still in ws), Ps the prior's sub-diagonal
 * precision blocks.  With a zero-mean prior the log-likelihood and the prediction factorise the same system with the same right-hand
 * side, H^T nat1 (kalman_filter.py:184-255 vs posterior.py:207-260 at the data): CVIGaussianProcess.update_sites after elbo()
 * (variational_cvi.py:351-379) needs the selected inverse and the projection only. */
int mfgm_kf_sites_predict_factored(const mfgm_plan* plan, const mfgm_kf_sites* sites, const double* Ps, const double* L, const double* y,
                                   double* Sig, double* x, double* Fmu, double* Fvar, void* ws, void* stream);

/* ---- sparse / inducing-state CVI (markovflow/models/sparse_variational_cvi.py; csrc/mfgm_sparse.h) ----------------------------------------
 * One chain, M inducing states, N data points sorted in time; natural-layout arrays, d <= 32.  Interval m = 0..M lies between inducing
 * states m-1 and m (the prior pads both ends); seg [M+2] are the CSR offsets of the data points per interval; w [N, 2d] = H P_i is the
 * projection of data point i onto the pair of states around it and c [N] = H T_i H^T its conditional variance (conditionals.py:207-256;
 * functions of the time points and the kernel only).  All pointers are device pointers. */
typedef struct mfgm_sparse_data {
    int M, d, N;
    const int* seg;
    const double* w;
    const double* c;
    const double* prior_mean;   /* [d]    kernel.initial_mean            (sde_kernel.py:402-419) */
    const double* prior_cov;    /* [d, d] kernel.initial_covariance_matrix */
    /* One chain shared between processes (mfgm_plan_set_shard_level): the intervals [m_lo, m_hi) this process owns -- interval m belongs
     * to the owner of inducing state m, the last process also takes interval M.  seg then has m_hi - m_lo + 1 entries and seg, w, c, fmu,
     * fvar, g1, g2 hold the N data points of the owned intervals only; the arrays over inducing states keep global indices.
     * m_hi <= 0: every interval (0, M + 1). */
    int m_lo, m_hi;
} mfgm_sparse_data;
/* posterior naturals = prior naturals (plin [T, d] or NULL, pdiag / psub [T, d, d]) + the sites nat1 [M+1, 2d], nat2 [M+1, 2d, 2d]
 * overlap-added into the block-tri-diagonal structure (sparse_variational_cvi.py:140-174); T = M. */
int mfgm_sparse_theta(int T, int d, const double* nat1, const double* nat2, const double* plin, const double* pdiag, const double* psub,
                      double* lin, double* diag, double* sub, void* stream);
/* Factorisation of the sparse-CVI posterior straight from the sites (plans with 8 < d <= 32, one chain, T = M inducing states):
 * mfgm_sparse_theta followed by mfgm_packed_factor_form(form 1, aD = -2, aS = -1, aR = 1) without materialising the posterior
 * naturals -- the level-0 passes form  theta_t = prior_t + overlap-added sites  while loading (csrc/mfgm_mfma_inv.h, site_diag /
 * site_sub / site_lin).  nat1 [M + 1, 2d], nat2 [M + 1, 2d, 2d]; plin (may be NULL) [M, d], pdiag / psub [M, d, d]: the prior's
 * naturals.  L, G, y: inverse-form factor arrays for mfgm_packed_selinv_form(form 1); logdet / quad [1] may be NULL. */
int mfgm_sparse_factor(const mfgm_plan* plan, const double* nat1, const double* nat2, const double* plin, const double* pdiag,
                       const double* psub, double* L, double* G, double* y, double* logdet, double* quad, void* ws, int* info,
                       void* stream);
/* The same factorisation of ONE chain shared between processes (mfgm_plan_set_shard_level; sparse_variational_cvi.py:140-174 with the
 * time axis cut at separators of a coarse level): phase 0 eliminates the interiors of the owned segments below the exchange level, the
 * caller sums the region mfgm_plan_exchange_region names over the processes, phase 1 solves the replicated upper levels and walks back
 * down the owned segments.  A process reads the sites node_lo .. node_hi of its node range [node_lo, node_hi) (site node_hi belongs to its
 * right neighbour: one [2d + 4d^2] halo per step) and the prior naturals of its own nodes and of node_lo - 1 (psub).  logdet / quad are
 * the partial sums over the owned nodes. */
int mfgm_sparse_factor_phase(const mfgm_plan* plan, int phase, const double* nat1, const double* nat2, const double* plin,
                             const double* pdiag, const double* psub, double* L, double* G, double* y, double* logdet, double* quad, void* ws,
                             int* info, void* stream);
/* The QUADRANT-PACKED site tensor nat2q [M + 1, QS], QS = d (d + 1) + d^2: per site the upper-left block of the symmetric [2d, 2d]
 * matrix as a packed lower triangle (row-major, ET = d (d + 1) / 2 entries), the lower-left block in full (d x d, rows = second state
 * of the pair), the lower-right block as a packed lower triangle; the upper-right block is the transpose of the lower-left one and is
 * not stored.  The resident form of SparseCVIGaussianProcess's sites on the wide path (the reference's `sites.nat2`,
 * sparse_variational_cvi.py:96-110, is materialised from it on demand): 528 instead of 1 024 doubles per site at d = 16 for the
 * site update and for the two factor passes that read the sites.
 *   mfgm_sparse_factor_q       mfgm_sparse_factor (phase -1) / mfgm_sparse_factor_phase (phase 0, 1) on nat2q
 *   mfgm_sparse_site_update_q  mfgm_sparse_site_update on nat2q
 *   mfgm_wide_stage_q          one level-0 pass alone (which = 0 reduce, 1 forward; inverse form), for profiling */
int mfgm_sparse_factor_q(const mfgm_plan* plan, int phase, const double* nat1, const double* nat2q, const double* plin, const double* pdiag,
                         const double* psub, double* L, double* G, double* y, double* logdet, double* quad, void* ws, int* info,
                         void* stream);
int mfgm_sparse_site_update_q(const mfgm_sparse_data* data, const double* g1, const double* g2, double lr, double* nat1, double* nat2q,
                              void* stream);
int mfgm_wide_stage_q(const mfgm_plan* plan, int which, const double* D, const double* S, const double* r, double aD, double aS, double aR,
                      double* L, double* G, double* y, const double* site1, const double* site2q, void* ws, int* info, void* stream);
/* After mfgm_packed_selinv_form on a shared chain: the marginal (Sig [T, d, d], x [T, d] or NULL) of the separator on the left of the
 * owned node range, node_lo - 1, copied from the replicated exchange level -- the pair marginal of the first owned interval needs it
 * (conditionals.py:380-421 with the left conditioning state owned by the neighbour).  No-op on the first process. */
int mfgm_plan_shard_left_marginal(const mfgm_plan* plan, double* Sig, double* x, const void* ws, void* stream);
/* q(f(t_i)) at the data points (posterior.py:207-260 through conditionals.py:380-470) from the posterior marginals of the inducing
 * states: mu [M, d], Sig [M, d, d], Sub [M, d, d] (Sigma_{t+1,t} at t); fmu, fvar [N]. */
int mfgm_sparse_predict(const mfgm_sparse_data* data, const double* mu, const double* Sig, const double* Sub, double* fmu, double* fvar,
                        void* stream);
/* the same pass with the trace and Mahalanobis terms of KL[q || p] (mfgm_packed_kl_terms: state_space_model.py:528-593) taken from the
 * pair covariances it holds anyway; plan: the wide plan (8 < d <= 32, one chain, T = M) whose workspace ws receives the partial sums;
 * Pd / Ps [M, d, d]: the prior's precision blocks times aD / aS; mup [M, d]: its marginal means; trace / maha [1]. */
int mfgm_sparse_predict_kl(const mfgm_sparse_data* data, const double* mu, const double* Sig, const double* Sub, double* fmu, double* fvar,
                           const mfgm_plan* plan, const double* Pd, const double* Ps, double aD, double aS, const double* mup,
                           double* trace, double* maha, void* ws, void* stream);
/* ConditionalProcess.predict_state (posterior.py:207-229 -> conditional_predict / base_conditional_predict, conditionals.py:29-76,
 * 380-421) at N query points: idx [N] = interval of each point (0 .. M, as above), P [N, d, 2d] / T [N, d, d] its conditional
 * statistics (conditionals.py:207-256), the marginals of the M conditioning states as in mfgm_sparse_predict;
 * out_mean [N, d] = P m_pair, out_cov [N, d, d] = T + P S_pair P^T (no [M+1, 2d, 2d] pairwise tensor, no per-point gather). */
int mfgm_cond_predict(int M, int d, int N, const int* idx, const double* P, const double* T, const double* prior_mean,
                      const double* prior_cov, const double* mu, const double* Sig, const double* Sub, double* out_mean, double* out_cov,
                      void* stream);
/* update_sites (sparse_variational_cvi.py:176-221): sites <- (1 - lr) sites + lr sum_{i in interval} (g1_i w_i, g2_i w_i w_i^T), in place;
 * g1, g2 [N] are the likelihood gradients with respect to the expectation parameters of f(t_i). */
int mfgm_sparse_site_update(const mfgm_sparse_data* data, const double* g1, const double* g2, double lr, double* nat1, double* nat2,
                            void* stream);

/* CVI-DP on the moment array: KL[q||p] = -H[q] - E_q[log p] where E_q[log p] of a per-dimension cubic drift with diagonal
 * diffusion depends on q only through mom, so d KL / d eta = theta_q - theta~(mom) with explicit "effective prior naturals"
 * theta~ (csrc/mfgm_sde.h) and no d x d factorisation:
 *   mode 0: kl_part[B] = sum_t 1/2 [ sum_i W_i T_i + logdet Qp ] + the x0 term; KL = kl_part + log|L_q| - T d / 2
 *           (Sig: packed marginal covariances, read at node 0 only);
 *   mode 3: update_girsanov_sites (variational_cvi_sde.py:279-299) as theta_q <- (1 - lr) theta_q + lr theta~
 *           (the sparse data-site term lr * scatter(data) is added by the caller). */
int mfgm_packed_sde_lean(const mfgm_plan* plan, int mode, const mfgm_sde_params* prm, const double* mom, const double* Sig,
                         double* kl_part, double* q1, double* qd, double* qs, void* ws, void* stream);

/* Linearise the SDE on the posterior path (set_linearized_prior, variational_cvi_sde.py:408-432; linearize_sde,
 * sde_utils.py:119-179; LinearDrift.to_ssm, drift.py:66-117): packed SSM parameters A (FULL), off (VEC), chol (TRI). */
int mfgm_packed_linearize_cubic(const mfgm_plan* plan, const mfgm_sde_params* prm, const double* mu, const double* Sig,
                                double* A, double* off, double* chol, void* stream);

/* A sum of up to 8 stationary SDE-kernel components (state dims add up to d <= 8): order 1 = Matern-1/2 or
 * Ornstein-Uhlenbeck (lam = 1/lengthscale or decay; var = variance or diffusion/(2 decay)), order 2 = Matern-3/2
 * (lam = sqrt(3)/l), order 3 = Matern-5/2 (lam = sqrt(5)/l).  kernels/matern.py:27-520, kernels/sde_kernel.py:540-687. */
typedef struct mfgm_kernel_spec {
    int ncomp;
    int order[8];
    int offset[8];
    double lam[8];
    double var[8];
    double mean[8];
    double jitter;
} mfgm_kernel_spec;

/* Kernel -> packed SSM parameters on a time grid: A_k = expm(F dt_k), Q_k = Pinf - A_k Pinf A_k^T + jitter (Cholesky
 * factored, zero matrices kept zero), b_k = (I - A_k) m  (SDEKernel.state_space_model, kernels/sde_kernel.py:153-171;
 * StationaryKernel.transition_statistics :421-446; state_space_model_from_covariances, state_space_model.py:613-664).
 * time_deltas: natural [B, T-1] device array.  Outputs packed A (FULL), off (VEC), chol (TRI). */
int mfgm_packed_stationary_ssm(const mfgm_plan* plan, const mfgm_kernel_spec* spec, const double* time_deltas, double* A,
                               double* off, double* chol, int* info, void* stream);

/* VDP (markovflow/models/vi_sde.py `VariationalMarkovGP`): drift f_i(x) = af_i x - bf_i x^3, diagonal diffusion q,
 * q(x0) = N(mu0, chol0 chol0^T) (packed lower triangle), grid step dt, learning rate lr. */
typedef struct mfgm_vdp_params {
    double af[8], bf[8];
    double q[8];
    double mu0[8];
    double chol0[36];
    double dt;
    double lr;
    double clip;          /* > 0: stabilize_system: NaN -> 1e-8 and clipping to [-clip, clip] of dE/dm, dE/dS and the jump conditions
                           * inside the Lagrange sweep (vi_sde.py:312-323; CLIP_MAX = 5000 in the reference) and of psi / lambda, in
                           * place, in update_param (vi_sde.py:393-397); the SSM's state transitions and offsets are clipped to
                           * [-1, 1] in vdp_to_ssm (vi_sde.py:186-200).  0: off */
} mfgm_vdp_params;

/* forward_pass (vi_sde.py:171-204; LinearDrift(-A, b).to_ssm, drift.py:66-117): variational parameters Am (FULL), bm (VEC),
 * stored at node t for the transition t -> t+1, to packed SSM parameters A, off, chol. */
int mfgm_packed_vdp_to_ssm(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* Am, const double* bm, double* A,
                           double* off, double* chol, void* stream);
/* E_sde / dt per chain (vi_sde.py:422-434; squared_drift_difference_along_Gaussian_path, sde_utils.py:182-249) in closed
 * form; gm (VEC) / gS (SYM) receive dE/dm / dt and dE/dS / dt (vi_sde.py:206-239) when non-NULL. */
int mfgm_packed_vdp_esde(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig,
                         const double* Am, const double* bm, double* e_over_dt, double* gm, double* gS, void* ws, void* stream);
/* forward_pass without the intermediate SSM arrays: the precision blocks (diag SYM, sub FULL) and linear term (lin VEC) of the Euler
 * chain of the drift (-A, b) with per-trajectory q(x0): p0inv [B][d(d+1)/2] = P0^{-1} (packed lower triangles), p0lin [B][d] =
 * P0^{-1} mu0.  Equals mfgm_packed_vdp_to_ssm + node-0 overwrite + mfgm_packed_ssm_to_naturals(cD = 1, cS = -1); factor the result
 * with scales (1, 1, 1). */
int mfgm_packed_vdp_to_naturals(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* Am, const double* bm,
                                const double* p0inv, const double* p0lin, double* lin, double* diag, double* sub, void* stream);

/* X_t = Phi_t X_{t-1} Phi_t^T + Q_t for t = 0 .. T-1 with X_{-1} = 0, on every chain of the plan (d <= 8): Phi FULL, Q and X SYM packed
 * arrays over all T nodes.  This is the recurrence behind the exact derivative of the marginals with respect to the natural parameters,
 * dSigma = -Sigma dP Sigma restricted to the band -- what the reference's GradientTape returns through banded_matrices' registered
 * gradients of cholesky_band / inverse_from_cholesky_band (ssm_natgrad.py:154-201 calls tape.gradient on naturals_to_ssm_params) --
 * partitioned over the plan's segments like the VDP moment recursion (three passes, no factorisation).
 * seg: scratch of mfgm_congruence_scan_workspace_doubles(plan) doubles. */
size_t mfgm_congruence_scan_workspace_doubles(const mfgm_plan* plan);
int mfgm_congruence_scan(const mfgm_plan* plan, const double* Phi, const double* Q, double* X, double* seg, void* stream);

/* The band of X = Sigma dP Sigma -- X_tt (Xd, SYM) and X_{t+1,t} (Xs, FULL) -- from the band of the covariance of a Gauss-Markov chain
 * (Sig SYM = Sigma_tt, Sub FULL = Sigma_{t+1,t}, what mfgm_packed_selinv returns) and a symmetric block-tri-diagonal dP (dPd SYM lower
 * triangles, dPs FULL = dP_{t+1,t}): d Sigma = -Sigma dP Sigma is the covariance half of the derivative of the marginals with respect
 * to the natural parameters, which the reference takes from a GradientTape through the banded ops (ssm_natgrad.py:154-201).  Exact, no
 * re-factorisation: one d x d Cholesky per node, two congruence recurrences (mfgm_congruence_scan), two local passes (csrc/mfgm_band.h).
 * d <= 8.  work: scratch of mfgm_band_workspace_doubles(plan) doubles. */
size_t mfgm_band_workspace_doubles(const mfgm_plan* plan);
int mfgm_band_sigma_dP_sigma(const mfgm_plan* plan, const double* Sig, const double* Sub, const double* dPd, const double* dPs, double* Xd,
                             double* Xs, double* work, void* stream);

/* The same band for block sizes up to 32 on NATURAL-layout arrays (no plan): Sig [B, T, d, d] = Sigma_tt, Sub [B, T-1, d, d] =
 * Sigma_{t+1,t}, dPd [B, T, d, d] symmetric (both triangles read), dPs [B, T-1, d, d] = dP_{t+1,t}; Xd [B, T, d, d], Xs [B, T-1, d, d].
 * This is the native backward of the natural-gradient tape where the reference differentiates naturals_to_ssm_params through the
 * banded ops with d up to 30 (ssm_natgrad.py:142-201; tests/integration/test_ssm_natgrad.py).  One wavefront per node / per segment,
 * every block product a Gram product of v_mfma_f64_16x16x4_f64 tiles, Sigma_t^-1 by 4 x 4-pivot block sweeps, each recurrence in three
 * passes over ~sqrt(T / 2.5)-node segments, both recurrences in the same launches (csrc/mfgm_wband.h): five launches whatever T.  T >= 2, d <= 32.
 * work: scratch of mfgm_wband_workspace_doubles(B, T, d) doubles; info: device word, non-zero when a Sigma_tt is not positive definite. */
size_t mfgm_wband_workspace_doubles(int B, int T, int d);
int mfgm_wband_sigma_dP_sigma(int B, int T, int d, const double* Sig, const double* Sub, const double* dPd, const double* dPs, double* Xd,
                              double* Xs, double* work, int* info, void* stream);

/* forward_pass as the moment recursion of the reference (vi_sde.py:171-204), partitioned over the segments of the plan: marginal
 * means mu (VEC) and covariances Sig (SYM) of the Euler chain of the drift (-A, b) started at q(x0) = N(q0_mu[b], q0_cov[b])
 * (q0_mu [B][d], q0_cov [B][d(d+1)/2] packed lower triangles).  No factorisation: 42 doubles read twice and 27 written per node.
 * e_over_dt (optional, [B]): E_sde / dt of these marginals under (Am, bm), what mfgm_packed_vdp_esde would return, accumulated by
 * the final sweep while the blocks are in registers (ws: the plan workspace, needed then).
 * seg: scratch of mfgm_vdp_workspace_doubles(plan) doubles. */
int mfgm_packed_vdp_marginals(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* Am, const double* bm,
                              const double* q0_mu, const double* q0_cov, double* mu, double* Sig, double* e_over_dt, double* seg,
                              void* ws, void* stream);
/* mfgm_packed_vdp_marginals that also prepares the Lagrange sweep of the same iteration (vi_markov_gp_trainer.py:55-57: forward_pass,
 * update_lagrange, update_param on one (A, b)), leaving the results in lagrange_seg -- a second array of
 * mfgm_vdp_workspace_doubles(plan) doubles, distinct from seg, to be handed to mfgm_packed_vdp_lagrange_update0 as its seg:
 *   yR == NULL: the first pass, which has every A_t in registers and is bound by reading them, forms the linear parts of that sweep's
 *     segment maps on the side (accumulators in LDS); the Lagrange call runs with mode = 2.  Saves one pass over A (8 d^2 bytes / node).
 *   yR != NULL (with dobsS or obs_count / dobs_const, as in the Lagrange calls; prm->clip as there): the final sweep, which produces
 *     (m_t, S_t) with (A_t, b_t) in registers, accumulates the affine offsets of the segment maps as well (the descending recurrence's
 *     segment map summed in ascending order, csrc/mfgm_vdp.h); the Lagrange call runs with mode = 3 and starts at its segment scan.
 * What rides in the sweeps depends on d (registers and LDS): everything for d <= 5, the products for d = 6 (the offsets pass follows the
 * final sweep as a launch of this call), neither for d = 7, 8 (both passes follow).  The contract is the same. */
int mfgm_packed_vdp_marginals_products(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* Am, const double* bm,
                                       const double* q0_mu, const double* q0_cov, double* mu, double* Sig, double* e_over_dt, double* seg,
                                       double* lagrange_seg, const double* yR, const double* dobsS, const int* obs_count,
                                       const double* dobs_const, void* ws, void* stream);

/* update_lagrange (vi_sde.py:289-347): psi (FULL) and lambda (VEC) on nodes 0..T-2.  yR (VEC) = R^{-1} y and dobsS (SYM) =
 * -1/2 R^{-1} at the observation nodes, zero elsewhere (jump conditions of a Gaussian likelihood, vi_sde.py:262-287).
 * seg: scratch of mfgm_vdp_workspace_doubles(plan) doubles. */
size_t mfgm_vdp_workspace_doubles(const mfgm_plan* plan);
int mfgm_packed_vdp_lagrange(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig,
                             const double* Am, const double* bm, const double* yR, const double* dobsS, double* psi,
                             double* lam, double* seg, const int* obs_count, const double* dobs_const, void* stream);
/* obs_count / dobs_const (both or neither): when every observation contributes the same block (one Gaussian likelihood), dobsS =
 * obs_count[node] * dobs_const with obs_count one int per node in the packed order [tile][step][64 lanes] (index
 * ((lane / 64) * R + step) * 64 + lane % 64, R and the lane of a node as in mfgm_plan_describe) and dobs_const [d(d+1)/2] on the
 * device; dobsS is then not read and may be NULL (d(d+1)/2 doubles per node that are zero almost everywhere). */
/* mfgm_packed_vdp_lagrange immediately followed by mfgm_packed_vdp_update_param on the same (mu, Sig), as the trainer calls them
 * (vi_markov_gp_trainer.py:56-57), in the same three passes: the final sweep replaces (Am, bm) node by node as soon as psi_t and
 * lambda_t are known.  psi / lam end up as update_param leaves them (clipped when prm->clip > 0). */
int mfgm_packed_vdp_lagrange_update(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig, double* Am,
                                    double* bm, const double* yR, const double* dobsS, double* psi, double* lam, double* seg,
                                    const int* obs_count, const double* dobs_const, void* stream);
/* The same with the multipliers kept at NODE 0 only: psi0 [B, d, d], lam0 [B, d] (natural layout) instead of the packed arrays.  In the
 * trainer's loop (vi_markov_gp_trainer.py:56-58: update_lagrange, update_param, update_initial_statistics) every other multiplier has
 * been consumed by the fused parameter update when the sweep leaves its node; only psi(0), lambda(0) are read afterwards
 * (vi_sde.py:241-260).  Saves the d^2 + d stores per node nobody loads (42 of 153 doubles at d = 6).  mode 0: all passes; 1: the last
 * kernel alone (roofline timing; seg must hold the segment scans of a full call); 2: all passes but the first one, the products of
 * (I - 2 dt A_t) / (I - dt A_t) over each segment, which a preceding mfgm_packed_vdp_marginals_products on the SAME (Am, bm) has left
 * in seg; 3: from the segment scan on -- that call was given the jump terms and has left the offsets of the segment maps too (then
 * (mu, Sig) must be the marginals it produced). */
int mfgm_packed_vdp_lagrange_update0(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig, double* Am,
                                     double* bm, const double* yR, const double* dobsS, double* psi0, double* lam0, double* seg,
                                     const int* obs_count, const double* dobs_const, int mode, void* stream);
/* Profiling / roofline entry point: the LAST kernel of mfgm_packed_vdp_lagrange_update alone (the final sweep that also replaces
 * (Am, bm)); seg must hold the segment scans of a full call with the same arguments. */
int mfgm_packed_vdp_lagrange_update_final(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig,
                                          double* Am, double* bm, const double* yR, const double* dobsS, double* psi, double* lam,
                                          double* seg, const int* obs_count, const double* dobs_const, void* stream);
/* update_param (vi_sde.py:377-414): A <- (1-lr) A + lr (-E f' + 2 q psi), b <- (1-lr) b + lr (E f + A~ m - q lambda).
 * With prm->clip > 0 psi and lam are overwritten by their clipped values (they are not const then). */
int mfgm_packed_vdp_update_param(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig,
                                 const double* psi, const double* lam, double* Am, double* bm, void* stream);

/* ---- natural-layout convenience entry points (what a TF custom-op kernel for the reference would call) -------------------
 * Inputs / outputs are the reference's row-major tensors: diag [B,T,d,d] (lower triangles read), sub [B,T-1,d,d],
 * rhs [B,T,d].  `nws` is scratch of mfgm_natural_workspace_bytes(plan) bytes (packed temporaries + the plan workspace).
 *   mfgm_btd_cholesky : L_diag [B,T,d,d] (upper triangles zero), L_sub [B,T-1,d,d] (may be NULL with T == 1), logdet[B]
 *                       = SymmetricBlockTriDiagonal.cholesky + abs_log_det (block_tri_diag.py:428-440, 353-366)
 *   mfgm_btd_posterior: logdet[B], x = K^{-1} rhs [B,T,d] (rhs / x may be NULL), Sdiag [B,T,d,d], Ssub [B,T-1,d,d] (may be NULL)
 *                       = cholesky + solve + block_diagonal_of_inverse fused (block_tri_diag.py:318-351; the marginals route of
 *                       state_space_model.py:232-262 and naturals_to_ssm_params, ssm_gaussian_transformations.py:440-458).
 * aD, aS, aR scale the inputs on load (-2, -1, 1 turn natural parameters into a precision). */
size_t mfgm_natural_workspace_bytes(const mfgm_plan* plan);
int mfgm_btd_cholesky(const mfgm_plan* plan, const double* diag, const double* sub, double aD, double aS, double* Ldiag,
                      double* Lsub, double* logdet, void* nws, int* info, void* stream);
int mfgm_btd_posterior(const mfgm_plan* plan, const double* diag, const double* sub, const double* rhs, double aD, double aS,
                       double aR, double* logdet, double* x, double* Sdiag, double* Ssub, void* nws, int* info, void* stream);

/* Profiling / roofline entry points: launch exactly ONE kernel of a sweep (stage 0 = reduce, 1 = forward; level 0 =
 * finest).  The coarser levels must already be in `ws` from a full mfgm_packed_factor / mfgm_packed_selinv call with the
 * same arguments; outputs are overwritten with identical values.  Used by bench.py to time the dominant kernel alone. */
int mfgm_packed_factor_stage(const mfgm_plan* plan, int stage, int level, const double* D, const double* S, const double* r,
                             double aD, double aS, double aR, double* L, double* G, double* y, void* ws, int* info,
                             void* stream);
int mfgm_packed_selinv_level(const mfgm_plan* plan, int level, const double* L, const double* G, const double* y, double* Sig,
                             double* Sub, double* x, void* ws, void* stream);

/* the same for plans with 8 < d <= 32: the level-0 kernel of one pass alone (which 0 reduce, 1 forward: D, S, r, L, G, y as in
 * mfgm_packed_factor_form; 2 backward: L, G, y, Sig, Sub, x as in mfgm_packed_selinv_form; unused arguments NULL; site1 / site2 non-NULL:
 * the inputs of mfgm_sparse_factor, D = pdiag, S = psub, r = plin) */
int mfgm_wide_stage(const mfgm_plan* plan, int form, int which, const double* D, const double* S, const double* r, double aD, double aS,
                    double aR, double* L, double* G, double* y, double* Sig, double* Sub, double* x, const double* site1,
                    const double* site2, void* ws, int* info, void* stream);

const char* mfgm_version(void);

/* Natural [B, T, 3d] view (mu, diag Sigma_tt, diag Sigma_{t+1,t}) of the packed moment array written by
 * mfgm_packed_selinv_mom; used by the prior-parameter gradients (variational_cvi_sde.py:495-506).  d <= 8 plans. */
int mfgm_unpack_moments(const mfgm_plan* plan, const double* packed_mom, double* natural, void* stream);

/* ---- one long chain over several processes (wide plans, i.e. 8 < d <= 32) ---------------------------------
 * The reference has no counterpart (it runs one chain on one device); this is the partitioned solver's own level structure used
 * across GPUs.  Every process creates the SAME plan (same B, T, d, R0, Rup), owns a contiguous range [seg_lo, seg_hi) of the
 * level-0 segments (nodes [seg_lo*R0, min(seg_hi*R0, T))), and holds the inputs of its own nodes plus the one sub-diagonal
 * block to the left of its first node (arrays are addressed with global node indices).
 *   factor:  mfgm_packed_factor_phase(phase 0)   zero the level-1 inputs, level-0 reduce of the owned segments
 *            all-reduce(sum) of the workspace region given by mfgm_plan_exchange_region (every entry has one writer)
 *            mfgm_packed_factor_phase(phase 1)   coarser levels (replicated on every process), level-0 forward of the owned
 *                                                segments; logdet / quad are the partial sums over the owned segments
 *   selected inverse: mfgm_packed_selinv as usual (coarser levels replicated, level 0 on the owned segments, no communication). */
int mfgm_plan_set_shard(mfgm_plan* plan, int seg_lo, int seg_hi);
/* The same with the exchange at a coarser level ("each GPU eliminates its interior ... then a reduced system of 8 d x d
 * interface blocks"): the process owns the nodes [node_lo, node_hi) of level `level` (1 <= level < nlevels) and, below it, the
 * segments those nodes stand for; phase 0 runs the reduces of the levels 0 .. level-1 on them, the exchange region is the
 * level's inputs -- n_level x (3 d^2 + 2 d) doubles per chain, e.g. 8 x 800 at config 5 with level chosen so that n_level = 8 --
 * and phase 1 solves the levels >= level on every process and sweeps back down its own segments.  No other data cross processes
 * (the forward sweep stores the factor blocks of the separator on the left of the range, which it reconstructs anyway). */
int mfgm_plan_set_shard_level(mfgm_plan* plan, int level, int node_lo, int node_hi);
/* Not positive definite: the factorisation kernels leave in the caller's `info` word (zeroed by the caller) the FIRST failing location --
 * lowest level, then lowest (chain, segment) -- instead of a bare flag.  mfgm_plan_decode_info turns a copy of the word into
 * out4 = (chain b, first node k_lo, one past the last node k_hi, level): the pivot block that failed belongs to a node in [k_lo, k_hi) of
 * chain b (level > 0: a separator system that stands for those nodes).  Returns 0 when the word is 0 (out4 = -1), 2 when a failure was
 * reported (out4 = -1 if the reporting kernel had no location to give), as TF's Cholesky op fails the step in the reference
 * (block_tri_diag.py:428-440).  mfgm_plan_check_info copies the word from the device (synchronises `stream`) and decodes it. */
int mfgm_plan_decode_info(const mfgm_plan* plan, int info_value, int* out4);
int mfgm_plan_check_info(const mfgm_plan* plan, const int* info, int* out4, void* stream);

/* out[0..3] = n, R, P, Lpad of a level */
int mfgm_plan_level(const mfgm_plan* plan, int level, int* out4);
int mfgm_plan_exchange_region(const mfgm_plan* plan, size_t* offset_doubles, size_t* count_doubles);
int mfgm_packed_factor_phase(const mfgm_plan* plan, int phase, const double* D, const double* S, const double* r, double aD,
                             double aS, double aR, double* L, double* G, double* y, double* logdet, double* quad, void* ws,
                             int* info, void* stream);
/* the phases with the factor arrays in the given form (see mfgm_packed_factor_form) */
int mfgm_packed_factor_phase_form(const mfgm_plan* plan, int form, int phase, const double* D, const double* S, const double* r,
                                  double aD, double aS, double aR, double* L, double* G, double* y, double* logdet, double* quad,
                                  void* ws, int* info, void* stream);

/* ---- batched small dense SPD algebra on natural-layout arrays -----------------------------------------------------------
 * The per-time-step algebra around the sweeps that has no fused kernel of its own: replaces the reference's
 * tf.linalg.cholesky / tf.linalg.cholesky_solve / tf.linalg.triangular_solve calls on [..., d, d] blocks
 * (ssm_gaussian_transformations.py:93-178, 459-511, 515-593; conditionals.py:207-256; kalman_filter.py:298-345).
 * A: [N, d, d] SPD (lower triangle read) -> L: [N, d, d] lower (upper zero); a non-positive pivot sets *info.  d <= 32. */
int mfgm_batched_cholesky(int N, int d, const double* A, double* L, int* info, void* stream);
/* X = L^{-1} B (mode 1), L^{-T} B (mode 2) or (L L^T)^{-1} B (mode 3) for B, X: [N, d, m]; L: [lbatch, d, d] with
 * lbatch = N or 1 (one factor shared by the whole batch).  X may alias B. */
int mfgm_batched_trsm(int N, int d, int m, int lbatch, const double* L, const double* B, double* X, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFGM_H */
