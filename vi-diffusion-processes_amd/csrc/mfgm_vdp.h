// VDP (Archambeau et al. 2007) kernels: markovflow/models/vi_sde.py `VariationalMarkovGP`.
//   k_vdp_to_ssm       forward_pass: linear drift (-A, b) -> Euler SSM parameters (LinearDrift.to_ssm, drift.py:66-117)
//   k_vdp_esde         E_sde = 1/2 dt sum_t E_q |f_L - f|^2_{q^{-1}} in closed form (sde_utils.py:182-249)
//   k_vdp_lagrange<P>  update_lagrange (vi_sde.py:289-347): the backward Euler sweep for (psi, lambda) as a partitioned
//                      affine recurrence  psi_{t-1} = psi_t (I - 2 dt A_t) + c_t,  lambda_{t-1} = (I - dt A_t) lambda_t + e_t
//                      (the reference's Python loop with O(T) tensor_scatter_nd_update copies), in three passes:
//                      segment summaries, a per-chain scan over the segments (one wavefront per chain), and the final sweep.
//   k_vdp_update_param update_param (vi_sde.py:377-414)
// Drifts are per-dimension cubics f_i(x) = af x - bf x^3 (OU, double-well) with diagonal diffusion q.
#pragma once
#include "mfgm_local.h"

namespace mfgm {

struct VdpParams {
    double af[8], bf[8];   // drift f_i(x) = af_i x - bf_i x^3
    double q[8];           // diagonal of the diffusion matrix
    double mu0[8];         // q(x0) mean
    double chol0[36];      // q(x0) Cholesky factor (packed lower triangle)
    double dt;
    double lr;
    double clip;           // > 0: NaN scrubbing and clipping of the Lagrange sweep's inputs (stabilize_system)
};

// value of x in lane `src` (uniform) of the wavefront
MFGM_DEV double vdp_bcast(double x, int src) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// value of x in lane (own - o) of the wavefront (own value where there is no such lane)
MFGM_DEV double vdp_up(double x, int o) { return __shfl_up(x, (unsigned)o, 64); }

MFGM_DEV double vdp_stab(double x, double c) {
    x = (x != x) ? 1e-8 : x;
    return fmin(fmax(x, -c), c);
}

// Gaussian moments of the cubic drift: E f, E f', Var f and partials (same algebra as cubic_moments of mfgm_sde.h)
template <int D>
MFGM_DEV void drift_moments(const VdpParams& pr, const double (&m)[D], const double (&S)[MFGM_NTRI(D)], double (&Ef)[D],
                            double (&Jf)[D], double (&Vf)[D], double (&Ef_v)[D], double (&J_m)[D], double (&J_v)[D],
                            double (&V_m)[D], double (&V_v)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double al = pr.af[i], be = pr.bf[i], mi = m[i], v = S[tix(i, i)];
        const double m2 = mi * mi, a = m2 + v;
        Ef[i] = al * mi - be * mi * (m2 + 3.0 * v);
        Jf[i] = al - 3.0 * be * a;
        Vf[i] = al * al * v - 6.0 * al * be * v * a + be * be * v * (9.0 * m2 * m2 + 36.0 * m2 * v + 15.0 * v * v);
        Ef_v[i] = -3.0 * be * mi;
        J_m[i] = -6.0 * be * mi;
        J_v[i] = -3.0 * be;
        V_m[i] = -12.0 * al * be * mi * v + be * be * mi * v * (36.0 * m2 + 72.0 * v);
        V_v[i] = al * al - 6.0 * al * be * (m2 + 2.0 * v) + be * be * (9.0 * m2 * m2 + 72.0 * m2 * v + 45.0 * v * v);
    }
}

// Per-node energy e_t = 1/2 sum_i (1/q_i) E (f_L,i - f_i)^2  (E_sde = dt * sum_t e_t) and, when GRAD, its gradients
// wrt m (dEdm) and the symmetric block S (dEdS, packed) -- i.e. the reference's dE/dm / dt and dE/dS / dt.
template <int D, bool GRAD>
MFGM_DEV double vdp_energy(const VdpParams& pr, const double (&m)[D], const double (&S)[MFGM_NTRI(D)], const double (&A)[D * D],
                           const double (&b)[D], double (&dEdm)[D], double (&dEdS)[MFGM_NTRI(D)]) {
    constexpr int ET = MFGM_NTRI(D);
    double Ef[D], Jf[D], Vf[D], Ef_v[D], J_m[D], J_v[D], V_m[D], V_v[D];
    drift_moments<D>(pr, m, S, Ef, Jf, Vf, Ef_v, J_m, J_v, V_m, V_v);
    double e = 0.0;
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < D; ++i) dEdm[i] = 0.0;
#pragma unroll
        for (int k = 0; k < ET; ++k) dEdS[k] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        // l = row i of the linear drift matrix (-A); LS_i = l^T S ; El_i = l.m + b_i
        double l[D], ls[D];
#pragma unroll
        for (int k = 0; k < D; ++k) l[k] = -A[i * D + k];
        double El = b[i], lsl = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) t = __builtin_fma(l[j], S[six(j, k)], t);
            ls[k] = t;
            El = __builtin_fma(l[k], m[k], El);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) lsl = __builtin_fma(ls[k], l[k], lsl);
        const double r = El - Ef[i], w = 1.0 / pr.q[i];
        e += 0.5 * w * (lsl - 2.0 * ls[i] * Jf[i] + Vf[i] + r * r);
        if (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) dEdm[k] += w * r * l[k];
            dEdm[i] += 0.5 * w * (-2.0 * r * Jf[i] - 2.0 * ls[i] * J_m[i] + V_m[i]);
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int c = 0; c <= a; ++c) {
                    double g = l[a] * l[c];
                    if (a == i) g -= Jf[i] * l[c];
                    if (c == i) g -= Jf[i] * l[a];
                    dEdS[tix(a, c)] += 0.5 * w * g;
                }
            dEdS[tix(i, i)] += 0.5 * w * (-2.0 * ls[i] * J_v[i] + V_v[i] - 2.0 * r * Ef_v[i]);
        }
    }
    return e;
}

// ---- forward_pass: SSM parameters of dx = (-A x + b) dt + sqrt(q) dB ---------------------------------------------
template <int D>
__global__ __launch_bounds__(64) void k_vdp_to_ssm(LevelDesc lv, VdpParams pr, const double* __restrict__ Am,
                                                  const double* __restrict__ bm, double* __restrict__ Ag,
                                                  double* __restrict__ offg, double* __restrict__ cholg) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    (void)b;
    double bprev[D];
    if (p > 0) ld_node<D>(bm, R, R - 1, LaneRef::of(lane - 1), bprev);
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            const int t = p * R + s;
            double off[D], ch[ET], A[EF];
#pragma unroll
            for (int e = 0; e < ET; ++e) ch[e] = 0.0;
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) off[i] = pr.mu0[i];
#pragma unroll
                for (int e = 0; e < ET; ++e) ch[e] = pr.chol0[e];
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    off[i] = pr.dt * bprev[i];
                    ch[tix(i, i)] = sqrt(pr.dt * pr.q[i]);
                }
            }
            if (t + 1 < n) {
                ld_node<EF>(Am, R, s, me, A);
                ld_node<D>(bm, R, s, me, bprev);
#pragma unroll
                for (int e = 0; e < EF; ++e) A[e] = -pr.dt * A[e];
#pragma unroll
                for (int i = 0; i < D; ++i) A[i * D + i] += 1.0;
            } else {
#pragma unroll
                for (int e = 0; e < EF; ++e) A[e] = 0.0;
            }
            if (pr.clip > 0.0) {
                // stabilize_system (vi_sde.py:186-200): NaN -> 1e-8, state transitions and offsets clipped to [-1, 1]
#pragma unroll
                for (int e = 0; e < EF; ++e) A[e] = vdp_stab(A[e], 1.0);
                if (t > 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) off[i] = vdp_stab(off[i], 1.0);
                }
            }
            st_node<D>(offg, R, s, me, off);
            st_node<ET>(cholg, R, s, me, ch);
            st_node<EF>(Ag, R, s, me, A);
        }
    }
}

// ---- forward_pass without the detour over SSM arrays: (A, b) -> precision blocks of the Euler chain ------------------------------
// With transitions T_t = I - dt A_t (t -> t+1), offsets o_{t+1} = dt b_t, process precision W = diag(1 / (dt q)) and q(x0) = N(mu0, P0):
//   diag_t = [t = 0 ? P0^{-1} : W] + T_t^T W T_t [t < n-1],   sub_t = -W T_t,   lin_t = [t = 0 ? P0^{-1} mu0 : W o_t] - T_t^T W o_{t+1} [t < n-1]
// i.e. k_vdp_to_ssm followed by k_ssm_to_naturals<precision> in one pass (42 doubles read, 63 written per node instead of
// 42 + 63 + 63 + 63).  p0inv [B][ET], p0lin [B][D]: P0^{-1} (packed lower triangle) and P0^{-1} mu0 of every trajectory.
template <int D>
__global__ __launch_bounds__(64) void k_vdp_to_naturals(LevelDesc lv, VdpParams pr, const double* __restrict__ Am,
                                                       const double* __restrict__ bm, const double* __restrict__ p0inv,
                                                       const double* __restrict__ p0lin, double* __restrict__ ling,
                                                       double* __restrict__ diagg, double* __restrict__ subg) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    double W[D];
#pragma unroll
    for (int i = 0; i < D; ++i) W[i] = 1.0 / (pr.dt * pr.q[i]);
    double bprev[D];
#pragma unroll
    for (int i = 0; i < D; ++i) bprev[i] = 0.0;
    if (p > 0) ld_node<D>(bm, R, R - 1, LaneRef::of(lane - 1), bprev);
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            const int t = p * R + s;
            const bool has_next = t + 1 < n;
            double lin[D], dg[ET], sb[EF];
            if (t == 0) {
#pragma unroll
                for (int e = 0; e < ET; ++e) dg[e] = p0inv[(size_t)b * ET + e];
#pragma unroll
                for (int i = 0; i < D; ++i) lin[i] = p0lin[(size_t)b * D + i];
            } else {
#pragma unroll
                for (int e = 0; e < ET; ++e) dg[e] = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double o = pr.dt * bprev[i];
                    if (pr.clip > 0.0) o = vdp_stab(o, 1.0);
                    dg[tix(i, i)] = W[i];
                    lin[i] = W[i] * o;
                }
            }
            if (has_next) {
                double A[EF], on[D];
                ld_node<EF>(Am, R, s, me, A);
                ld_node<D>(bm, R, s, me, bprev);
#pragma unroll
                for (int e = 0; e < EF; ++e) A[e] = -pr.dt * A[e];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    A[i * D + i] += 1.0;
                    on[i] = pr.dt * bprev[i];
                }
                if (pr.clip > 0.0) {      // stabilize_system, as in k_vdp_to_ssm
#pragma unroll
                    for (int e = 0; e < EF; ++e) A[e] = vdp_stab(A[e], 1.0);
#pragma unroll
                    for (int i = 0; i < D; ++i) on[i] = vdp_stab(on[i], 1.0);
                }
#pragma unroll
                for (int i = 0; i < D; ++i) {
#pragma unroll
                    for (int j = 0; j < D; ++j) sb[i * D + j] = -W[i] * A[i * D + j];
                }
                // diag += T^T W T = -T^T sub ;  lin -= T^T W o_next
#pragma unroll
                for (int i = 0; i < D; ++i) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        double acc = dg[tix(i, j)];
#pragma unroll
                        for (int k = 0; k < D; ++k) acc = __builtin_fma(-A[k * D + i], sb[k * D + j], acc);
                        dg[tix(i, j)] = acc;
                    }
                    double acc = lin[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) acc = __builtin_fma(-A[k * D + i] * W[k], on[k], acc);
                    lin[i] = acc;
                }
            } else {
#pragma unroll
                for (int e = 0; e < EF; ++e) sb[e] = 0.0;
            }
            st_node<D>(ling, R, s, me, lin);
            st_node<ET>(diagg, R, s, me, dg);
            st_node<EF>(subg, R, s, me, sb);
        }
    }
}

// ---- forward_pass as the moment recursion it is in the reference (vi_sde.py:171-204: the marginals of the SSM) --------------------
//   m_{t+1} = T_t m_t + o_{t+1},   S_{t+1} = T_t S_t T_t^T + Q        (T_t = I - dt A_t, o_{t+1} = dt b_t, Q = diag(dt q))
// is an affine recurrence for m and a linear one for S, so it partitions like the Lagrange sweep: PASS 1 composes a segment's
// transitions into (Phi, Qacc, macc) with  value(first node of segment p+1) = Phi value(first node of p) Phi^T + Qacc  (and
// Phi m + macc), a wavefront per chain chains the segment maps from q(x0), PASS 3 sweeps each segment from its first node.
// 42 doubles read twice and 27 written per node, against 348 for the route over precision blocks, factorisation and selected
// inverse; and a sum of positive semi-definite terms instead of an inverse.
template <int D>
MFGM_DEV void vdp_transition(const VdpParams& pr, double (&A)[D * D], double (&o)[D]) {
    // in: A_t, b_t; out: T_t = I - dt A_t, o_{t+1} = dt b_t (stabilize_system: NaN -> 1e-8, clipped to [-1, 1])
#pragma unroll
    for (int e = 0; e < D * D; ++e) A[e] = -pr.dt * A[e];
#pragma unroll
    for (int i = 0; i < D; ++i) { A[i * D + i] += 1.0; o[i] = pr.dt * o[i]; }
    if (pr.clip > 0.0) {
#pragma unroll
        for (int e = 0; e < D * D; ++e) A[e] = vdp_stab(A[e], 1.0);
#pragma unroll
        for (int i = 0; i < D; ++i) o[i] = vdp_stab(o[i], 1.0);
    }
}
// S <- Phi S Phi^T + Qd  (S, result: packed lower triangles; Qd added on the diagonal when given)
template <int D>
MFGM_DEV void vdp_congruence(const double (&Phi)[D * D], double (&S)[MFGM_NTRI(D)]) {
    double PS[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(Phi[i * D + k], S[six(k, j)], t);
            PS[i * D + j] = t;
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = __builtin_fma(PS[i * D + k], Phi[j * D + k], t);
            S[tix(i, j)] = t;
        }
}
template <int D>
MFGM_DEV void vdp_advance(const VdpParams& pr, const double (&T)[D * D], const double (&o)[D], double (&m)[D],
                          double (&S)[MFGM_NTRI(D)]) {
    double t[D];
    gemv<D>(T, m, t);
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = t[i] + o[i];
    vdp_congruence<D>(T, S);
#pragma unroll
    for (int i = 0; i < D; ++i) S[tix(i, i)] += pr.dt * pr.q[i];
}

// seg: one record per lane (segment), contiguous -- Phi [d^2], Qacc [ET], macc [d], then the boundary values m [d], S [ET] at the
// segment's first node -- so that the scan kernel reads the consecutive segments of a lane as one run (with one array per element,
// strided by the lane count, every load of the scan was a cache line of its own: 0.29 ms, 0.16 ms like this).
// LAG: what the pass does for the Lagrange call that follows on the SAME (A, b) and on the marginals this recursion produces; results go
// to `lseg`, that call's seg array.
//   LAG 1 (PASS 1): the linear parts of the Lagrange sweep's segment maps (k_vdp_lagrange_products: Mpsi, Mlam, products of
//     I - 2 dt A_t / I - dt A_t over the segment's transitions t >= 1).  They depend on A alone and this pass has A in registers with
//     the vector ALU a quarter busy (it is bound by reading A): the two d x d accumulators live in LDS (one wavefront per SIMD,
//     2 d^2 x 64 doubles = 36 KB per workgroup at d = 6) and the 8 d^2 bytes per node of the products pass are not read again
//     (config 3: the pass goes from 0.20 to 0.28 ms and the products pass, 0.18 ms, is gone).
//   LAG 2 (PASS 3): the whole of the Lagrange sweep's PASS 1 as well -- the affine offsets (Cpsi, Clam) of the segment maps.  The
//     descending recurrence  psi <- psi X_t + c_t,  lam <- Y_t lam + e_t  (X_t = I - 2 dt A_t, Y_t = I - dt A_t, c_t = dt dE/dS - d_obs_S,
//     e_t = dt dE/dm - d_obs_m) has the segment map  psi_out = psi_in (X_hi .. X_lo) + sum_s c_s (X_{s-1} .. X_lo), which an ASCENDING
//     sweep accumulates as  Cpsi += c_s Mp,  Mp <- X_s Mp  (Clam += Ml e_s,  Ml <- Ml Y_s): this sweep visits the nodes in that order
//     with (m_t, S_t, A_t, b_t) in registers, and its energy term is the value of the function whose gradient c_t, e_t need.  The
//     Lagrange call then starts at its segment scan: PASS 1 and the products pass (0.43 + 0.18 ms of config 3's 1.95) are not run and
//     their 600 + 288 bytes per node not read.
template <int D, int PASS, int LAG = 0>
__global__ __launch_bounds__(64) void k_vdp_marginals(LevelDesc lv, VdpParams pr, const double* __restrict__ Am,
                                                     const double* __restrict__ bm, double* __restrict__ mug,
                                                     double* __restrict__ Sigg, double* __restrict__ seg,
                                                     double* __restrict__ part /* PASS 3, optional: per-lane sum of the E_sde terms */,
                                                     double* __restrict__ lseg = nullptr, const double* __restrict__ yR = nullptr,
                                                     const double* __restrict__ dobsS = nullptr, const int* __restrict__ obs_count = nullptr,
                                                     const double* __restrict__ dobs_const = nullptr) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, MAP = EF + ET + D, STR = MAP + D + ET;
    static_assert(LAG == 0 || (LAG == 1 && PASS == 1) || (LAG == 2 && PASS == 3), "products ride in pass 1, the whole first Lagrange pass in pass 3");
    __shared__ double prod_lds[LAG ? 2 * EF * 64 : 1];
    double* const Mp = prod_lds + threadIdx.x;                 // element e at Mp[e * 64]
    double* const Ml = prod_lds + (LAG ? EF * 64 : 0) + threadIdx.x;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    if (LAG) {
#pragma unroll
        for (int e = 0; e < EF; ++e) { Mp[e * 64] = (e % (D + 1) == 0) ? 1.0 : 0.0; Ml[e * 64] = (e % (D + 1) == 0) ? 1.0 : 0.0; }
    }
    double Cpsi[LAG == 2 ? EF : 1], Clam[LAG == 2 ? D : 1];
    if (LAG == 2) {
#pragma unroll
        for (int e = 0; e < EF; ++e) Cpsi[e] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) Clam[i] = 0.0;
    }
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int p = lane % P;
    const int len = min(R, n - p * R);
    const int nt = min(len, n - 1 - p * R);        // transitions that start in this segment: nodes s < nt
    double m[D], S[ET], Phi[EF];
    if (PASS == 1) {
#pragma unroll
        for (int e = 0; e < EF; ++e) Phi[e] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) { Phi[i * D + i] = 1.0; m[i] = 0.0; }
#pragma unroll
        for (int e = 0; e < ET; ++e) S[e] = 0.0;
    } else {
        const double* bnd = seg + MAP;                  // boundary values follow the map in the lane's record
#pragma unroll
        for (int i = 0; i < D; ++i) m[i] = bnd[(size_t)lane * STR + i];
#pragma unroll
        for (int e = 0; e < ET; ++e) S[e] = bnd[(size_t)lane * STR + (D + e)];
    }
    double An[EF], bn[D];
    int cntn = 0;
    double esde = 0.0;
    auto load_jump = [&](int s) {        // the observation count of node s (LAG 2), requested with that node's A, b
        if (LAG == 2 && obs_count) cntn = obs_count[((size_t)me.tile * R + s) * 64 + me.l];
    };
    if (nt > 0) {
        ld_node<EF>(Am, R, 0, me, An);
        ld_node<D>(bm, R, 0, me, bn);
        load_jump(0);
    }
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            if (PASS == 3) {
                st_node<D>(mug, R, s, me, m);
                st_node<ET>(Sigg, R, s, me, S);
            }
            if (s < nt) {
                double T[EF], o[D];
#pragma unroll
                for (int e = 0; e < EF; ++e) T[e] = An[e];
#pragma unroll
                for (int i = 0; i < D; ++i) o[i] = bn[i];
                const int cnt = cntn;
                auto request_next = [&]() {
                    if (s + 1 < nt) {
                        ld_node<EF>(Am, R, s + 1, me, An);
                        ld_node<D>(bm, R, s + 1, me, bn);
                        load_jump(s + 1);
                    }
                };
                if (LAG != 2) request_next();
                if (PASS == 3 && LAG != 2 && part) {
                    // E_sde term of this transition (k_vdp_esde) while (m_t, S_t, A_t, b_t) are in registers
                    double dm[D], dS[ET];
                    esde += vdp_energy<D, false>(pr, m, S, T, o, dm, dS);
                }
                if (LAG == 2) {
                    // the energy term with its gradients; at t >= 1 the offsets of the Lagrange segment map take c_t, e_t (k_vdp_lagrange)
                    double dm[D], dS[ET];
                    esde += vdp_energy<D, true>(pr, m, S, T, o, dm, dS);
                    // the next node's inputs are requested HERE, not before the energy term, whose temporaries and the 48 doubles
                    // in flight do not fit the register file together (716 B of scratch, 1.2 ms); what is left of this node's work
                    // (~1 100 fp64 instructions) covers the latency
                    __builtin_amdgcn_sched_barrier(0);
                    request_next();
                    __builtin_amdgcn_sched_barrier(0);
                    if (p * R + s >= 1) {
                        if (pr.clip > 0.0) {        // vi_sde.py:312-323
#pragma unroll
                            for (int i = 0; i < D; ++i) dm[i] = vdp_stab(dm[i], pr.clip);
#pragma unroll
                            for (int e = 0; e < ET; ++e) dS[e] = vdp_stab(dS[e], pr.clip);
                        }
                        // e_t = dt dE/dm - d_obs_m  (in dm),  c_t = dt dE/dS - d_obs_S  (in dS)
#pragma unroll
                        for (int i = 0; i < D; ++i) dm[i] *= pr.dt;
#pragma unroll
                        for (int e = 0; e < ET; ++e) dS[e] *= pr.dt;
                        if (!obs_count || cnt != 0) {
                            // the jump terms: zero except at observation nodes (whole wavefronts when the segments are aligned with the
                            // observation grid), so R^-1 y and the observation block are read here, not with every node
                            double yr[D], dob[ET];
                            ld_node<D>(yR, R, s, me, yr);
                            if (obs_count) {
#pragma unroll
                                for (int e = 0; e < ET; ++e) dob[e] = (double)cnt * dobs_const[e];
                            } else {
                                ld_node<ET>(dobsS, R, s, me, dob);
                            }
                            if (pr.clip > 0.0) {
#pragma unroll
                                for (int e = 0; e < ET; ++e) dob[e] = vdp_stab(dob[e], pr.clip);
                            }
#pragma unroll
                            for (int i = 0; i < D; ++i) {
                                double dom = yr[i];
#pragma unroll
                                for (int k = 0; k < D; ++k) dom = __builtin_fma(2.0 * dob[six(i, k)], m[k], dom);
                                if (pr.clip > 0.0) dom = vdp_stab(dom, pr.clip);
                                dm[i] -= dom;
                            }
#pragma unroll
                            for (int e = 0; e < ET; ++e) dS[e] -= dob[e];
                        }
                        // Cpsi += c Mp, then Mp <- (I - 2 dt A) Mp, a column at a time
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            double c[D];
#pragma unroll
                            for (int k = 0; k < D; ++k) c[k] = Mp[(k * D + j) * 64];
#pragma unroll
                            for (int i = 0; i < D; ++i) {
                                double t = 0.0, u = Cpsi[i * D + j];
#pragma unroll
                                for (int k = 0; k < D; ++k) {
                                    t = __builtin_fma(T[i * D + k], c[k], t);
                                    u = __builtin_fma(dS[six(i, k)], c[k], u);
                                }
                                Cpsi[i * D + j] = u;
                                Mp[(i * D + j) * 64] = __builtin_fma(-2.0 * pr.dt, t, c[i]);
                            }
                        }
                        // Clam += Ml e, then Ml <- Ml (I - dt A), a row at a time
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            double r[D], u = Clam[i];
#pragma unroll
                            for (int k = 0; k < D; ++k) { r[k] = Ml[(i * D + k) * 64]; u = __builtin_fma(r[k], dm[k], u); }
                            Clam[i] = u;
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                double t = 0.0;
#pragma unroll
                                for (int k = 0; k < D; ++k) t = __builtin_fma(r[k], T[k * D + j], t);
                                Ml[(i * D + j) * 64] = __builtin_fma(-pr.dt, t, r[j]);
                            }
                        }
                    }
                }
                if (LAG == 1 && p * R + s >= 1) {
                    // ascending order of the same products: Mpsi <- (I - 2 dt A_t) Mpsi, Mlam <- Mlam (I - dt A_t); T still holds A_t here.
                    // Each accumulator is read from LDS in one batch (14 waits per node instead of 36; the pass, ~1 900 instructions per node
                    // with the products in it, is bound by instruction issue either way: 0.28 ms against 0.20 + 0.18 apart)
                    {
                        double mp[EF];
#pragma unroll
                        for (int e = 0; e < EF; ++e) mp[e] = Mp[e * 64];
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                double t = 0.0;
#pragma unroll
                                for (int k = 0; k < D; ++k) t = __builtin_fma(T[i * D + k], mp[k * D + j], t);
                                Mp[(i * D + j) * 64] = __builtin_fma(-2.0 * pr.dt, t, mp[i * D + j]);
                            }
                    }
                    {
                        double ml[EF];
#pragma unroll
                        for (int e = 0; e < EF; ++e) ml[e] = Ml[e * 64];
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                double t = 0.0;
#pragma unroll
                                for (int k = 0; k < D; ++k) t = __builtin_fma(ml[i * D + k], T[k * D + j], t);
                                Ml[(i * D + j) * 64] = __builtin_fma(-pr.dt, t, ml[i * D + j]);
                            }
                    }
                }
                vdp_transition<D>(pr, T, o);
                vdp_advance<D>(pr, T, o, m, S);
                if (PASS == 1) {
                    double t[EF];
                    gemm<D>(T, Phi, t);
#pragma unroll
                    for (int e = 0; e < EF; ++e) Phi[e] = t[e];
                }
            }
        }
    }
    if (PASS == 3 && part) part[lane] = esde;
    if (LAG == 2) {
        constexpr int LSTR = 4 * EF + 2 * D;                  // the Lagrange sweep's segment record: Mpsi, Cpsi, Mlam, Clam, boundary values
#pragma unroll
        for (int e = 0; e < EF; ++e) {
            lseg[(size_t)lane * LSTR + e] = Mp[e * 64];
            lseg[(size_t)lane * LSTR + (EF + e)] = Cpsi[e];
            lseg[(size_t)lane * LSTR + (2 * EF + e)] = Ml[e * 64];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) lseg[(size_t)lane * LSTR + (3 * EF + i)] = Clam[i];
    }
    if (PASS == 1) {
#pragma unroll
        for (int e = 0; e < EF; ++e) seg[(size_t)lane * STR + e] = Phi[e];
#pragma unroll
        for (int e = 0; e < ET; ++e) seg[(size_t)lane * STR + (EF + e)] = S[e];
#pragma unroll
        for (int i = 0; i < D; ++i) seg[(size_t)lane * STR + (EF + ET + i)] = m[i];
        if (LAG == 1) {
            constexpr int LSTR = 4 * EF + 2 * D;              // the Lagrange sweep's segment record (k_vdp_lagrange_products)
#pragma unroll
            for (int e = 0; e < EF; ++e) {
                lseg[(size_t)lane * LSTR + e] = Mp[e * 64];
                lseg[(size_t)lane * LSTR + (2 * EF + e)] = Ml[e * 64];
            }
        }
    }
}

// value at the first node of segment p+1 = map_p(value at the first node of segment p), from q(x0) = N(q0_mu[b], q0_cov[b]):
// one wavefront per chain, as k_vdp_lagrange_scan_wave (lane j composes K = ceil(P / 64) consecutive maps, readlane chain, replay).
constexpr int kScanBlock = 256;       // lanes per chain in the segment-map scans: four wavefronts, K = ceil(P / 256) maps per lane
template <int D>
__global__ __launch_bounds__(kScanBlock) void k_vdp_marginals_scan(LevelDesc lv, const double* __restrict__ q0_mu,
                                                                  const double* __restrict__ q0_cov, double* __restrict__ seg0,
                                                                  double* __restrict__ seg1 = nullptr /* blockIdx.y == 1 */) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, MAP = EF + ET + D, STR = MAP + D + ET;
    double* __restrict__ seg = blockIdx.y ? seg1 : seg0;
    __shared__ double wtot[kScanBlock / 64][MAP];  // the composed map of each wavefront's 64 lanes
    const int b = blockIdx.x, jl = threadIdx.x, j = jl & 63, wv = jl >> 6;
    const int P = lv.P;
    const int K = (P + kScanBlock - 1) / kScanBlock;
    const int lo = jl * K;                         // this lane's segments: lo, lo+1, ..., min(lo+K, P) - 1
    double* bnd = seg + MAP;
    auto load_map = [&](int p, double (&Ph)[EF], double (&Qa)[ET], double (&ma)[D]) {
        const size_t lane = (size_t)b * P + p;
#pragma unroll
        for (int e = 0; e < EF; ++e) Ph[e] = seg[(size_t)lane * STR + e];
#pragma unroll
        for (int e = 0; e < ET; ++e) Qa[e] = seg[(size_t)lane * STR + (EF + e)];
#pragma unroll
        for (int i = 0; i < D; ++i) ma[i] = seg[(size_t)lane * STR + (EF + ET + i)];
    };
    auto apply = [&](const double (&Ph)[EF], const double (&Qa)[ET], const double (&ma)[D], double (&m)[D], double (&S)[ET]) {
        double t[D];
        gemv<D>(Ph, m, t);
#pragma unroll
        for (int i = 0; i < D; ++i) m[i] = t[i] + ma[i];
        vdp_congruence<D>(Ph, S);
#pragma unroll
        for (int e = 0; e < ET; ++e) S[e] += Qa[e];
    };
    // the lane's composed map (Phi, Qacc, macc): identity when it has no segment
    double Phi[EF], Qc[ET], mc[D];
#pragma unroll
    for (int e = 0; e < EF; ++e) Phi[e] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) { Phi[i * D + i] = 1.0; mc[i] = 0.0; }
#pragma unroll
    for (int e = 0; e < ET; ++e) Qc[e] = 0.0;
    for (int k = 0; k < K; ++k) {
        const int p = lo + k;
        if (p < P) {
            double Ph[EF], Qa[ET], ma[D], t[EF];
            load_map(p, Ph, Qa, ma);
            apply(Ph, Qa, ma, mc, Qc);             // (Phi_p Phi, Phi_p Qc Phi_p^T + Q_p, Phi_p mc + m_p)
            gemm<D>(Ph, Phi, t);
#pragma unroll
            for (int e = 0; e < EF; ++e) Phi[e] = t[e];
        }
    }
    // Inclusive scan of the lanes' composed maps in log2(64) rounds (Kogge-Stone): after round o a lane holds the composition of up to 2 o
    // consecutive lanes' maps ending with its own.  (A serial chain of 64 applications through v_readlane broadcasts took 67 of the
    // kernel's 160 us at d = 6: every one of them a d x d congruence behind 126 broadcasts.)  later o earlier:
    // (Phi, Q, m) o (Pp, Qp, mp) = (Phi Pp, Phi Qp Phi^T + Q, Phi mp + m).
    for (int o = 1; o < 64; o <<= 1) {
        double Pp[EF], Qp[ET], mp[D];
#pragma unroll
        for (int e = 0; e < EF; ++e) Pp[e] = vdp_up(Phi[e], o);
#pragma unroll
        for (int e = 0; e < ET; ++e) Qp[e] = vdp_up(Qc[e], o);
#pragma unroll
        for (int i = 0; i < D; ++i) mp[i] = vdp_up(mc[i], o);
        if (j >= o) {
            double t[EF];
            apply(Phi, Qc, mc, mp, Qp);
            gemm<D>(Phi, Pp, t);
#pragma unroll
            for (int e = 0; e < EF; ++e) Phi[e] = t[e];
#pragma unroll
            for (int e = 0; e < ET; ++e) Qc[e] = Qp[e];
#pragma unroll
            for (int i = 0; i < D; ++i) mc[i] = mp[i];
        }
    }
    // the value that enters lane j's segments: q(x0) through the composed maps of the wavefronts before this one (at most three
    // applications, maps handed over through LDS), then through the composition of the lanes before it in its own wavefront
    if (j == 63) {
#pragma unroll
        for (int e = 0; e < EF; ++e) wtot[wv][e] = Phi[e];
#pragma unroll
        for (int e = 0; e < ET; ++e) wtot[wv][EF + e] = Qc[e];
#pragma unroll
        for (int i = 0; i < D; ++i) wtot[wv][EF + ET + i] = mc[i];
    }
    __syncthreads();
    double mym[D], myS[ET];
#pragma unroll
    for (int i = 0; i < D; ++i) mym[i] = q0_mu[(size_t)b * D + i];
#pragma unroll
    for (int e = 0; e < ET; ++e) myS[e] = q0_cov[(size_t)b * ET + e];
    for (int w = 0; w < wv; ++w) {
        double Pw[EF], Qw[ET], mw[D];
#pragma unroll
        for (int e = 0; e < EF; ++e) Pw[e] = wtot[w][e];
#pragma unroll
        for (int e = 0; e < ET; ++e) Qw[e] = wtot[w][EF + e];
#pragma unroll
        for (int i = 0; i < D; ++i) mw[i] = wtot[w][EF + ET + i];
        apply(Pw, Qw, mw, mym, myS);
    }
    {
        double Pe[EF], Qe[ET], me[D];
#pragma unroll
        for (int e = 0; e < EF; ++e) Pe[e] = vdp_up(Phi[e], 1);
#pragma unroll
        for (int e = 0; e < ET; ++e) Qe[e] = vdp_up(Qc[e], 1);
#pragma unroll
        for (int i = 0; i < D; ++i) me[i] = vdp_up(mc[i], 1);
        if (j > 0) apply(Pe, Qe, me, mym, myS);
    }
    for (int k = 0; k < K; ++k) {
        const int p = lo + k;
        if (p < P) {
            const size_t lane = (size_t)b * P + p;
#pragma unroll
            for (int i = 0; i < D; ++i) bnd[(size_t)lane * STR + i] = mym[i];
#pragma unroll
            for (int e = 0; e < ET; ++e) bnd[(size_t)lane * STR + (D + e)] = myS[e];
            double Ph[EF], Qa[ET], ma[D];
            load_map(p, Ph, Qa, ma);
            apply(Ph, Qa, ma, mym, myS);
        }
    }
}

// ---- a general congruence recurrence on the same three passes ------------------------------------------------------------------------
//   X_t = Phi_t X_{t-1} Phi_t^T + Q_t,   X_{-1} = 0        (Phi: FULL blocks, Q / X: SYM blocks, all T nodes of every chain)
// The two recurrences behind the exact Fisher-vector product of the tape (tape.band_of_sigma_dP_sigma: the band of Sigma dP Sigma
// from L_{t+1} = A_t L_t A_t^T + QL and its mirror image) are of this form.  PASS 1 composes a segment's nodes into (Phi, Qacc), the
// segment maps are chained by k_vdp_marginals_scan (mean part zero, start value 0), PASS 3 sweeps each segment from the value that
// enters it and stores X.  The torch route (a Hillis-Steele scan, ceil(log2 T) rounds of three batched products) moves ~17 x more bytes.
// Two independent recurrences (the ascending and the descending one of the band) ride in one launch: blockIdx.y picks the array set.
struct ScanSet { const double* Phi; const double* Q; double* X; double* seg; };
struct ScanSets { ScanSet s[2]; };
template <int D, int PASS>
__global__ __launch_bounds__(64) void k_congruence_scan(LevelDesc lv, ScanSets sets) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D, MAP = EF + ET + D, STR = MAP + D + ET;
    const double* __restrict__ Phig = sets.s[blockIdx.y].Phi;
    const double* __restrict__ Qg = sets.s[blockIdx.y].Q;
    double* __restrict__ Xg = sets.s[blockIdx.y].X;
    double* __restrict__ seg = sets.s[blockIdx.y].seg;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int p = lane % P;
    const int len = min(R, n - p * R);
    double S[ET], Phi[EF];
    if (PASS == 1) {
#pragma unroll
        for (int e = 0; e < EF; ++e) Phi[e] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) Phi[i * D + i] = 1.0;
#pragma unroll
        for (int e = 0; e < ET; ++e) S[e] = 0.0;
    } else {
#pragma unroll
        for (int e = 0; e < ET; ++e) S[e] = seg[(size_t)lane * STR + (MAP + D + e)];     // the value entering the segment
    }
    double Pn[EF], Qn[ET];
    ld_node<EF>(Phig, R, 0, me, Pn);
    ld_node<ET>(Qg, R, 0, me, Qn);
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            double T[EF], Q[ET];
#pragma unroll
            for (int e = 0; e < EF; ++e) T[e] = Pn[e];
#pragma unroll
            for (int e = 0; e < ET; ++e) Q[e] = Qn[e];
            if (s + 1 < len) {
                ld_node<EF>(Phig, R, s + 1, me, Pn);
                ld_node<ET>(Qg, R, s + 1, me, Qn);
            }
            vdp_congruence<D>(T, S);
#pragma unroll
            for (int e = 0; e < ET; ++e) S[e] += Q[e];
            if (PASS == 3) st_node<ET>(Xg, R, s, me, S);
            if (PASS == 1) {
                double t[EF];
                gemm<D>(T, Phi, t);
#pragma unroll
                for (int e = 0; e < EF; ++e) Phi[e] = t[e];
            }
        }
    }
    if (PASS == 1) {
#pragma unroll
        for (int e = 0; e < EF; ++e) seg[(size_t)lane * STR + e] = Phi[e];
#pragma unroll
        for (int e = 0; e < ET; ++e) seg[(size_t)lane * STR + (EF + e)] = S[e];
#pragma unroll
        for (int i = 0; i < D; ++i) seg[(size_t)lane * STR + (EF + ET + i)] = 0.0;
    }
}

// ---- E_sde value (per-lane partials; times dt on the host) and optional gradient arrays --------------------------
template <int D, bool GRAD>
__global__ __launch_bounds__(64) void k_vdp_esde(LevelDesc lv, VdpParams pr, const double* __restrict__ mug,
                                                const double* __restrict__ Sigg, const double* __restrict__ Am,
                                                const double* __restrict__ bm, double* __restrict__ part,
                                                double* __restrict__ gm, double* __restrict__ gS) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    (void)b;
    double acc = 0.0;
    for (int s = 0; s < R; ++s) {
        if (s < len && p * R + s + 1 < n) {
            double m[D], S[ET], A[EF], bb[D], dm[D], dS[ET];
            ld_node<D>(mug, R, s, me, m);
            ld_node<ET>(Sigg, R, s, me, S);
            ld_node<EF>(Am, R, s, me, A);
            ld_node<D>(bm, R, s, me, bb);
            acc += vdp_energy<D, GRAD>(pr, m, S, A, bb, dm, dS);
            if (GRAD) {
                st_node<D>(gm, R, s, me, dm);
                st_node<ET>(gS, R, s, me, dS);
            }
        }
    }
    part[lane] = acc;
}

// ---- update_lagrange ------------------------------------------------------------------------------------------------
// yR (VEC) = R^{-1} y at observation nodes (zero elsewhere), dobsS (SYM) = -1/2 R^{-1} at observation nodes: the jump
// conditions d_obs_m = yR + 2 dobsS m, d_obs_S = dobsS (vi_sde.py:262-287 for a Gaussian likelihood).  With obs_count (one int per
// node, packed [tile][step][64]) and dobs_const [ET] given, dobsS is taken as obs_count * dobs_const and not read.
// PASS 1: the offsets (Cpsi, Clam) of the per-segment affine maps, i.e. the sweep with zero input (their linear parts Mpsi, Mlam come
// from k_vdp_lagrange_products); PASS 3: the sweep from the known value
// at the segment's last node (bpsi, blam: [lanes] arrays written by k_vdp_lagrange_scan_wave).
// PASS 4: PASS 3 that also makes update_param at every node it visits (A, b replaced in place; with pr.clip > 0 the stored psi /
// lambda are the clipped values update_param would leave, the recurrence itself continues with the unclipped ones).
// PASS 5: PASS 4 that keeps the multipliers of NODE 0 only, in psig [B, d, d] / lamg [B, d] (natural layout): in the trainer's loop
// (vi_markov_gp_trainer.py:56-58) update_param has consumed every other multiplier by the time the sweep leaves its node and only
// update_initial_statistics reads psi(0), lambda(0) afterwards -- 42 of the 153 doubles per node the sweep moves at d = 6 are stores
// nobody loads.
template <int D, int PASS>
__global__ __launch_bounds__(64) void k_vdp_lagrange(LevelDesc lv, VdpParams pr, const double* __restrict__ mug,
                                                    const double* __restrict__ Sigg, double* Am,
                                                    double* bm, const double* __restrict__ yR,
                                                    const double* __restrict__ dobsS, double* __restrict__ psig,
                                                    double* __restrict__ lamg, double* __restrict__ seg /* per-lane summaries */,
                                                    const int* __restrict__ obs_count, const double* __restrict__ dobs_const) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    constexpr int SEG = 2 * EF + EF + D;   // a lane's record: Mpsi, Cpsi, Mlam, Clam, then the boundary values psi [d^2], lambda [d]
    constexpr int STR = SEG + EF + D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    (void)b;
    const int N = n - 1;                     // number of transitions; psi / lambda live on nodes 0 .. N-1
    // the block every observation contributes, read once into scalar registers (read inside the node loop it is ET loads per node)
    double dcst[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e) dcst[e] = obs_count ? dobs_const[e] : 0.0;
#pragma unroll
    for (int e = 0; e < ET; ++e)
        dcst[e] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(dcst[e])), __builtin_amdgcn_readfirstlane(__double2loint(dcst[e])));
    // state: psi (full), lam (the running products of the segment maps are k_vdp_lagrange_products' job)
    double psi[EF], lam[D];
    if (PASS == 1) {
#pragma unroll
        for (int e = 0; e < EF; ++e) psi[e] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) lam[i] = 0.0;
    } else {
        // boundary value at this segment's last node (slot layout: seg2[(e)*Lpad + lane])
        const double* bnd = seg + SEG;                  // boundary values follow the map in the lane's record
#pragma unroll
        for (int e = 0; e < EF; ++e) psi[e] = bnd[(size_t)lane * STR + e];
#pragma unroll
        for (int i = 0; i < D; ++i) lam[i] = bnd[(size_t)lane * STR + (EF + i)];
    }
    for (int s = R - 1; s >= 0; --s) {
        if (s < len) {
            const int t = p * R + s;
            if (t <= N - 1) {
                if (PASS == 3) {
                    st_node<EF>(psig, R, s, me, psi);
                    st_node<D>(lamg, R, s, me, lam);
                }
                double m[D], S[ET], A[EF], bb[D];
                if (PASS >= 4 || t >= 1) {
                    ld_node<D>(mug, R, s, me, m);
                    ld_node<ET>(Sigg, R, s, me, S);
                    ld_node<EF>(Am, R, s, me, A);
                    ld_node<D>(bm, R, s, me, bb);
                }
                if (PASS >= 4) {
                    // update_param at node t (vi_sde.py:377-414) from the multipliers just obtained.  Element by element, straight to
                    // memory: the clipped multipliers, A~ = 2 q psi_c - diag(J) and the blended A never exist as arrays (this kernel
                    // has no registers to spare: three more d x d arrays put it into scratch)
                    double Ef[D], Jf[D], Vf[D], t0[D], t1[D], t2[D], t3[D], t4[D];
                    drift_moments<D>(pr, m, S, Ef, Jf, Vf, t0, t1, t2, t3, t4);
                    double* pP = psig + ((size_t)me.tile * R + s) * (size_t)(EF * 64) + me.l;
                    double* pA = Am + ((size_t)me.tile * R + s) * (size_t)(EF * 64) + me.l;
                    double bn[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        const double lc = pr.clip > 0.0 ? vdp_stab(lam[i], pr.clip) : lam[i];
                        double bt = Ef[i] - pr.q[i] * lc;
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            const double pc = pr.clip > 0.0 ? vdp_stab(psi[i * D + j], pr.clip) : psi[i * D + j];
                            const double At = 2.0 * pr.q[i] * pc - (i == j ? Jf[i] : 0.0);
                            bt = __builtin_fma(At, m[j], bt);
                            if (PASS == 4) pP[(i * D + j) * 64] = pc;
                            else if (t == 0) psig[(size_t)b * EF + i * D + j] = pc;
                            pA[(i * D + j) * 64] = (1.0 - pr.lr) * A[i * D + j] + pr.lr * At;
                        }
                        t0[i] = lc;
                        bn[i] = (1.0 - pr.lr) * bb[i] + pr.lr * bt;
                    }
                    if (PASS == 4) {
                        st_node<D>(lamg, R, s, me, t0);
                    } else if (t == 0) {
#pragma unroll
                        for (int i = 0; i < D; ++i) lamg[(size_t)b * D + i] = t0[i];
                    }
                    st_node<D>(bm, R, s, me, bn);
                }
                if (t >= 1) {
                    double dm[D], dS[ET], yr[D], dob[ET];
                    ld_node<D>(yR, R, s, me, yr);
                    if (obs_count) {
                        // every observation contributes the same block: dobsS = count * dobs_const, one int per node read instead of
                        // d (d + 1) / 2 doubles that are zero at all but the observation nodes
                        const double cnt = (double)obs_count[((size_t)me.tile * R + s) * 64 + me.l];
#pragma unroll
                        for (int e = 0; e < ET; ++e) dob[e] = cnt * dcst[e];
                    } else {
                        ld_node<ET>(dobsS, R, s, me, dob);
                    }
                    vdp_energy<D, true>(pr, m, S, A, bb, dm, dS);
                    if (pr.clip > 0.0) {        // vi_sde.py:312-323
#pragma unroll
                        for (int i = 0; i < D; ++i) dm[i] = vdp_stab(dm[i], pr.clip);
#pragma unroll
                        for (int e = 0; e < ET; ++e) { dS[e] = vdp_stab(dS[e], pr.clip); dob[e] = vdp_stab(dob[e], pr.clip); }
                    }
                    // psi <- psi - dt (psi A + psi A - dEdS) - d_obs_S ;  lam <- lam - dt (A lam - dEdm) - d_obs_m
                    double pa[EF], al[D];
                    gemm<D>(psi, A, pa);
                    gemv<D>(A, lam, al);
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double dom = yr[i];
#pragma unroll
                        for (int k = 0; k < D; ++k) dom = __builtin_fma(2.0 * dob[six(i, k)], m[k], dom);
                        if (pr.clip > 0.0) dom = vdp_stab(dom, pr.clip);
                        lam[i] = lam[i] - pr.dt * (al[i] - dm[i]) - dom;
#pragma unroll
                        for (int j = 0; j < D; ++j)
                            psi[i * D + j] = psi[i * D + j] - pr.dt * (2.0 * pa[i * D + j] - dS[six(i, j)]) - dob[six(i, j)];
                    }
                }
            }
        }
    }
    if (PASS == 1) {
        // psi / lam now hold the affine offsets (C) of the segment map: value entering the previous segment's last node
#pragma unroll
        for (int e = 0; e < EF; ++e) seg[(size_t)lane * STR + (EF + e)] = psi[e];
#pragma unroll
        for (int i = 0; i < D; ++i) seg[(size_t)lane * STR + (3 * EF + i)] = lam[i];
    }
}

// The linear parts of the segment maps: Mpsi = prod (I - 2 dt A_t), Mlam = prod (I - dt A_t) over the segment's transitions
// t >= 1, in the order the sweep visits them.  They depend on A alone: a pass of their own reads 8 d^2 bytes per node with the
// next block prefetched, and relieves PASS 1 (which otherwise carries 3 d^2 + d more doubles of state than it has registers for).
template <int D>
__global__ __launch_bounds__(64) void k_vdp_lagrange_products(LevelDesc lv, VdpParams pr, const double* __restrict__ Am,
                                                             double* __restrict__ seg) {
    constexpr int EF = D * D, STR = 4 * EF + 2 * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int p = lane % P;
    const int len = min(R, n - p * R);
    const int N = n - 1;
    double Mp[EF], Ml[EF], An[EF];
#pragma unroll
    for (int e = 0; e < EF; ++e) { Mp[e] = 0.0; Ml[e] = 0.0; An[e] = 0.0; }
#pragma unroll
    for (int i = 0; i < D; ++i) { Mp[i * D + i] = 1.0; Ml[i * D + i] = 1.0; }
    // nodes of this segment that take part: s with 1 <= t = p R + s <= N - 1
    const int s_hi = min(len - 1, N - 1 - p * R), s_lo = (p == 0) ? 1 : 0;
    if (s_hi >= s_lo) ld_node<EF>(Am, R, s_hi, me, An);
    for (int s = R - 1; s >= 0; --s) {
        if (s <= s_hi && s >= s_lo) {
            double A[EF], t1[EF], t2[EF];
#pragma unroll
            for (int e = 0; e < EF; ++e) A[e] = An[e];
            if (s - 1 >= s_lo) ld_node<EF>(Am, R, s - 1, me, An);
            gemm<D>(Mp, A, t1);
            gemm<D>(A, Ml, t2);
#pragma unroll
            for (int e = 0; e < EF; ++e) {
                Mp[e] -= 2.0 * pr.dt * t1[e];
                Ml[e] -= pr.dt * t2[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EF; ++e) {
        seg[(size_t)lane * STR + e] = Mp[e];
        seg[(size_t)lane * STR + (2 * EF + e)] = Ml[e];
    }
}

// PASS 2: value at the last node of segment p-1 = (value at the last node of segment p) o map_p, for every chain.
// One wavefront per chain (one lane walking all the segments costs a memory round trip per segment: 10.7 ms at P = 1021):
// lane j composes the maps of K = ceil(P / 64) consecutive segments, the 64 composed maps are chained through v_readlane
// broadcasts, and each lane then replays its own K segments from the value that enters them.
//   PART 0: psi (value X, maps X -> X M + C);  PART 1: lambda (value v, maps v -> M v + C)

template <int D, int PART>
MFGM_DEV void vdp_lagrange_scan_wave_body(const LevelDesc& lv, double* __restrict__ seg) {
    constexpr int EF = D * D, SEG = 3 * EF + D, STR = SEG + EF + D, NV = PART == 0 ? EF : D;
    __shared__ double wtot[kScanBlock / 64][EF + NV];   // the composed map of each wavefront's 64 lanes
    const int b = blockIdx.x, jl = threadIdx.x, j = jl & 63, wv = jl >> 6;
    const int P = lv.P;
    const int K = (P + kScanBlock - 1) / kScanBlock;
    const int hi = P - jl * K;                     // this lane's segments: hi-1, hi-2, ..., max(hi-K, 0)
    const double* Mg = seg + (PART == 0 ? 0 : 2 * EF);
    const double* Cg = seg + (PART == 0 ? EF : 3 * EF);
    double* bnd = seg + SEG + (PART == 0 ? 0 : EF);

    auto apply = [&](double (&val)[NV], const double (&Mm)[EF], const double (&Cc)[NV]) {
        double t[NV];
        if constexpr (PART == 0) gemm<D>(val, Mm, t); else gemv<D>(Mm, val, t);
#pragma unroll
        for (int e = 0; e < NV; ++e) val[e] = t[e] + Cc[e];
    };
    auto load_map = [&](int p, double (&Mm)[EF], double (&Cc)[NV]) {
        const size_t lane = (size_t)b * P + p;
#pragma unroll
        for (int e = 0; e < EF; ++e) Mm[e] = Mg[(size_t)lane * STR + e];
#pragma unroll
        for (int e = 0; e < NV; ++e) Cc[e] = Cg[(size_t)lane * STR + e];
    };

    // the lane's composed map
    double M[EF], C[NV];
#pragma unroll
    for (int e = 0; e < EF; ++e) M[e] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) M[i * D + i] = 1.0;
#pragma unroll
    for (int e = 0; e < NV; ++e) C[e] = 0.0;
    for (int k = 0; k < K; ++k) {
        const int p = hi - 1 - k;
        if (p >= 0) {
            double Mm[EF], Cc[NV], t[EF];
            load_map(p, Mm, Cc);
            apply(C, Mm, Cc);
            if constexpr (PART == 0) gemm<D>(M, Mm, t); else gemm<D>(Mm, M, t);
#pragma unroll
            for (int e = 0; e < EF; ++e) M[e] = t[e];
        }
    }
    // Inclusive scan of the lanes' composed maps (lane 0 holds the END of the chain) in log2(64) rounds, as k_vdp_marginals_scan:
    //   PART 0, X -> X M + C:  later o earlier = (Mp M, Cp M + C);      PART 1, v -> M v + C:  later o earlier = (M Mp, M Cp + C).
    for (int o = 1; o < 64; o <<= 1) {
        double Mp[EF], Cp[NV];
#pragma unroll
        for (int e = 0; e < EF; ++e) Mp[e] = vdp_up(M[e], o);
#pragma unroll
        for (int e = 0; e < NV; ++e) Cp[e] = vdp_up(C[e], o);
        if (j >= o) {
            double t[EF];
            apply(Cp, M, C);                       // Cp M + C  /  M Cp + C
            if constexpr (PART == 0) gemm<D>(Mp, M, t); else gemm<D>(M, Mp, t);
#pragma unroll
            for (int e = 0; e < EF; ++e) M[e] = t[e];
#pragma unroll
            for (int e = 0; e < NV; ++e) C[e] = Cp[e];
        }
    }
    // value at the last node of the chain: psi_{N-1} = 1e-10 I, lambda_{N-1} = 0; what enters lane j's segments is the composition of the
    // lanes before it applied to that value
    if (j == 63) {
#pragma unroll
        for (int e = 0; e < EF; ++e) wtot[wv][e] = M[e];
#pragma unroll
        for (int e = 0; e < NV; ++e) wtot[wv][EF + e] = C[e];
    }
    __syncthreads();
    double mine[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) mine[e] = 0.0;
    if constexpr (PART == 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) mine[i * D + i] = 1e-10;
    }
    for (int w = 0; w < wv; ++w) {                 // through the wavefronts before this one (they hold the later segments)
        double Mw[EF], Cw[NV];
#pragma unroll
        for (int e = 0; e < EF; ++e) Mw[e] = wtot[w][e];
#pragma unroll
        for (int e = 0; e < NV; ++e) Cw[e] = wtot[w][EF + e];
        apply(mine, Mw, Cw);
    }
    {
        double Me[EF], Ce[NV];
#pragma unroll
        for (int e = 0; e < EF; ++e) Me[e] = vdp_up(M[e], 1);
#pragma unroll
        for (int e = 0; e < NV; ++e) Ce[e] = vdp_up(C[e], 1);
        if (j > 0) apply(mine, Me, Ce);
    }
    // replay: boundary value of every segment of this lane
    for (int k = 0; k < K; ++k) {
        const int p = hi - 1 - k;
        if (p >= 0) {
            const size_t lane = (size_t)b * P + p;
#pragma unroll
            for (int e = 0; e < NV; ++e) bnd[(size_t)lane * STR + e] = mine[e];
            double Mm[EF], Cc[NV];
            load_map(p, Mm, Cc);
            apply(mine, Mm, Cc);
        }
    }
}
// the psi and the lambda recurrence are independent: one launch, blockIdx.y picks the part (the lambda scan, 0.04 ms at config 3, runs
// under the psi scan, 0.08 ms, instead of after it)
template <int D>
__global__ __launch_bounds__(kScanBlock) void k_vdp_lagrange_scan_wave(LevelDesc lv, double* __restrict__ seg) {
    if (blockIdx.y == 0) vdp_lagrange_scan_wave_body<D, 0>(lv, seg);
    else vdp_lagrange_scan_wave_body<D, 1>(lv, seg);
}

// ---- update_param ---------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(64) void k_vdp_update_param(LevelDesc lv, VdpParams pr, const double* __restrict__ mug,
                                                        const double* __restrict__ Sigg, double* __restrict__ psig,
                                                        double* __restrict__ lamg, double* __restrict__ Am,
                                                        double* __restrict__ bm) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    (void)b;
    for (int s = 0; s < R; ++s) {
        if (s < len && p * R + s + 1 < n) {
            double m[D], S[ET], psi[EF], lam[D], A[EF], bb[D];
            ld_node<D>(mug, R, s, me, m);
            ld_node<ET>(Sigg, R, s, me, S);
            ld_node<EF>(psig, R, s, me, psi);
            ld_node<D>(lamg, R, s, me, lam);
            ld_node<EF>(Am, R, s, me, A);
            ld_node<D>(bm, R, s, me, bb);
            if (pr.clip > 0.0) {
                // stabilize_system (vi_sde.py:393-397): the multipliers are scrubbed and clipped in place before they are used
#pragma unroll
                for (int e = 0; e < EF; ++e) psi[e] = vdp_stab(psi[e], pr.clip);
#pragma unroll
                for (int i = 0; i < D; ++i) lam[i] = vdp_stab(lam[i], pr.clip);
                st_node<EF>(psig, R, s, me, psi);
                st_node<D>(lamg, R, s, me, lam);
            }
            double Ef[D], Jf[D], Vf[D], t0[D], t1[D], t2[D], t3[D], t4[D];
            drift_moments<D>(pr, m, S, Ef, Jf, Vf, t0, t1, t2, t3, t4);
            double At[EF], bt[D];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) At[i * D + j] = 2.0 * pr.q[i] * psi[i * D + j] - (i == j ? Jf[i] : 0.0);
            gemv<D>(At, m, bt);
#pragma unroll
            for (int i = 0; i < D; ++i) bt[i] += Ef[i] - pr.q[i] * lam[i];
#pragma unroll
            for (int e = 0; e < EF; ++e) A[e] = (1.0 - pr.lr) * A[e] + pr.lr * At[e];
#pragma unroll
            for (int i = 0; i < D; ++i) bb[i] = (1.0 - pr.lr) * bb[i] + pr.lr * bt[i];
            st_node<EF>(Am, R, s, me, A);
            st_node<D>(bm, R, s, me, bb);
        }
    }
}

}  // namespace mfgm
