// Per-node ("local") kernels on the packed layout: everything on the path that is embarrassingly parallel in
// time but needs the neighbouring node (t+1).  One lane walks its segment in order and carries the
// neighbour in registers, so every tensor is still read exactly once, coalesced.
#pragma once
#include "mfgm_layout.h"
#include "mfgm_math.h"
#include "mfgm_sweeps.h"

namespace mfgm {

// node (lane, s+1) of the same chain: the next step of this segment, or step 0 of the next segment.
// Caller guarantees the global node t+1 exists.
template <int E>
MFGM_DEV void ld_next(const double* __restrict__ base, int R, int s, int len, int lane, LaneRef me, double (&out)[E]) {
    if (s + 1 < len) ld_node<E>(base, R, s + 1, me, out);
    else ld_node<E>(base, R, 0, LaneRef::of(lane + 1), out);
}

// ---- out = a*x + b*y + c*z on flat arrays (y, z optional) ------------------------------------------------
static __global__ __launch_bounds__(256) void k_lincomb(size_t n, double* out, double a, const double* x,
                                                double b, const double* y, double c,
                                                const double* z) {
    const size_t n2 = n / 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 v = reinterpret_cast<const double2*>(x)[i];
        v.x *= a; v.y *= a;
        if (y) { double2 w = reinterpret_cast<const double2*>(y)[i]; v.x = __builtin_fma(b, w.x, v.x); v.y = __builtin_fma(b, w.y, v.y); }
        if (z) { double2 w = reinterpret_cast<const double2*>(z)[i]; v.x = __builtin_fma(c, w.x, v.x); v.y = __builtin_fma(c, w.y, v.y); }
        reinterpret_cast<double2*>(out)[i] = v;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double v = a * x[n - 1];
        if (y) v += b * y[n - 1];
        if (z) v += c * z[n - 1];
        out[n - 1] = v;
    }
}

// ---- gather / scatter of a sparse list of nodes (observation times) --------------------------------------
// node_ids[i] = b*T + t.  values: natural [n, E_nat].  mode 0: packed -> values, 1: values -> packed (set),
// 2: packed += scale*values (and packed2 += scale*values when given).  SYM scatters read the lower triangle.
// one element of one listed node (shared by the single-array and the paired kernels)
MFGM_DEV void node_io_elem(const LevelDesc& lv, int T, int d, int kind, double* packed, double* packed2,
                           const long long* __restrict__ node_ids, double* values, int mode, double scale, unsigned idx) {
    const unsigned En = (kind == 0) ? d : d * d;
    const unsigned i = idx / En, ne = idx - i * En;
    // the host guarantees B * T < 2^32: 32-bit division (a 64-bit one costs more than the memory access it addresses)
    const unsigned id = (unsigned)node_ids[i];
    const unsigned b = id / (unsigned)T, t = id - b * (unsigned)T;
    const unsigned p = t / lv.R, s = t - p * lv.R, lane = b * lv.P + p;
    unsigned Ep = En, e = ne;
    bool skip = false, zero = false;
    if (kind >= 2) {
        Ep = d * (d + 1) / 2;
        const unsigned r = ne / d, c = ne - r * d;
        if (mode == 0) {
            if (kind == 3 && c > r) zero = true;
            e = six(r, c);
        } else {
            if (c > r) skip = true;
            e = tix(r, c > r ? r : c);
        }
    }
    if (skip) return;
    const size_t off = (((size_t)(lane >> 6) * lv.R + s) * Ep + e) * 64 + (lane & 63);
    if (mode == 0) values[idx] = zero ? 0.0 : packed[off];
    else if (mode == 1) packed[off] = values[idx];
    else {
        const double v = scale * values[idx];
        packed[off] += v;
        if (packed2) packed2[off] += v;
    }
}

static __global__ __launch_bounds__(256) void k_node_io(LevelDesc lv, int T, int d, int kind, double* packed, double* packed2,
                                                const long long* __restrict__ node_ids, int n, double* values, int mode,
                                                double scale) {
    const unsigned total = (unsigned)n * ((kind == 0) ? d : d * d);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
        node_io_elem(lv, T, d, kind, packed, packed2, node_ids, values, mode, scale, idx);
}

// a vector array and a symmetric array at the same nodes in ONE launch (the site updates always move both; the two scattered
// access streams then overlap instead of paying their latency one after the other)
static __global__ __launch_bounds__(256) void k_node_io_pair(LevelDesc lv, int T, int d, double* packed_vec, double* packed_sym,
                                                             const long long* __restrict__ node_ids, int n, double* values_vec,
                                                             double* values_sym, int mode, double scale) {
    const unsigned nv = (unsigned)n * d, total = nv + (unsigned)n * d * d;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        if (idx < nv) node_io_elem(lv, T, d, 0, packed_vec, nullptr, node_ids, values_vec, mode, scale, idx);
        else node_io_elem(lv, T, d, 2, packed_sym, nullptr, node_ids, values_sym, mode, scale, idx - nv);
    }
}

// ---- CVI site update at the listed nodes (variational_cvi_sde.py:301-317) --------------------------------------------------------
// sites <- (1 - lr) sites + lr g  and  packed += (new - old), for the linear part (natural [n, d]) and the full d x d diagonal-block
// part (natural [n, d, d], symmetric; the packed array holds its lower triangle): the blend, the difference, the copy back and the
// scatter of update_data_sites in one pass over the site arrays.
static __global__ __launch_bounds__(256) void k_site_update_pair(LevelDesc lv, int T, int d, double* packed_vec, double* packed_sym,
                                                                 const long long* __restrict__ node_ids, int n, double* sites_vec,
                                                                 double* sites_sym, const double* __restrict__ g_vec,
                                                                 const double* __restrict__ g_sym, double lr) {
    const unsigned nv = (unsigned)n * d, total = nv + (unsigned)n * d * d;
    const unsigned Et = d * (d + 1) / 2;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const bool vec = idx < nv;
        const unsigned k = vec ? idx : idx - nv, En = vec ? d : d * d;
        const unsigned i = k / En, ne = k - i * En;
        double* sites = vec ? sites_vec : sites_sym;
        const double old = sites[k], g = vec ? g_vec[k] : g_sym[k];
        const double nw = (1.0 - lr) * old + lr * g;
        sites[k] = nw;
        unsigned Ep = d, e = ne;
        if (!vec) {
            const unsigned r = ne / d, c = ne - r * d;
            if (c > r) continue;                      // the packed block is the lower triangle
            Ep = Et;
            e = tix(r, c);
        }
        const unsigned id = (unsigned)node_ids[i];
        const unsigned b = id / (unsigned)T, t = id - b * (unsigned)T;
        const unsigned p = t / lv.R, s = t - p * lv.R, lane = b * lv.P + p;
        const size_t off = (((size_t)(lane >> 6) * lv.R + s) * Ep + e) * 64 + (lane & 63);
        double* packed = vec ? packed_vec : packed_sym;
        packed[off] += nw - old;
    }
}

// ---- variational expectations of a multivariate Gaussian likelihood at the observation nodes -------------------------------------
// ve[b] = sum_i { -1/2 tr(Sinv Sigma_i) - 1/2 (y_i - mu_i)^T Sinv (y_i - mu_i) } + n cst   (multivariate_gaussian.py:80-115), the
// marginals gathered from the packed arrays (and written to out_mu [n, d], out_cov [n, d, d] for the callers that want them):
// gather_nodes_pair, the element-wise VE arithmetic and the sum in one launch.  Block (x, b) sums observations x*256 ... of trajectory b
// into ve[b * gridDim.x + x] (fixed order; the caller adds the few partials of a trajectory).  node_ids / y are trajectory-major with
// n_per observations each.
template <int D>
static __global__ __launch_bounds__(256) void k_mvn_obs_ve(LevelDesc lv, int T, const double* __restrict__ mug,
                                                          const double* __restrict__ Sigg, const long long* __restrict__ node_ids,
                                                          int n_per, const double* __restrict__ y, const double* __restrict__ Sinv,
                                                          double cst, double* __restrict__ out_mu, double* __restrict__ out_cov,
                                                          double* __restrict__ ve) {
    constexpr int ET = MFGM_NTRI(D);
    __shared__ double sS[D * D];
    __shared__ double red[256];
    for (int e = threadIdx.x; e < D * D; e += blockDim.x) sS[e] = Sinv[e];
    __syncthreads();
    const int b = blockIdx.y;
    double acc = 0.0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_per; j += gridDim.x * blockDim.x) {
        const size_t i = (size_t)b * n_per + j;
        const unsigned id = (unsigned)node_ids[i];
        const unsigned cb = id / (unsigned)T, t = id - cb * (unsigned)T;
        const unsigned p = t / lv.R, s = t - p * lv.R, lane = cb * lv.P + p;
        const size_t nb = (size_t)(lane >> 6) * lv.R + s;       // node block: element e of a kind with E doubles at (nb * E + e) * 64 + lane % 64
        double m[D], S[ET];
#pragma unroll
        for (int e = 0; e < D; ++e) m[e] = mug[(nb * D + e) * 64 + (lane & 63)];
#pragma unroll
        for (int e = 0; e < ET; ++e) S[e] = Sigg[(nb * ET + e) * 64 + (lane & 63)];
        double diff[D], v = 0.0;
#pragma unroll
        for (int r = 0; r < D; ++r) diff[r] = y[i * D + r] - m[r];
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int c = 0; c < D; ++c) v += sS[r * D + c] * (S[six(r, c)] + diff[r] * diff[c]);
        acc += -0.5 * v + cst;
        if (out_mu) {
#pragma unroll
            for (int e = 0; e < D; ++e) out_mu[i * D + e] = m[e];
        }
        if (out_cov) {
#pragma unroll
            for (int r = 0; r < D; ++r)
#pragma unroll
                for (int c = 0; c < D; ++c) out_cov[(i * D + r) * D + c] = S[six(r, c)];
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ve[(size_t)b * gridDim.x + blockIdx.x] = red[0];
}

// ---- SSM parameters -> natural parameters / precision blocks ---------------------------------------------
// Per node t:  A (FULL, transition t -> t+1, unused at the last node), off (VEC: mu0 at node 0, b_{t-1} after),
// chol (TRI: chol P0 at node 0, chol Q_{t-1} after).  Writes
//     lin_t  = Qi_t off_t - A_t^T Qi_{t+1} off_{t+1}
//     diag_t = cD * (Qi_t + A_t^T Qi_{t+1} A_t)        (cD = -1/2: naturals, cD = 1: precision)
//     sub_t  = cS * Qi_{t+1} A_t                        (cS = +1: naturals,  cS = -1: precision)
// restating ssm_to_naturals (ssm_gaussian_transformations.py:182-253) and _build_precision
// (state_space_model.py:431-483) with the same operation order (L^{-1}A, then L^{-T}, then the Gram product).
template <int D, bool WANT_LIN>
static __global__ __launch_bounds__(64) void k_ssm_to_naturals(LevelDesc lv, const double* __restrict__ Ag,
                                                       const double* __restrict__ offg, const double* __restrict__ cholg,
                                                       double cD, double cS, double* __restrict__ ling,
                                                       double* __restrict__ diagg, double* __restrict__ subg,
                                                       double* __restrict__ part_logdet) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, Lp = lv.Lpad, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    (void)b;
    // X = chol^{-1}; Qi = X^T X; z = Qi off
    auto prep = [&](const double (&C)[ET], const double (&o)[D], double (&X)[ET], double (&z)[D]) {
        double invd[D];
#pragma unroll
        for (int j = 0; j < D; ++j) invd[j] = rcp_nr(C[tix(j, j)]);
        tri_inverse<D>(C, invd, X);
        if (WANT_LIN) {
#pragma unroll
            for (int e = 0; e < D; ++e) z[e] = o[e];
            trsv_lower<D>(C, invd, z);
            trsv_lower_t<D>(C, invd, z);
        }
    };
    double Xc[ET], zc[D];
    LogAcc la;
    la.init();
    {
        double C[ET], o[D];
        ld_node<ET>(cholg, R, 0, me, C);
        if (WANT_LIN) ld_node<D>(offg, R, 0, me, o);
        prep(C, o, Xc, zc);
#pragma unroll
        for (int j = 0; j < D; ++j) la.mul(C[tix(j, j)]);
        la.renorm();
    }
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            const bool has_next = (p * R + s + 1 < n);
            double Qi[ET], lin[D];
            tri_t_tri<D>(Xc, Qi);
#pragma unroll
            for (int e = 0; e < D; ++e) lin[e] = WANT_LIN ? zc[e] : 0.0;
            if (has_next) {
                double C[ET], o[D], A[EF], Xn[ET], zn[D];
                ld_next<ET>(cholg, R, s, len, lane, me, C);
                if (WANT_LIN) ld_next<D>(offg, R, s, len, lane, me, o);
                ld_node<EF>(Ag, R, s, me, A);
                prep(C, o, Xn, zn);
                if (s + 1 < len) {   // the next node belongs to this lane: account its log-det here
#pragma unroll
                    for (int j = 0; j < D; ++j) la.mul(C[tix(j, j)]);
                    la.renorm();
                }
                // M = Xn A  (lower-tri times full)
                double M[EF];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k <= i; ++k) t = __builtin_fma(Xn[tix(i, k)], A[k * D + j], t);
                        M[i * D + j] = t;
                    }
                // sub = Xn^T M
                double Sb[EF];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double t = 0.0;
#pragma unroll
                        for (int k = i; k < D; ++k) t = __builtin_fma(Xn[tix(k, i)], M[k * D + j], t);
                        Sb[i * D + j] = t;
                    }
                syrk_t_acc<D>(M, Qi);   // Qi += M^T M = A^T Qi_{t+1} A
                if (WANT_LIN) {
                    double t[D];
                    gemv_t<D>(A, zn, t);
#pragma unroll
                    for (int e = 0; e < D; ++e) lin[e] -= t[e];
                }
#pragma unroll
                for (int e = 0; e < EF; ++e) Sb[e] *= cS;
                st_node<EF>(subg, R, s, me, Sb);
#pragma unroll
                for (int e = 0; e < ET; ++e) Xc[e] = Xn[e];
#pragma unroll
                for (int e = 0; e < D; ++e) zc[e] = zn[e];
            } else {
                st_node_zero<EF>(subg, R, s, me);
            }
#pragma unroll
            for (int e = 0; e < ET; ++e) Qi[e] *= cD;
            st_node<ET>(diagg, R, s, me, Qi);
            if (WANT_LIN) st_node<D>(ling, R, s, me, lin);
        }
    }
    if (part_logdet) part_logdet[lane] = la.value();   // sum log diag(chol) over this lane's nodes
}

// ---- KL(q || p) local terms -------------------------------------------------------------------------------
// q: marginal covariances Sig (SYM), Sub = Sigma_{t+1,t} (FULL), means mu (VEC);
// p: precision blocks aD*Pd (SYM), aS*Ps (FULL, block (t+1,t)), means mup (VEC).
// Per-lane partials:  part[lane]        = sum_t <P_tt, Sig_t> + 2 <P_{t+1,t}, Sub_t>      (trace term)
//                     part[Lpad + lane] = sum_t dl_t^T P_tt dl_t + 2 dl_{t+1}^T P_{t+1,t} dl_t   (Mahalanobis term)
// restating StateSpaceModel.kl_divergence (state_space_model.py:557-593).
template <int D>
static __global__ __launch_bounds__(64) void k_kl_terms(LevelDesc lv, const double* __restrict__ Sigg, const double* __restrict__ Subg,
                                                const double* __restrict__ mug, const double* __restrict__ Pdg,
                                                const double* __restrict__ Psg, double aD, double aS,
                                                const double* __restrict__ mupg, double* __restrict__ part) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, Lp = lv.Lpad, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    (void)b;
    double tr = 0.0, mh = 0.0;
    double dl[D];
    {
        double m[D], mp[D];
        ld_node<D>(mug, R, 0, me, m);
        ld_node<D>(mupg, R, 0, me, mp);
#pragma unroll
        for (int e = 0; e < D; ++e) dl[e] = mp[e] - m[e];
    }
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            const bool has_next = (p * R + s + 1 < n);
            double Sg[ET], Pd[ET];
            ld_node<ET>(Sigg, R, s, me, Sg);
            ld_node<ET>(Pdg, R, s, me, Pd);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double w = (i == j) ? 1.0 : 2.0;
                    tr = __builtin_fma(w * aD * Pd[tix(i, j)], Sg[tix(i, j)], tr);
                    mh = __builtin_fma(w * aD * Pd[tix(i, j)] * dl[i], dl[j], mh);
                }
            if (has_next) {
                double Sb[EF], Ps[EF], m[D], mp[D], dn[D];
                ld_node<EF>(Subg, R, s, me, Sb);
                ld_node<EF>(Psg, R, s, me, Ps);
                ld_next<D>(mug, R, s, len, lane, me, m);
                ld_next<D>(mupg, R, s, len, lane, me, mp);
#pragma unroll
                for (int e = 0; e < D; ++e) dn[e] = mp[e] - m[e];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        tr = __builtin_fma(2.0 * aS * Ps[i * D + j], Sb[i * D + j], tr);
                        mh = __builtin_fma(2.0 * aS * Ps[i * D + j] * dn[i], dl[j], mh);
                    }
#pragma unroll
                for (int e = 0; e < D; ++e) dl[e] = dn[e];
            }
        }
    }
    part[lane] = tr;
    part[Lp + lane] = mh;
}

}  // namespace mfgm

namespace mfgm {

// ---- stationary SDE kernels -> packed SSM parameters ---------------------------------------------------------
// A sum of up to 8 stationary components (Matern-1/2, -3/2, -5/2, Ornstein-Uhlenbeck), block-diagonal state.
// Per transition (time step dt):  A = exp(-lam dt) (I + N dt + N^2 dt^2 / 2),  N = F + lam I nilpotent,
//   Q = Pinf - A Pinf A^T + jitter I,  b = (I - A) m,  chol(Q) ("cholesky_or_zero": exactly-zero Q stays zero)
// restating StationaryKernel.transition_statistics / state_offsets (kernels/sde_kernel.py:421-475),
// Matern12/32/52.state_transitions + steady_state_covariance (kernels/matern.py:66-86, 299-324, 434-501),
// Sum / ConcatKernel block-diagonal stacking (sde_kernel.py:592-656) and state_space_model_from_covariances
// (state_space_model.py:613-664).  Node 0 carries (initial mean, chol(Pinf + jitter)).
struct KernelSpec {
    int ncomp;
    int order[8];        // state dimension of the component: 1, 2 or 3
    int offset[8];       // first state index of the component
    double lam[8];       // decay rate lambda
    double var[8];       // variance sigma^2 (Pinf scale)
    double mean[8];      // state mean m (total dim <= 8)
    double jitter;
};

template <int D>
MFGM_DEV void stationary_pinf(const KernelSpec& ks, double (&Pinf)[D * D]) {
#pragma unroll
    for (int e = 0; e < D * D; ++e) Pinf[e] = 0.0;
    for (int c = 0; c < ks.ncomp; ++c) {
        const int o = ks.offset[c];
        const double l = ks.lam[c], v = ks.var[c];
        if (ks.order[c] == 1) {
            Pinf[o * D + o] = v;
        } else if (ks.order[c] == 2) {
            Pinf[o * D + o] = v;
            Pinf[(o + 1) * D + o + 1] = v * l * l;
        } else {
            const double l23 = l * l / 3.0;
            Pinf[o * D + o] = v;
            Pinf[o * D + o + 2] = -v * l23;
            Pinf[(o + 2) * D + o] = -v * l23;
            Pinf[(o + 1) * D + o + 1] = v * l23;
            Pinf[(o + 2) * D + o + 2] = v * l * l * l * l;
        }
    }
}

template <int D>
static __global__ __launch_bounds__(64) void k_stationary_ssm(LevelDesc lv, KernelSpec ks, const double* __restrict__ dts /* natural [B, n-1] */,
                                                      double* __restrict__ Ag, double* __restrict__ offg,
                                                      double* __restrict__ cholg, int* info) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const int lane = blockIdx.x * 64 + threadIdx.x;
    if (lane >= lv.L) return;
    const LaneRef me{(int)blockIdx.x, (int)threadIdx.x};
    const int P = lv.P, R = lv.R, n = lv.n;
    const int b = lane / P, p = lane - b * P;
    const int len = min(R, n - p * R);
    int bad = 0;
    double Pinf[EF];
    stationary_pinf<D>(ks, Pinf);
    for (int s = 0; s < R; ++s) {
        if (s < len) {
            const int t = p * R + s;
            // --- this node's incoming process noise / offset: transition t-1 -> t (or the initial state) ---
            double A[EF];
            auto build_A = [&](double dt) {
#pragma unroll
                for (int e = 0; e < EF; ++e) A[e] = 0.0;
                for (int c = 0; c < ks.ncomp; ++c) {
                    const int o = ks.offset[c];
                    const double l = ks.lam[c], ex = exp(-l * dt);
                    if (ks.order[c] == 1) {
                        A[o * D + o] = ex;
                    } else if (ks.order[c] == 2) {
                        // N = [[l, 1], [-l^2, -l]]
                        A[o * D + o] = ex * (1.0 + l * dt);
                        A[o * D + o + 1] = ex * dt;
                        A[(o + 1) * D + o] = ex * (-l * l * dt);
                        A[(o + 1) * D + o + 1] = ex * (1.0 - l * dt);
                    } else {
                        // N = [[l,1,0],[0,l,1],[-l^3,-3l^2,-2l]],  N^2 = [[l^2,2l,1],[-l^3,-2l^2,-l],[l^4,2l^3,l^2]]
                        const double l2 = l * l, l3 = l2 * l, l4 = l2 * l2, h = 0.5 * dt * dt;
                        A[o * D + o] = ex * (1.0 + l * dt + l2 * h);
                        A[o * D + o + 1] = ex * (dt + 2.0 * l * h);
                        A[o * D + o + 2] = ex * h;
                        A[(o + 1) * D + o] = ex * (-l3 * h);
                        A[(o + 1) * D + o + 1] = ex * (1.0 + l * dt - 2.0 * l2 * h);
                        A[(o + 1) * D + o + 2] = ex * (dt - l * h);
                        A[(o + 2) * D + o] = ex * (-l3 * dt + l4 * h);
                        A[(o + 2) * D + o + 1] = ex * (-3.0 * l2 * dt + 2.0 * l3 * h);
                        A[(o + 2) * D + o + 2] = ex * (1.0 - 2.0 * l * dt + l2 * h);
                    }
                }
            };
            double off[D], Q[ET];
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) off[i] = ks.mean[i];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) Q[tix(i, j)] = Pinf[i * D + j] + (i == j ? ks.jitter : 0.0);
            } else {
                build_A(dts[(size_t)b * (n - 1) + (t - 1)]);
                double AP[EF];
                gemm<D>(A, Pinf, AP);
                bool zero = true;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double o = ks.mean[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) o = __builtin_fma(-A[i * D + k], ks.mean[k], o);
                    off[i] = o;
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        double q = Pinf[i * D + j];
#pragma unroll
                        for (int k = 0; k < D; ++k) q = __builtin_fma(-AP[i * D + k], A[j * D + k], q);
                        q += (i == j ? ks.jitter : 0.0);
                        Q[tix(i, j)] = q;
                        zero = zero && (q == 0.0);
                    }
                }
                if (zero) {
#pragma unroll
                    for (int i = 0; i < D; ++i) Q[tix(i, i)] = 1.0;   // cholesky_or_zero: factor the identity, store zeros
                }
                double invd[D];
                int bd = 0;
                chol_inplace<D>(Q, invd, bd);
                bad |= bd;
                if (zero) {
#pragma unroll
                    for (int e = 0; e < ET; ++e) Q[e] = 0.0;
                }
            }
            if (t == 0) {
                double invd[D];
                int bd = 0;
                chol_inplace<D>(Q, invd, bd);
                bad |= bd;
            }
            st_node<D>(offg, R, s, me, off);
            st_node<ET>(cholg, R, s, me, Q);
            // --- outgoing transition t -> t+1 ---
            if (t + 1 < n) {
                build_A(dts[(size_t)b * (n - 1) + t]);
            } else {
#pragma unroll
                for (int e = 0; e < EF; ++e) A[e] = 0.0;
            }
            st_node<EF>(Ag, R, s, me, A);
        }
    }
    if (bad) atomicMax(info, 1);
}

}  // namespace mfgm
