// VDP (Archambeau et al.) kernels on packed arrays.
#include "mfgm_internal.h"
#include "mfgm_sweeps.h"
#include "mfgm_vdp.h"
#include "mfgm_band.h"

using namespace mfgm;

static_assert(sizeof(mfgm_vdp_params) == sizeof(mfgm::VdpParams), "public and internal VDP parameter structs must match");

namespace {
template <int D>
int vdp_impl(int what, const Plan& P, const VdpParams& pr, const double* a0, const double* a1, const double* a2, const double* a3,
             const double* a4, const double* a5, double* o0, double* o1, double* o2, double* ws, hipStream_t st,
             const int* obs_count = nullptr, const double* dobs_const = nullptr, const double* x0 = nullptr, const double* x1 = nullptr) {
    const LevelDesc& lv = P.lv[0];
    dim3 grid(lv.Lpad / 64), block(64);
    if (what == 0) {
        hipLaunchKernelGGL((k_vdp_to_ssm<D>), grid, block, 0, st, lv, pr, a0, a1, o0, o1, o2);
    } else if (what == 1) {
        double* part = ws + P.off_part[0];
        if (o1) hipLaunchKernelGGL((k_vdp_esde<D, true>), grid, block, 0, st, lv, pr, a0, a1, a2, a3, part, o1, o2);
        else hipLaunchKernelGGL((k_vdp_esde<D, false>), grid, block, 0, st, lv, pr, a0, a1, a2, a3, part, o1, o2);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, part, lv.P, 0, o0, (double*)nullptr);
    } else if (what == 2 || what == 5 || what == 8 || what == 10 || what == 11) {
        // segment summaries, per-chain scan of the segment maps, final sweep (what == 5: the sweep also makes update_param)
        double* Aw = const_cast<double*>(a2);
        double* bw = const_cast<double*>(a3);
        // what == 10: what == 8 with the linear parts of the segment maps already in o2 (k_vdp_marginals<D, 1, 1>);
        // what == 11: with the affine offsets as well (k_vdp_marginals<D, 3, 2>): the call starts at the segment scan
        if (what != 10 && what != 11) {
            hipLaunchKernelGGL((k_vdp_lagrange_products<D>), grid, block, 0, st, lv, pr, a2, o2);
            MFGM_CHECK_LAUNCH();
        }
        if (what != 11) {
            hipLaunchKernelGGL((k_vdp_lagrange<D, 1>), grid, block, 0, st, lv, pr, a0, a1, Aw, bw, a4, a5, o0, o1, o2, obs_count, dobs_const);
            MFGM_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL((k_vdp_lagrange_scan_wave<D>), dim3(P.B, 2), dim3(kScanBlock), 0, st, lv, o2);
        MFGM_CHECK_LAUNCH();
        if (what == 2) hipLaunchKernelGGL((k_vdp_lagrange<D, 3>), grid, block, 0, st, lv, pr, a0, a1, Aw, bw, a4, a5, o0, o1, o2, obs_count, dobs_const);
        else if (what == 5) hipLaunchKernelGGL((k_vdp_lagrange<D, 4>), grid, block, 0, st, lv, pr, a0, a1, Aw, bw, a4, a5, o0, o1, o2, obs_count, dobs_const);
        // what == 8: the multipliers of node 0 only (o0 = psi0 [B, d, d], o1 = lam0 [B, d], natural layout)
        else hipLaunchKernelGGL((k_vdp_lagrange<D, 5>), grid, block, 0, st, lv, pr, a0, a1, Aw, bw, a4, a5, o0, o1, o2, obs_count, dobs_const);
    } else if (what == 7 || what == 9) {
        // the final Lagrange sweep with the parameter update alone (roofline timing; the segment scans of a full call must be in o2);
        // what == 9: its node-0-multipliers form
        double* Aw = const_cast<double*>(a2);
        double* bw = const_cast<double*>(a3);
        if (what == 7) hipLaunchKernelGGL((k_vdp_lagrange<D, 4>), grid, block, 0, st, lv, pr, a0, a1, Aw, bw, a4, a5, o0, o1, o2, obs_count, dobs_const);
        else hipLaunchKernelGGL((k_vdp_lagrange<D, 5>), grid, block, 0, st, lv, pr, a0, a1, Aw, bw, a4, a5, o0, o1, o2, obs_count, dobs_const);
    } else if (what == 6) {
        // forward_pass as the partitioned moment recursion: a2 = q0_mu [B, d], a3 = q0_cov [B, ET]; o0 = mu, o1 = Sig, o2 = seg
        // a4 (optional) = E_sde / dt per trajectory [B], then ws holds the per-lane partials; a5 (optional) = the seg array of the
        // Lagrange call that follows on the same (A, b): pass 1 leaves the linear parts of its segment maps there
        double* part = a4 ? ws + P.off_part[0] : nullptr;
        // x0 (optional, with a5) = R^-1 y at the observation nodes (x1 = dobsS or obs_count / dobs_const as in the Lagrange calls): the
        // final sweep then makes that call's first pass too
        // d <= 5: everything rides in the sweeps; d = 6: the offsets pass does not fit next to the final sweep (256 + 256 registers and
        // 612 B of scratch: 1.2 ms against 0.33 + 0.43 apart) and follows it as a launch of its own; d = 7, 8: the LDS accumulators of
        // the products (50 / 64 KB per wavefront) would halve the occupancy of the first pass, so both passes follow.  Same contract.
        const bool lag1 = a5 && (D == 6 || (D < 6 && !x0)), lag2 = a5 && x0 && D <= 5;
        if (lag1) hipLaunchKernelGGL((k_vdp_marginals<D, 1, 1>), grid, block, 0, st, lv, pr, a0, a1, o0, o1, o2, (double*)nullptr, const_cast<double*>(a5));
        else hipLaunchKernelGGL((k_vdp_marginals<D, 1>), grid, block, 0, st, lv, pr, a0, a1, o0, o1, o2, (double*)nullptr);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL((k_vdp_marginals_scan<D>), dim3(P.B), dim3(kScanBlock), 0, st, lv, a2, a3, o2);
        MFGM_CHECK_LAUNCH();
        if (lag2) hipLaunchKernelGGL((k_vdp_marginals<D, 3, 2>), grid, block, 0, st, lv, pr, a0, a1, o0, o1, o2, part, const_cast<double*>(a5), x0, x1,
                                     obs_count, dobs_const);
        else hipLaunchKernelGGL((k_vdp_marginals<D, 3>), grid, block, 0, st, lv, pr, a0, a1, o0, o1, o2, part);
        if (part) {
            MFGM_CHECK_LAUNCH();
            hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(256), 0, st, part, lv.P, 0, const_cast<double*>(a4), (double*)nullptr);
        }
        if (a5 && D > 6) {
            MFGM_CHECK_LAUNCH();
            hipLaunchKernelGGL((k_vdp_lagrange_products<D>), grid, block, 0, st, lv, pr, a0, const_cast<double*>(a5));
        }
        if (a5 && x0 && !lag2) {
            MFGM_CHECK_LAUNCH();
            hipLaunchKernelGGL((k_vdp_lagrange<D, 1>), grid, block, 0, st, lv, pr, o0, o1, const_cast<double*>(a0), const_cast<double*>(a1), x0, x1,
                               (double*)nullptr, (double*)nullptr, const_cast<double*>(a5), obs_count, dobs_const);
        }
    } else if (what == 4) {
        hipLaunchKernelGGL((k_vdp_to_naturals<D>), grid, block, 0, st, lv, pr, a0, a1, a2, a3, o0, o1, o2);
    } else {
        hipLaunchKernelGGL((k_vdp_update_param<D>), grid, block, 0, st, lv, pr, a0, a1, const_cast<double*>(a2), const_cast<double*>(a3), o0, o1);
    }
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

namespace {
// one recurrence (n = 1) or two independent ones in the same launches (n = 2; each with its own seg array)
template <int D>
int congruence_scan_impl(const Plan& P, int n, const double* const* Phi, const double* const* Q, double* const* X, double* const* seg, hipStream_t st) {
    constexpr int ET = MFGM_NTRI(D), STR = 2 * (D + ET) + D * D;
    const LevelDesc& lv = P.lv[0];
    double* zeros = seg[0] + (size_t)STR * lv.Lpad;          // q0 = (0, 0) of every chain
    if (hipMemsetAsync(zeros, 0, (size_t)P.B * (D + ET) * sizeof(double), st) != hipSuccess) return 3;
    ScanSets sets;
    for (int k = 0; k < 2; ++k) sets.s[k] = ScanSet{Phi[k % n], Q[k % n], X[k % n], seg[k % n]};
    dim3 grid(lv.Lpad / 64, n), block(64);
    hipLaunchKernelGGL((k_congruence_scan<D, 1>), grid, block, 0, st, lv, sets);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_vdp_marginals_scan<D>), dim3(P.B, n), dim3(kScanBlock), 0, st, lv, (const double*)zeros, (const double*)(zeros + (size_t)P.B * D),
                       seg[0], seg[n - 1]);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_congruence_scan<D, 3>), grid, block, 0, st, lv, sets);
    MFGM_CHECK_LAUNCH();
    return 0;
}
template <int D>
int congruence_scan_impl(const Plan& P, const double* Phi, const double* Q, double* X, double* seg, hipStream_t st) {
    return congruence_scan_impl<D>(P, 1, &Phi, &Q, &X, &seg, st);
}
}  // namespace

namespace {
size_t scan_ws_doubles(const Plan& P) {
    const int d = P.d, et = d * (d + 1) / 2;
    return ((size_t)(2 * (d + et) + d * d) * P.lv[0].Lpad + (size_t)P.B * (d + et) + 63) / 64 * 64;
}
template <int D>
int band_impl(const Plan& P, const double* Sig, const double* Sub, const double* dPd, const double* dPs, double* Xd, double* Xs,
              double* work, hipStream_t st) {
    constexpr int ET = MFGM_NTRI(D), EF = D * D;
    const LevelDesc& lv = P.lv[0];
    const size_t nF = packed_elems(lv, EF), nS = packed_elems(lv, ET);
    BandArgs a;
    a.Sig = Sig; a.Sub = Sub; a.dPd = dPd; a.dPs = dPs; a.Xd = Xd; a.Xs = Xs;
    double* w = work;
    a.PhiL = w; w += nF;
    a.PhiR = w; w += nF;
    a.QL = w; w += nS;
    a.QR = w; w += nS;
    a.loc = w; w += nS;
    double* Lr = w; w += nS;
    double* Rr = w; w += nS;
    double* seg = w;
    a.Lr = Lr; a.Rr = Rr;
    dim3 grid(lv.Lpad / 64), block(64);
    hipLaunchKernelGGL((k_band_prepare<D>), grid, block, 0, st, lv, a);
    MFGM_CHECK_LAUNCH();
    // the ascending and the descending recurrence are independent: same launches, own seg arrays
    const double* Phis[2] = {a.PhiL, a.PhiR};
    const double* Qs[2] = {a.QL, a.QR};
    double* Xs2[2] = {Lr, Rr};
    double* segs[2] = {seg, seg + scan_ws_doubles(P)};
    int rc = congruence_scan_impl<D>(P, 2, Phis, Qs, Xs2, segs, st);
    if (rc) return rc;
    hipLaunchKernelGGL((k_band_finish<D>), grid, block, 0, st, lv, a);
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

size_t mfgm_congruence_scan_workspace_doubles(const mfgm_plan* plan) {
    if (!plan || plan->p.wide) return 0;
    const int d = plan->p.d, et = d * (d + 1) / 2;
    return (size_t)(2 * (d + et) + d * d) * plan->p.lv[0].Lpad + (size_t)plan->p.B * (d + et);
}

int mfgm_congruence_scan(const mfgm_plan* plan, const double* Phi, const double* Q, double* X, double* seg, void* stream) {
    if (!plan || !Phi || !Q || !X || !seg || plan->p.wide) return 1;
    const Plan& P = plan->p;
    MFGM_DISPATCH_D(P.d, (congruence_scan_impl<DD>(P, Phi, Q, X, seg, (hipStream_t)stream)));
}

size_t mfgm_band_workspace_doubles(const mfgm_plan* plan) {
    if (!plan || plan->p.wide) return 0;
    const Plan& P = plan->p;
    const int d = P.d;
    return 2 * packed_elems(P.lv[0], d * d) + 5 * packed_elems(P.lv[0], d * (d + 1) / 2) + 2 * scan_ws_doubles(P);
}

int mfgm_band_sigma_dP_sigma(const mfgm_plan* plan, const double* Sig, const double* Sub, const double* dPd, const double* dPs, double* Xd,
                             double* Xs, double* work, void* stream) {
    if (!plan || !Sig || !Sub || !dPd || !dPs || !Xd || !Xs || !work || plan->p.wide) return 1;
    const Plan& P = plan->p;
    MFGM_DISPATCH_D(P.d, (band_impl<DD>(P, Sig, Sub, dPd, dPs, Xd, Xs, work, (hipStream_t)stream)));
}

size_t mfgm_vdp_workspace_doubles(const mfgm_plan* plan) {
    if (!plan) return 0;
    const int d = plan->p.d;
    return (size_t)(4 * d * d + 2 * d) * plan->p.lv[0].Lpad;
}

int mfgm_packed_vdp_to_ssm(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* Am, const double* bm, double* A,
                           double* off, double* chol, void* stream) {
    if (!plan || !prm || !Am || !bm || !A || !off || !chol) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(0, P, pr, Am, bm, nullptr, nullptr, nullptr, nullptr, A, off, chol, nullptr, st)));
}

int mfgm_packed_vdp_to_naturals(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* Am, const double* bm,
                                const double* p0inv, const double* p0lin, double* lin, double* diag, double* sub, void* stream) {
    if (!plan || !prm || !Am || !bm || !p0inv || !p0lin || !lin || !diag || !sub) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(4, P, pr, Am, bm, p0inv, p0lin, nullptr, nullptr, lin, diag, sub, nullptr, st)));
}

int mfgm_packed_vdp_marginals(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* Am, const double* bm,
                              const double* q0_mu, const double* q0_cov, double* mu, double* Sig, double* e_over_dt, double* seg,
                              void* ws, void* stream) {
    if (!plan || !prm || !Am || !bm || !q0_mu || !q0_cov || !mu || !Sig || !seg || (e_over_dt && !ws)) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(6, P, pr, Am, bm, q0_mu, q0_cov, e_over_dt, nullptr, mu, Sig, seg, (double*)ws, st)));
}

int mfgm_packed_vdp_marginals_products(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* Am, const double* bm,
                                       const double* q0_mu, const double* q0_cov, double* mu, double* Sig, double* e_over_dt, double* seg,
                                       double* lagrange_seg, const double* yR, const double* dobsS, const int* obs_count,
                                       const double* dobs_const, void* ws, void* stream) {
    if (!plan || !prm || !Am || !bm || !q0_mu || !q0_cov || !mu || !Sig || !seg || !lagrange_seg || lagrange_seg == seg || (e_over_dt && !ws))
        return 1;
    if ((obs_count != nullptr) != (dobs_const != nullptr) || (yR && !obs_count && !dobsS) || (!yR && (dobsS || obs_count))) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(6, P, pr, Am, bm, q0_mu, q0_cov, e_over_dt, lagrange_seg, mu, Sig, seg, (double*)ws, st, obs_count, dobs_const,
                                       yR, dobsS)));
}

int mfgm_packed_vdp_esde(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig, const double* Am,
                         const double* bm, double* e_over_dt, double* gm, double* gS, void* ws, void* stream) {
    if (!plan || !prm || !mu || !Sig || !Am || !bm || !e_over_dt || !ws || ((gm != nullptr) != (gS != nullptr))) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(1, P, pr, mu, Sig, Am, bm, nullptr, nullptr, e_over_dt, gm, gS, (double*)ws, st)));
}

int mfgm_packed_vdp_lagrange(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig,
                             const double* Am, const double* bm, const double* yR, const double* dobsS, double* psi, double* lam,
                             double* seg, const int* obs_count, const double* dobs_const, void* stream) {
    if (!plan || !prm || !mu || !Sig || !Am || !bm || !yR || !psi || !lam || !seg) return 1;
    if ((obs_count != nullptr) != (dobs_const != nullptr) || (!obs_count && !dobsS)) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(2, P, pr, mu, Sig, Am, bm, yR, dobsS, psi, lam, seg, nullptr, st, obs_count, dobs_const)));
}

int mfgm_packed_vdp_lagrange_update(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig, double* Am,
                                    double* bm, const double* yR, const double* dobsS, double* psi, double* lam, double* seg,
                                    const int* obs_count, const double* dobs_const, void* stream) {
    if (!plan || !prm || !mu || !Sig || !Am || !bm || !yR || !psi || !lam || !seg) return 1;
    if ((obs_count != nullptr) != (dobs_const != nullptr) || (!obs_count && !dobsS)) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(5, P, pr, mu, Sig, Am, bm, yR, dobsS, psi, lam, seg, nullptr, st, obs_count, dobs_const)));
}

int mfgm_packed_vdp_lagrange_update_final(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig,
                                          double* Am, double* bm, const double* yR, const double* dobsS, double* psi, double* lam,
                                          double* seg, const int* obs_count, const double* dobs_const, void* stream) {
    if (!plan || !prm || !mu || !Sig || !Am || !bm || !yR || !psi || !lam || !seg) return 1;
    if ((obs_count != nullptr) != (dobs_const != nullptr) || (!obs_count && !dobsS)) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(7, P, pr, mu, Sig, Am, bm, yR, dobsS, psi, lam, seg, nullptr, st, obs_count, dobs_const)));
}

int mfgm_packed_vdp_lagrange_update0(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig, double* Am,
                                     double* bm, const double* yR, const double* dobsS, double* psi0, double* lam0, double* seg,
                                     const int* obs_count, const double* dobs_const, int mode, void* stream) {
    if (!plan || !prm || !mu || !Sig || !Am || !bm || !yR || !psi0 || !lam0 || !seg || mode < 0 || mode > 3) return 1;
    if ((obs_count != nullptr) != (dobs_const != nullptr) || (!obs_count && !dobsS)) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(mode == 1 ? 9 : (mode == 2 ? 10 : (mode == 3 ? 11 : 8)), P, pr, mu, Sig, Am, bm, yR, dobsS, psi0, lam0, seg, nullptr, st, obs_count, dobs_const)));
}

int mfgm_packed_vdp_update_param(const mfgm_plan* plan, const mfgm_vdp_params* prm, const double* mu, const double* Sig,
                                 const double* psi, const double* lam, double* Am, double* bm, void* stream) {
    if (!plan || !prm || !mu || !Sig || !psi || !lam || !Am || !bm) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(3, P, pr, mu, Sig, psi, lam, nullptr, nullptr, Am, bm, nullptr, nullptr, st)));
}

}  // extern "C"

