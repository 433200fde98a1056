// C-ABI entry points (include/mfgm.h): plan construction, re-layout, and the multi-level drivers that
// chain the reduce / forward / backward sweeps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mfgm.h"
#include "mfgm_layout.h"
#include "mfgm_pack.h"
#include "mfgm_sweeps.h"
#include "mfgm_local.h"
#include "mfgm_sde.h"
#include "mfgm_vdp.h"
#include "mfgm_wide.h"
#include "mfgm_batched.h"

using namespace mfgm;

struct mfgm_plan {
    Plan p;
};

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

#define MFGM_CHECK_LAUNCH()                        \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return 3;           \
    } while (0)

void fill_level(LevelDesc& lv, int B, int n, int R) {
    lv.n = n;
    lv.R = R;
    lv.P = ceil_div(n, R);
    lv.L = B * lv.P;
    lv.Lpad = ceil_div(lv.L, 64) * 64;
}

template <int D>
int launch_reduce(const SweepArgs& a, bool has_rhs, bool has_corr, hipStream_t st) {
    dim3 grid(a.lv.Lpad / 64), block(64);
    if (has_rhs) {
        if (has_corr) hipLaunchKernelGGL((k_reduce<D, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_reduce<D, true, false>), grid, block, 0, st, a);
    } else {
        if (has_corr) hipLaunchKernelGGL((k_reduce<D, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_reduce<D, false, false>), grid, block, 0, st, a);
    }
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D>
int launch_forward(const SweepArgs& a, bool has_rhs, bool has_corr, bool has_up, hipStream_t st) {
    dim3 grid(a.lv.Lpad / 64), block(64);
#define FW(R_, C_, U_) hipLaunchKernelGGL((k_forward<D, R_, C_, U_>), grid, block, 0, st, a)
    if (has_rhs) {
        if (has_corr) { if (has_up) FW(true, true, true); else FW(true, true, false); }
        else { if (has_up) FW(true, false, true); else FW(true, false, false); }
    } else {
        if (has_corr) { if (has_up) FW(false, true, true); else FW(false, true, false); }
        else { if (has_up) FW(false, false, true); else FW(false, false, false); }
    }
#undef FW
    MFGM_CHECK_LAUNCH();
    return 0;
}

template <int D>
int launch_backward(const SweepArgs& a, bool has_rhs, bool has_up, bool want_sub, hipStream_t st) {
    dim3 grid(a.lv.Lpad / 64), block(64);
    const bool mom = (a.momg != nullptr);
#define BW(R_, U_, S_, M_) hipLaunchKernelGGL((k_backward<D, R_, U_, S_, M_>), grid, block, 0, st, a)
    if (mom) {
        // the moment array needs the means: has_rhs is guaranteed by the entry point
        if (has_up) { if (want_sub) BW(true, true, true, true); else BW(true, true, false, true); }
        else { if (want_sub) BW(true, false, true, true); else BW(true, false, false, true); }
    } else if (has_rhs) {
        if (has_up) { if (want_sub) BW(true, true, true, false); else BW(true, true, false, false); }
        else { if (want_sub) BW(true, false, true, false); else BW(true, false, false, false); }
    } else {
        if (has_up) { if (want_sub) BW(false, true, true, false); else BW(false, true, false, false); }
        else { if (want_sub) BW(false, false, true, false); else BW(false, false, false, false); }
    }
#undef BW
    MFGM_CHECK_LAUNCH();
    return 0;
}

// fills the coarse-level pointers of `a` for level l from the workspace
void bind_level_inputs(const Plan& P, int l, double* ws, SweepArgs& a) {
    // inputs of level l >= 1 are the reduced system written by reduce(l-1)
    a.Dg = ws + P.off_Dhat[l];
    a.Dcorr = ws + P.off_Rsub[l];
    a.Sg = ws + P.off_S[l];
    a.rg = ws + P.off_rhat[l];
    a.rcorr = ws + P.off_rho[l];
    a.aD = a.aS = a.aR = 1.0;
    a.Lg = ws + P.off_L[l];
    a.Gg = ws + P.off_G[l];
    a.yg = ws + P.off_y[l];
    a.Sigg = ws + P.off_Sig[l];
    a.Subg = nullptr;
    a.mug = ws + P.off_mu[l];
    a.part = nullptr;
}

void bind_up(const Plan& P, int l, double* ws, SweepArgs& a) {
    // coarser level l+1
    a.up = P.lv[l + 1];
    a.uDhat = ws + P.off_Dhat[l + 1];
    a.uRsub = ws + P.off_Rsub[l + 1];
    a.uS = ws + P.off_S[l + 1];
    a.urhat = ws + P.off_rhat[l + 1];
    a.urho = ws + P.off_rho[l + 1];
    a.uL = ws + P.off_L[l + 1];
    a.uy = ws + P.off_y[l + 1];
    a.uSig = ws + P.off_Sig[l + 1];
    a.umu = ws + P.off_mu[l + 1];
}

template <int D>
int factor_impl(const Plan& P, const double* Dg, const double* Sg, const double* rg, double aD, double aS, double aR,
                double* Lg, double* Gg, double* yg, double* logdet, double* quad, double* ws, int* info,
                hipStream_t st, int only_stage = -1, int only_level = -1) {
    const bool has_rhs = (rg != nullptr);
    const int K = P.nlevels - 1;  // top level index (single segment per chain)
    auto make = [&](int l) {
        SweepArgs a;
        memset(&a, 0, sizeof(a));
        a.lv = P.lv[l];
        a.info = info;
        if (l == 0) {
            a.Dg = Dg; a.Sg = Sg; a.rg = rg; a.aD = aD; a.aS = aS; a.aR = aR;
            a.Lg = Lg; a.Gg = Gg; a.yg = yg;
            a.part = (logdet || quad) ? ws + P.off_part[0] : nullptr;
        } else {
            bind_level_inputs(P, l, ws, a);
        }
        if (l < K) bind_up(P, l, ws, a);
        return a;
    };
    for (int l = 0; l < K; ++l) {
        if (only_stage >= 0 && !(only_stage == 0 && only_level == l)) continue;
        SweepArgs a = make(l);
        int rc = launch_reduce<D>(a, has_rhs, l > 0, st);
        if (rc) return rc;
    }
    for (int l = K; l >= 0; --l) {
        if (only_stage >= 0 && !(only_stage == 1 && only_level == l)) continue;
        SweepArgs a = make(l);
        int rc = launch_forward<D>(a, has_rhs, l > 0, l < K, st);
        if (rc) return rc;
    }
    if (only_stage < 0 && (logdet || quad)) {
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, ws + P.off_part[0], P.lv[0].P, P.lv[0].Lpad,
                           logdet, quad);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}

template <int D>
int selinv_impl(const Plan& P, const double* Lg, const double* Gg, const double* yg, double* Sig, double* Sub,
                double* x, double* ws, hipStream_t st, int only_level = -1, double* mom = nullptr) {
    const bool has_rhs = (yg != nullptr);
    const int K = P.nlevels - 1;
    for (int l = K; l >= 0; --l) {
        if (only_level >= 0 && only_level != l) continue;
        SweepArgs a;
        memset(&a, 0, sizeof(a));
        a.lv = P.lv[l];
        if (l == 0) {
            a.Lg = const_cast<double*>(Lg); a.Gg = const_cast<double*>(Gg); a.yg = const_cast<double*>(yg);
            a.Sigg = Sig; a.Subg = Sub; a.mug = x; a.momg = mom;
        } else {
            bind_level_inputs(P, l, ws, a);
        }
        if (l < K) bind_up(P, l, ws, a);
        int rc = launch_backward<D>(a, has_rhs, l < K, l == 0 && Sub != nullptr, st);
        if (rc) return rc;
    }
    return 0;
}

// ---- wide (8 < d <= 32) drivers: same level recursion, one wavefront per segment ------------------------------------
template <int DM>
int wide_launch(int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    dim3 grid(a.lv.L), block(64);
#define KW(K) hipLaunchKernelGGL((K), grid, block, 0, st, a)
    if (which == 0) {
        if (has_rhs) { if (has_corr) KW((kw_reduce<DM, true, true>)); else KW((kw_reduce<DM, true, false>)); }
        else { if (has_corr) KW((kw_reduce<DM, false, true>)); else KW((kw_reduce<DM, false, false>)); }
    } else if (which == 1) {
        if (has_rhs) {
            if (has_corr) { if (has_up) KW((kw_forward<DM, true, true, true>)); else KW((kw_forward<DM, true, true, false>)); }
            else { if (has_up) KW((kw_forward<DM, true, false, true>)); else KW((kw_forward<DM, true, false, false>)); }
        } else {
            if (has_corr) { if (has_up) KW((kw_forward<DM, false, true, true>)); else KW((kw_forward<DM, false, true, false>)); }
            else { if (has_up) KW((kw_forward<DM, false, false, true>)); else KW((kw_forward<DM, false, false, false>)); }
        }
    } else {
        if (has_rhs) {
            if (has_up) { if (want_sub) KW((kw_backward<DM, true, true, true>)); else KW((kw_backward<DM, true, true, false>)); }
            else { if (want_sub) KW((kw_backward<DM, true, false, true>)); else KW((kw_backward<DM, true, false, false>)); }
        } else {
            if (has_up) { if (want_sub) KW((kw_backward<DM, false, true, true>)); else KW((kw_backward<DM, false, true, false>)); }
            else { if (want_sub) KW((kw_backward<DM, false, false, true>)); else KW((kw_backward<DM, false, false, false>)); }
        }
    }
#undef KW
    MFGM_CHECK_LAUNCH();
    return 0;
}

int wide_dispatch(int d, int which, const WideArgs& a, bool has_rhs, bool has_corr, bool has_up, bool want_sub, hipStream_t st) {
    if (d <= 16) return wide_launch<16>(which, a, has_rhs, has_corr, has_up, want_sub, st);
    return wide_launch<32>(which, a, has_rhs, has_corr, has_up, want_sub, st);
}

void wide_bind(const Plan& P, int l, double* ws, WideArgs& a) {
    const int K = P.nlevels - 1;
    a.lv = P.lv[l];
    a.d = P.d;
    if (l > 0) {
        a.Dg = ws + P.off_Dhat[l]; a.Dcorr = ws + P.off_Rsub[l]; a.Sg = ws + P.off_S[l];
        a.rg = ws + P.off_rhat[l]; a.rcorr = ws + P.off_rho[l];
        a.aD = a.aS = a.aR = 1.0;
        a.Lg = ws + P.off_L[l]; a.Gg = ws + P.off_G[l]; a.yg = ws + P.off_y[l];
        a.Sigg = ws + P.off_Sig[l]; a.Subg = nullptr; a.mug = ws + P.off_mu[l];
    }
    if (l < K) {
        a.up = P.lv[l + 1];
        a.uDhat = ws + P.off_Dhat[l + 1]; a.uRsub = ws + P.off_Rsub[l + 1]; a.uS = ws + P.off_S[l + 1];
        a.urhat = ws + P.off_rhat[l + 1]; a.urho = ws + P.off_rho[l + 1];
        a.uL = ws + P.off_L[l + 1]; a.uy = ws + P.off_y[l + 1]; a.uSig = ws + P.off_Sig[l + 1]; a.umu = ws + P.off_mu[l + 1];
    }
}

int wide_factor(const Plan& P, const double* Dg, const double* Sg, const double* rg, double aD, double aS, double aR,
                double* Lg, double* Gg, double* yg, double* logdet, double* quad, double* ws, int* info, hipStream_t st) {
    const bool has_rhs = (rg != nullptr);
    const int K = P.nlevels - 1;
    auto make = [&](int l) {
        WideArgs a;
        memset(&a, 0, sizeof(a));
        a.info = info;
        if (l == 0) {
            a.Dg = Dg; a.Sg = Sg; a.rg = rg; a.aD = aD; a.aS = aS; a.aR = aR;
            a.Lg = Lg; a.Gg = Gg; a.yg = yg;
            a.part = (logdet || quad) ? ws + P.off_part[0] : nullptr;
        }
        wide_bind(P, l, ws, a);
        return a;
    };
    for (int l = 0; l < K; ++l) {
        int rc = wide_dispatch(P.d, 0, make(l), has_rhs, l > 0, true, false, st);
        if (rc) return rc;
    }
    for (int l = K; l >= 0; --l) {
        int rc = wide_dispatch(P.d, 1, make(l), has_rhs, l > 0, l < K, false, st);
        if (rc) return rc;
    }
    if (logdet || quad) {
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, ws + P.off_part[0], P.lv[0].P, P.lv[0].Lpad, logdet, quad);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}

int wide_selinv(const Plan& P, const double* Lg, const double* Gg, const double* yg, double* Sig, double* Sub, double* x,
                double* ws, hipStream_t st) {
    const bool has_rhs = (yg != nullptr);
    const int K = P.nlevels - 1;
    for (int l = K; l >= 0; --l) {
        WideArgs a;
        memset(&a, 0, sizeof(a));
        if (l == 0) {
            a.Lg = const_cast<double*>(Lg); a.Gg = const_cast<double*>(Gg); a.yg = const_cast<double*>(yg);
            a.Sigg = Sig; a.Subg = Sub; a.mug = x;
        }
        wide_bind(P, l, ws, a);
        int rc = wide_dispatch(P.d, 2, a, has_rhs, false, l < K, l == 0 && Sub != nullptr, st);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

const char* mfgm_version(void) { return "mfgm 0.1 (gfx950)"; }

int mfgm_plan_create(int B, int T, int d, int R0, int Rup, mfgm_plan** out) {
    if (!out || B < 1 || T < 1 || d < 1 || d > 32) return 1;
    mfgm_plan* h = new mfgm_plan();
    Plan& P = h->p;
    memset(&P, 0, sizeof(P));
    P.B = B; P.T = T; P.d = d;
    P.wide = (d > 8);
    if (Rup <= 1) Rup = 8;   // measured best on MI355X for the coarse levels (tools/sweep_partition.sh)
    if (const char* e = getenv("MFGM_RUP")) { int v = atoi(e); if (v > 1) Rup = v; }
    if (R0 <= 0) {
        if (const char* e = getenv("MFGM_R0")) R0 = atoi(e);
    }
    if (R0 <= 0) {
        // narrow: one lane per segment, ~ one wavefront per SIMD on 256 CUs; wide: one wavefront per segment
        const long long target = P.wide ? 8192 : 65536;
        long long r = ((long long)B * T + target - 1) / target;
        R0 = (int)std::min<long long>(std::max<long long>(r, 8), 1 << 20);
    }
    int n = T, l = 0;
    const int top = 48;  // chains this short are swept by one lane
    while (true) {
        int R = (l == 0) ? R0 : Rup;
        if (R < 2) R = 2;
        const bool single = (n <= R) || (l > 0 && n <= top) || (l == kMaxLevels - 1);
        if (single) {
            fill_level(P.lv[l], B, n, n);  // one segment per chain: plain sequential sweep
            ++l;
            break;
        }
        fill_level(P.lv[l], B, n, R);
        n = P.lv[l].P;
        ++l;
    }
    P.nlevels = l;
    size_t off = 0;
    auto take = [&](size_t nd) { size_t o = off; off += (nd + 63) / 64 * 64; return o; };
    // per-segment partial sums; the wide local kernels keep one partial per node
    P.off_part[0] = take(2 * (P.wide ? std::max<size_t>(P.lv[0].Lpad, (size_t)B * T) : (size_t)P.lv[0].Lpad));
    for (int i = 1; i < P.nlevels; ++i) {
        const LevelDesc& lv = P.lv[i];
        P.off_Dhat[i] = take(level_elems(P, lv, 2));
        P.off_Rsub[i] = take(level_elems(P, lv, 2));
        P.off_S[i] = take(level_elems(P, lv, 1));
        P.off_rhat[i] = take(level_elems(P, lv, 0));
        P.off_rho[i] = take(level_elems(P, lv, 0));
        P.off_L[i] = take(level_elems(P, lv, 3));
        P.off_G[i] = take(level_elems(P, lv, 1));
        P.off_y[i] = take(level_elems(P, lv, 0));
        P.off_Sig[i] = take(level_elems(P, lv, 2));
        P.off_mu[i] = take(level_elems(P, lv, 0));
    }
    P.ws_doubles = off;
    *out = h;
    return 0;
}

void mfgm_plan_destroy(mfgm_plan* plan) { delete plan; }

int mfgm_plan_describe(const mfgm_plan* plan, int* out6) {
    if (!plan || !out6) return 1;
    const Plan& P = plan->p;
    out6[0] = P.nlevels; out6[1] = P.lv[0].R; out6[2] = P.lv[0].P; out6[3] = P.lv[0].Lpad; out6[4] = P.B; out6[5] = P.T;
    return 0;
}

size_t mfgm_plan_workspace_bytes(const mfgm_plan* plan) { return plan ? plan->p.ws_doubles * sizeof(double) : 0; }

size_t mfgm_packed_doubles(const mfgm_plan* plan, int kind) {
    if (!plan || kind < 0 || kind > 3) return 0;
    return level_elems(plan->p, plan->p.lv[0], kind);
}

static int repack(const mfgm_plan* plan, int kind, const double* src, double* dst, int n_nodes, bool pack, void* stream) {
    if (!plan || !src || !dst || kind < 0 || kind > 3) return 1;
    const Plan& P = plan->p;
    if (n_nodes < 0 || n_nodes > P.T) return 1;
    const LevelDesc& lv = P.lv[0];
    const int En = kind_enat(kind, P.d);
    if (P.wide) {
        const size_t total = (size_t)P.B * (pack ? P.T : n_nodes) * En;
        if (total == 0) return 0;
        int blocks = (int)std::min<size_t>((total + 255) / 256, 16384);
        hipLaunchKernelGGL(kw_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, P.B, P.T, P.d, kind, n_nodes, pack);
        MFGM_CHECK_LAUNCH();
        return 0;
    }
    int CH = std::max(1, 64 / En);
    CH = std::min(CH, lv.R);
    dim3 grid(lv.Lpad / 64, ceil_div(lv.R, CH)), block(256);
    size_t shmem = (size_t)64 * (CH * En + 1) * sizeof(double);
    hipStream_t st = (hipStream_t)stream;
    if (pack) hipLaunchKernelGGL((k_repack<true>), grid, block, shmem, st, src, dst, lv, P.d, kind, n_nodes, CH);
    else hipLaunchKernelGGL((k_repack<false>), grid, block, shmem, st, src, dst, lv, P.d, kind, n_nodes, CH);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_pack(const mfgm_plan* plan, int kind, const double* natural, int n_nodes, double* packed, void* stream) {
    return repack(plan, kind, natural, packed, n_nodes, true, stream);
}

int mfgm_unpack(const mfgm_plan* plan, int kind, const double* packed, double* natural, int n_nodes, void* stream) {
    return repack(plan, kind, packed, natural, n_nodes, false, stream);
}

int mfgm_lincomb(size_t n, double* out, double a, const double* x, double b, const double* y, double c, const double* z,
                 void* stream) {
    if (!out || !x) return 1;
    if (n == 0) return 0;
    if (((uintptr_t)out | (uintptr_t)x | (uintptr_t)y | (uintptr_t)z) & 15) return 1;   // 16-byte aligned flat arrays
    const size_t n2 = n / 2 + 1;
    int blocks = (int)std::min<size_t>((n2 + 255) / 256, 2048 * 4);
    hipLaunchKernelGGL(k_lincomb, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, out, a, x, b, y, c, z);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_node_io(const mfgm_plan* plan, int kind, double* packed, double* packed2, const long long* node_ids, int n,
                 double* values, int mode, double scale, void* stream) {
    if (!plan || !packed || kind < 0 || kind > 3 || mode < 0 || mode > 2 || n < 0) return 1;
    if (n == 0) return 0;
    if (!node_ids || !values) return 1;
    const Plan& P = plan->p;
    const size_t total = (size_t)n * kind_enat(kind, P.d);
    if (total >= (1ull << 32)) return 1;
    int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    if (P.wide) {
        hipLaunchKernelGGL(kw_node_io, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P.d, kind, packed, packed2, node_ids, n,
                           values, mode, scale);
        MFGM_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(k_node_io, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P.lv[0], P.T, P.d, kind, packed, packed2,
                       node_ids, n, values, mode, scale);
    MFGM_CHECK_LAUNCH();
    return 0;
}

#define MFGM_DISPATCH_D(d, CALL)             \
    switch (d) {                             \
        case 1: { constexpr int DD = 1; return CALL; } \
        case 2: { constexpr int DD = 2; return CALL; } \
        case 3: { constexpr int DD = 3; return CALL; } \
        case 4: { constexpr int DD = 4; return CALL; } \
        case 5: { constexpr int DD = 5; return CALL; } \
        case 6: { constexpr int DD = 6; return CALL; } \
        case 7: { constexpr int DD = 7; return CALL; } \
        case 8: { constexpr int DD = 8; return CALL; } \
        default: return 1;                   \
    }

int mfgm_packed_factor(const mfgm_plan* plan, const double* D, const double* S, const double* r, double aD, double aS,
                       double aR, double* L, double* G, double* y, double* logdet, double* quad, void* ws, int* info,
                       void* stream) {
    if (!plan || !D || !L || !G || !info) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !S) return 1;
    if ((r != nullptr) != (y != nullptr)) return 1;
    if (!ws && P.ws_doubles > 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (P.wide) return wide_factor(P, D, S, r, aD, aS, aR, L, G, y, logdet, quad, (double*)ws, info, st);
    MFGM_DISPATCH_D(P.d, (factor_impl<DD>(P, D, S, r, aD, aS, aR, L, G, y, logdet, quad, (double*)ws, info, st)));
}

int mfgm_packed_selinv(const mfgm_plan* plan, const double* L, const double* G, const double* y, double* Sig,
                       double* Sub, double* x, void* ws, void* stream) {
    if (!plan || !L || !G || !Sig) return 1;
    const Plan& P = plan->p;
    if ((y != nullptr) != (x != nullptr)) return 1;
    if (!ws && P.ws_doubles > 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (P.wide) return wide_selinv(P, L, G, y, Sig, Sub, x, (double*)ws, st);
    MFGM_DISPATCH_D(P.d, (selinv_impl<DD>(P, L, G, y, Sig, Sub, x, (double*)ws, st)));
}

}  // extern "C"

namespace {
template <int D>
int s2n_impl(const Plan& P, const double* A, const double* off, const double* chol, double cD, double cS, double* lin,
             double* diag, double* sub, double* sumlogchol, double* ws, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    double* part = sumlogchol ? ws + P.off_part[0] : nullptr;
    if (lin) hipLaunchKernelGGL((k_ssm_to_naturals<D, true>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, A, off, chol, cD, cS, lin, diag, sub, part);
    else hipLaunchKernelGGL((k_ssm_to_naturals<D, false>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, A, off, chol, cD, cS, lin, diag, sub, part);
    MFGM_CHECK_LAUNCH();
    if (sumlogchol) {
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, part, lv.P, 0, sumlogchol, (double*)nullptr);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}
template <int D>
int kl_impl(const Plan& P, const double* Sig, const double* Sub, const double* mu, const double* Pd, const double* Ps, double aD,
            double aS, const double* mup, double* trace, double* maha, double* ws, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    double* part = ws + P.off_part[0];
    hipLaunchKernelGGL((k_kl_terms<D>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, Sig, Sub, mu, Pd, Ps, aD, aS, mup, part);
    MFGM_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, part, lv.P, lv.Lpad, trace, maha);
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

int mfgm_packed_ssm_to_naturals(const mfgm_plan* plan, const double* A, const double* off, const double* chol, double cD,
                                double cS, double* lin, double* diag, double* sub, double* sumlogchol, void* ws,
                                void* stream) {
    if (!plan || !chol || !diag || !sub || !ws) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !A) return 1;
    if ((lin != nullptr) != (off != nullptr)) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (P.wide) {
        double* part = sumlogchol ? (double*)ws + P.off_part[0] : nullptr;
        dim3 grid(P.B * P.T), block(64);
#define S2N(DM_, LIN_) hipLaunchKernelGGL((kw_ssm_to_naturals<DM_, LIN_>), grid, block, 0, st, P.B, P.T, P.d, A, off, chol, cD, cS, lin, diag, sub, part)
        if (P.d <= 16) { if (lin) S2N(16, true); else S2N(16, false); }
        else { if (lin) S2N(32, true); else S2N(32, false); }
#undef S2N
        MFGM_CHECK_LAUNCH();
        if (sumlogchol) {
            hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, part, P.T, 0, sumlogchol, (double*)nullptr);
            MFGM_CHECK_LAUNCH();
        }
        return 0;
    }
    MFGM_DISPATCH_D(P.d, (s2n_impl<DD>(P, A, off, chol, cD, cS, lin, diag, sub, sumlogchol, (double*)ws, st)));
}

int mfgm_packed_kl_terms(const mfgm_plan* plan, const double* Sig, const double* Sub, const double* mu, const double* Pd,
                         const double* Ps, double aD, double aS, const double* mup, double* trace, double* maha, void* ws,
                         void* stream) {
    if (!plan || !Sig || !Sub || !mu || !Pd || !Ps || !mup || !trace || !maha || !ws) return 1;
    const Plan& P = plan->p;
    hipStream_t st = (hipStream_t)stream;
    if (P.wide) {
        double* part = (double*)ws + P.off_part[0];
        hipLaunchKernelGGL(kw_kl_terms, dim3(P.B * P.T), dim3(64), 0, st, P.B, P.T, P.d, Sig, Sub, mu, Pd, Ps, aD, aS, mup, part);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, part, P.T, P.B * P.T, trace, maha);
        MFGM_CHECK_LAUNCH();
        return 0;
    }
    MFGM_DISPATCH_D(P.d, (kl_impl<DD>(P, Sig, Sub, mu, Pd, Ps, aD, aS, mup, trace, maha, (double*)ws, st)));
}

}  // extern "C"

extern "C" {

// Profiling / roofline entry points: launch exactly ONE kernel of a sweep (stage 0 = reduce, 1 = forward; level 0 is
// the finest).  The coarser levels must already be in `ws` from a full mfgm_packed_factor / mfgm_packed_selinv call
// with the same arguments; outputs are overwritten with identical values.
int mfgm_packed_factor_stage(const mfgm_plan* plan, int stage, int level, const double* D, const double* S, const double* r,
                             double aD, double aS, double aR, double* L, double* G, double* y, void* ws, int* info,
                             void* stream) {
    if (!plan || !D || !L || !G || !info || stage < 0 || stage > 1) return 1;
    const Plan& P = plan->p;
    if (level < 0 || level >= P.nlevels || (stage == 0 && level >= P.nlevels - 1)) return 1;
    if ((r != nullptr) != (y != nullptr)) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (factor_impl<DD>(P, D, S, r, aD, aS, aR, L, G, y, nullptr, nullptr, (double*)ws, info, st, stage, level)));
}

int mfgm_packed_selinv_level(const mfgm_plan* plan, int level, const double* L, const double* G, const double* y, double* Sig,
                             double* Sub, double* x, void* ws, void* stream) {
    if (!plan || !L || !G || !Sig) return 1;
    const Plan& P = plan->p;
    if (level < 0 || level >= P.nlevels) return 1;
    if ((y != nullptr) != (x != nullptr)) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (selinv_impl<DD>(P, L, G, y, Sig, Sub, x, (double*)ws, st, level)));
}

}  // extern "C"

static_assert(sizeof(mfgm_sde_params) == sizeof(mfgm::SdeParams), "public and internal SDE parameter structs must match");

namespace {
template <int D>
int sde_kl_impl(const Plan& P, int mode, const SdeParams& pr, const double* mu, const double* Sig, const double* Sub, double* kl,
                double* o1, double* od, double* os, double* q1, double* qd, double* qs, double* ws, int* info, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    double* part = kl ? ws + P.off_part[0] : nullptr;
    dim3 grid(lv.Lpad / 64), block(64);
    if (mode == 0) hipLaunchKernelGGL((k_sde_kl<D, 0>), grid, block, 0, st, lv, pr, mu, Sig, Sub, part, o1, od, os, q1, qd, qs, info);
    else if (mode == 1) hipLaunchKernelGGL((k_sde_kl<D, 1>), grid, block, 0, st, lv, pr, mu, Sig, Sub, part, o1, od, os, q1, qd, qs, info);
    else if (mode == 2) hipLaunchKernelGGL((k_sde_kl<D, 2>), grid, block, 0, st, lv, pr, mu, Sig, Sub, part, o1, od, os, q1, qd, qs, info);
    else hipLaunchKernelGGL((k_sde_kl<D, 3>), grid, block, 0, st, lv, pr, mu, Sig, Sub, part, o1, od, os, q1, qd, qs, info);
    MFGM_CHECK_LAUNCH();
    if (kl) {
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, part, lv.P, 0, kl, (double*)nullptr);
        MFGM_CHECK_LAUNCH();
    }
    return 0;
}
template <int D>
int linearize_impl(const Plan& P, const SdeParams& pr, const double* mu, const double* Sig, double* A, double* off, double* chol,
                   hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    hipLaunchKernelGGL((k_linearize_cubic<D>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, pr, mu, Sig, A, off, chol);
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

int mfgm_packed_sde_kl(const mfgm_plan* plan, int mode, const mfgm_sde_params* prm, const double* mu, const double* Sig,
                       const double* Sub, double* kl, double* o1, double* od, double* os, double* q1, double* qd, double* qs,
                       void* ws, int* info, void* stream) {
    if (!plan || !prm || !mu || !Sig || !Sub || !info || !ws || mode < 0 || mode > 3) return 1;
    if (mode == 0 && !kl) return 1;
    if ((mode == 1 || mode == 2) && (!o1 || !od || !os)) return 1;
    if (mode >= 2 && (!q1 || !qd || !qs)) return 1;
    const Plan& P = plan->p;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (sde_kl_impl<DD>(P, mode, pr, mu, Sig, Sub, kl, o1, od, os, q1, qd, qs, (double*)ws, info, st)));
}

int mfgm_packed_linearize_cubic(const mfgm_plan* plan, const mfgm_sde_params* prm, const double* mu, const double* Sig,
                                double* A, double* off, double* chol, void* stream) {
    if (!plan || !prm || !mu || !Sig || !A || !off || !chol) return 1;
    const Plan& P = plan->p;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (linearize_impl<DD>(P, pr, mu, Sig, A, off, chol, st)));
}

}  // extern "C"

static_assert(sizeof(mfgm_kernel_spec) == sizeof(mfgm::KernelSpec), "public and internal kernel spec structs must match");

namespace {
template <int D>
int stationary_impl(const Plan& P, const KernelSpec& ks, const double* dts, double* A, double* off, double* chol, int* info,
                    hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    hipLaunchKernelGGL((k_stationary_ssm<D>), dim3(lv.Lpad / 64), dim3(64), 0, st, lv, ks, dts, A, off, chol, info);
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int mfgm_packed_stationary_ssm(const mfgm_plan* plan, const mfgm_kernel_spec* spec, const double* time_deltas,
                                          double* A, double* off, double* chol, int* info, void* stream) {
    if (!plan || !spec || !A || !off || !chol || !info) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !time_deltas) return 1;
    KernelSpec ks;
    memcpy(&ks, spec, sizeof(ks));
    if (ks.ncomp < 1 || ks.ncomp > 8) return 1;
    int dim = 0;
    for (int c = 0; c < ks.ncomp; ++c) {
        if (ks.order[c] < 1 || ks.order[c] > 3 || ks.offset[c] != dim) return 1;
        dim += ks.order[c];
    }
    if (dim != P.d) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (stationary_impl<DD>(P, ks, time_deltas, A, off, chol, info, st)));
}

static_assert(sizeof(mfgm_vdp_params) == sizeof(mfgm::VdpParams), "public and internal VDP parameter structs must match");

namespace {
template <int D>
int vdp_impl(int what, const Plan& P, const VdpParams& pr, const double* a0, const double* a1, const double* a2, const double* a3,
             const double* a4, const double* a5, double* o0, double* o1, double* o2, double* ws, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    dim3 grid(lv.Lpad / 64), block(64);
    if (what == 0) {
        hipLaunchKernelGGL((k_vdp_to_ssm<D>), grid, block, 0, st, lv, pr, a0, a1, o0, o1, o2);
    } else if (what == 1) {
        double* part = ws + P.off_part[0];
        if (o1) hipLaunchKernelGGL((k_vdp_esde<D, true>), grid, block, 0, st, lv, pr, a0, a1, a2, a3, part, o1, o2);
        else hipLaunchKernelGGL((k_vdp_esde<D, false>), grid, block, 0, st, lv, pr, a0, a1, a2, a3, part, o1, o2);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, part, lv.P, 0, o0, (double*)nullptr);
    } else if (what == 2) {
        hipLaunchKernelGGL((k_vdp_lagrange<D, 1>), grid, block, 0, st, lv, pr, a0, a1, a2, a3, a4, a5, o0, o1, o2);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL((k_vdp_lagrange_scan<D>), dim3((P.B + 63) / 64), block, 0, st, lv, o2);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL((k_vdp_lagrange<D, 3>), grid, block, 0, st, lv, pr, a0, a1, a2, a3, a4, a5, o0, o1, o2);
    } else {
        hipLaunchKernelGGL((k_vdp_update_param<D>), grid, block, 0, st, lv, pr, a0, a1, a2, a3, o0, o1);
    }
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

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
                             double* seg, void* stream) {
    if (!plan || !prm || !mu || !Sig || !Am || !bm || !yR || !dobsS || !psi || !lam || !seg) return 1;
    const Plan& P = plan->p;
    VdpParams pr; memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (vdp_impl<DD>(2, P, pr, mu, Sig, Am, bm, yR, dobsS, psi, lam, seg, nullptr, st)));
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

namespace {
template <int D>
int sde_lean_impl(const Plan& P, int mode, const SdeParams& pr, const double* mom, const double* Sig, double* out, double* q1,
                  double* qd, double* qs, double* ws, hipStream_t st) {
    const LevelDesc& lv = P.lv[0];
    dim3 grid(lv.Lpad / 64), block(64);
    if (mode == 0) {
        double* part = ws + P.off_part[0];
        hipLaunchKernelGGL((k_sde_lean<D, 0>), grid, block, 0, st, lv, pr, mom, Sig, part, q1, qd, qs);
        MFGM_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_sum_partials, dim3(P.B), dim3(64), 0, st, part, lv.P, 0, out, (double*)nullptr);
    } else {
        hipLaunchKernelGGL((k_sde_lean<D, 3>), grid, block, 0, st, lv, pr, mom, Sig, (double*)nullptr, q1, qd, qs);
    }
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

int mfgm_packed_selinv_mom(const mfgm_plan* plan, int only_level, const double* L, const double* G, const double* y, double* Sig,
                           double* Sub, double* x, double* mom, void* ws, void* stream) {
    if (!plan || !L || !G || !Sig || !y || !x || !mom) return 1;
    const Plan& P = plan->p;
    if (only_level >= P.nlevels) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (selinv_impl<DD>(P, L, G, y, Sig, Sub, x, (double*)ws, st, only_level, mom)));
}

int mfgm_packed_sde_lean(const mfgm_plan* plan, int mode, const mfgm_sde_params* prm, const double* mom, const double* Sig,
                         double* kl_part, double* q1, double* qd, double* qs, void* ws, void* stream) {
    if (!plan || !prm || !mom || !ws || (mode != 0 && mode != 3)) return 1;
    if (mode == 0 && (!kl_part || !Sig)) return 1;
    if (mode == 3 && (!q1 || !qd || !qs)) return 1;
    const Plan& P = plan->p;
    SdeParams pr;
    memcpy(&pr, prm, sizeof(pr));
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (sde_lean_impl<DD>(P, mode, pr, mom, Sig, kl_part, q1, qd, qs, (double*)ws, st)));
}

}  // extern "C"

// ---- natural-layout convenience entry points ---------------------------------------------------------------------------
namespace {
struct NatWs {
    double *D, *S, *r, *L, *G, *y, *Sig, *Sub, *x, *ws;
};
NatWs carve(const Plan& P, void* nws) {
    const LevelDesc& lv = P.lv[0];
    auto al = [](size_t n) { return (n + 63) / 64 * 64; };
    double* p = (double*)nws;
    NatWs w;
    w.D = p; p += al(level_elems(P, lv, 2));
    w.S = p; p += al(level_elems(P, lv, 1));
    w.r = p; p += al(level_elems(P, lv, 0));
    w.L = p; p += al(level_elems(P, lv, 3));
    w.G = p; p += al(level_elems(P, lv, 1));
    w.y = p; p += al(level_elems(P, lv, 0));
    w.Sig = p; p += al(level_elems(P, lv, 2));
    w.Sub = p; p += al(level_elems(P, lv, 1));
    w.x = p; p += al(level_elems(P, lv, 0));
    w.ws = p;
    return w;
}
}  // namespace

extern "C" {

size_t mfgm_natural_workspace_bytes(const mfgm_plan* plan) {
    if (!plan) return 0;
    const Plan& P = plan->p;
    const LevelDesc& lv = P.lv[0];
    auto al = [](size_t n) { return (n + 63) / 64 * 64; };
    size_t n = 2 * al(level_elems(P, lv, 2)) + al(level_elems(P, lv, 3)) + 3 * al(level_elems(P, lv, 1)) +
               3 * al(level_elems(P, lv, 0)) + P.ws_doubles;
    return n * sizeof(double);
}

int mfgm_btd_cholesky(const mfgm_plan* plan, const double* diag, const double* sub, double aD, double aS, double* Ldiag,
                      double* Lsub, double* logdet, void* nws, int* info, void* stream) {
    if (!plan || !diag || !Ldiag || !nws || !info) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && (!sub || !Lsub)) return 1;
    NatWs w = carve(P, nws);
    int rc;
    if ((rc = mfgm_pack(plan, MFGM_SYM, diag, P.T, w.D, stream))) return rc;
    if (P.T > 1 && (rc = mfgm_pack(plan, MFGM_FULL, sub, P.T - 1, w.S, stream))) return rc;
    if ((rc = mfgm_packed_factor(plan, w.D, w.S, nullptr, aD, aS, 1.0, w.L, w.G, nullptr, logdet, nullptr, w.ws, info, stream))) return rc;
    if ((rc = mfgm_unpack(plan, MFGM_TRI, w.L, Ldiag, P.T, stream))) return rc;
    if (P.T > 1 && (rc = mfgm_unpack(plan, MFGM_FULL, w.G, Lsub, P.T - 1, stream))) return rc;
    return 0;
}

int mfgm_btd_posterior(const mfgm_plan* plan, const double* diag, const double* sub, const double* rhs, double aD, double aS,
                       double aR, double* logdet, double* x, double* Sdiag, double* Ssub, void* nws, int* info, void* stream) {
    if (!plan || !diag || !Sdiag || !nws || !info) return 1;
    const Plan& P = plan->p;
    if (P.T > 1 && !sub) return 1;
    if ((rhs != nullptr) != (x != nullptr)) return 1;
    NatWs w = carve(P, nws);
    int rc;
    if ((rc = mfgm_pack(plan, MFGM_SYM, diag, P.T, w.D, stream))) return rc;
    if (P.T > 1 && (rc = mfgm_pack(plan, MFGM_FULL, sub, P.T - 1, w.S, stream))) return rc;
    if (rhs && (rc = mfgm_pack(plan, MFGM_VEC, rhs, P.T, w.r, stream))) return rc;
    if ((rc = mfgm_packed_factor(plan, w.D, w.S, rhs ? w.r : nullptr, aD, aS, aR, w.L, w.G, rhs ? w.y : nullptr, logdet, nullptr,
                                 w.ws, info, stream))) return rc;
    const bool want_sub = (Ssub != nullptr) && P.T > 1;
    if ((rc = mfgm_packed_selinv(plan, w.L, w.G, rhs ? w.y : nullptr, w.Sig, want_sub ? w.Sub : nullptr, rhs ? w.x : nullptr, w.ws,
                                 stream))) return rc;
    if ((rc = mfgm_unpack(plan, MFGM_SYM, w.Sig, Sdiag, P.T, stream))) return rc;
    if (want_sub && (rc = mfgm_unpack(plan, MFGM_FULL, w.Sub, Ssub, P.T - 1, stream))) return rc;
    if (rhs && (rc = mfgm_unpack(plan, MFGM_VEC, w.x, x, P.T, stream))) return rc;
    return 0;
}

}  // extern "C"

// ---- batched small dense SPD algebra (natural layout) ------------------------------------------------------------------
extern "C" {

int mfgm_batched_cholesky(int N, int d, const double* A, double* L, int* info, void* stream) {
    if (N < 0 || d < 1 || d > 32 || !info) return 1;
    if (N == 0) return 0;
    if (!A || !L || A == L) return 1;
    hipLaunchKernelGGL(k_batched_chol, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, N, d, A, L, info);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_batched_trsm(int N, int d, int m, int lbatch, const double* L, const double* B, double* X, int mode, void* stream) {
    if (N < 0 || d < 1 || d > 32 || m < 1 || mode < 1 || mode > 3 || (lbatch != 1 && lbatch != N)) return 1;
    if (N == 0) return 0;
    if (!L || !B || !X) return 1;
    const long long total = (long long)N * m;
    hipLaunchKernelGGL(k_batched_trsm, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream, N, d, m, lbatch, L, B,
                       X, mode);
    MFGM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
