// C-ABI entry points (include/mfgm.h): plan construction, re-layout, flat helpers, natural-layout convenience calls,
// batched small dense algebra.
#include "mfgm_internal.h"
#include "mfgm_pack.h"
#include "mfgm_wide_io.h"
#include "mfgm_local.h"
#include "mfgm_batched.h"

using namespace mfgm;

namespace {

void fill_level(LevelDesc& lv, int B, int n, int R, int level) {
    lv.level = level;
    lv.n = n;
    lv.R = R;
    lv.P = ceil_div(n, R);
    lv.L = B * lv.P;
    lv.Lpad = ceil_div(lv.L, 64) * 64;
}

}  // namespace

extern "C" {

#ifndef MFGM_BUILD_ID
#define MFGM_BUILD_ID "unknown"
#endif
const char* mfgm_version(void) { return "mfgm 0.2 (gfx950) build " MFGM_BUILD_ID; }

int mfgm_plan_create(int B, int T, int d, int R0, int Rup, mfgm_plan** out) {
    if (!out || B < 1 || T < 1 || d < 1 || d > 32) return 1;
    mfgm_plan* h = new mfgm_plan();
    Plan& P = h->p;
    memset(&P, 0, sizeof(P));
    P.B = B; P.T = T; P.d = d;
    P.wide = (d > 8);
    // measured best on MI355X for the coarse levels (tools/sweep_coarse.sh, tools/sweep_wide.sh): short segments above level 0 --
    // those levels are latency-bound (few lanes, dependent steps), so their depth counts, not their traffic
    if (Rup <= 1) Rup = 4;
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
    // chains this short are swept sequentially by one lane (narrow) or one wavefront (wide)
    int top = 12;
    if (const char* e = getenv("MFGM_TOP")) { int v = atoi(e); if (v > 1) top = v; }
    while (true) {
        int R = (l == 0) ? R0 : Rup;
        if (R < 2) R = 2;
        const bool single = (n <= R) || (l > 0 && n <= top) || (l == kMaxLevels - 1);
        if (single) {
            fill_level(P.lv[l], B, n, n, l);  // one segment per chain: plain sequential sweep
            ++l;
            break;
        }
        fill_level(P.lv[l], B, n, R, l);
        n = P.lv[l].P;
        ++l;
    }
    P.nlevels = l;
    for (int k = 0; k < l; ++k) P.lv[k].nt = 0;
    if (!P.wide) {
        // Cache policy of the level-0 arrays (LevelDesc::nt, ld_node / st_node): streamed when one array of d x d blocks is larger than
        // half the 256 MB of Infinity Cache -- then nothing a pass writes is still cached when the next pass reads it, and dirty lines
        // left behind are written back under the next pass's traffic.  Used by the CVI-DP sweeps (mfgm_cq.h): headline step -1.0 ...
        // -1.4 % in three same-box A/Bs (MFGM_NT=2 against 0), KL sweep -2.5 ... -6 %; models that fit the caches keep the default
        // policy (config 2 is 3.5 % slower without it).  MFGM_NT=0 / 1 / 2 forces it.
        const double bytes = 8.0 * (double)B * (double)T * (double)d * (double)d;
        int nt = bytes >= 128.0 * 1024.0 * 1024.0 ? 2 : 0;
        if (const char* e = getenv("MFGM_NT")) nt = std::max(0, std::min(2, atoi(e)));
        P.lv[0].nt = nt;
    }
    P.seg_lo = 0;
    P.seg_hi = P.lv[0].P;
    size_t off = 0;
    auto take = [&](size_t nd) { size_t o = off; off += (nd + 63) / 64 * 64; return o; };
    // per-segment partial sums (narrow: also the d-vector hand-over of the fused Girsanov sweep); the wide local kernels keep one
    // partial per node
    P.off_part[0] = take(P.wide ? 2 * std::max<size_t>(P.lv[0].Lpad, (size_t)B * (T + 1)) : (size_t)std::max(4, d) * P.lv[0].Lpad);
    P.off_part2 = take(std::max((size_t)2 * B * 128, (size_t)B * (d * d + d)));    // also: the saved boundary correction of a sharded chain
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
    P.off_alt = (!P.wide && P.nlevels >= 2) ? take(P.off_L[1] - P.off_Dhat[1]) : 0;
    P.ws_doubles = off;
    *out = h;
    return 0;
}

void mfgm_plan_destroy(mfgm_plan* plan) { delete plan; }

int mfgm_plan_set_shard_level(mfgm_plan* plan, int level, int node_lo, int node_hi) {
    if (!plan) return 1;
    Plan& P = plan->p;
    if (!P.wide || level < 1 || level >= P.nlevels || node_lo < 0 || node_hi > P.lv[level].n || node_lo >= node_hi) return 1;
    P.shard_level = level;
    // the nodes [node_lo, node_hi) of level `level` are the segments [node_lo, node_hi) of level `level - 1`; going down, the segments
    // of level l-1 inside the segments [lo, hi) of level l are [lo * R_l, min(hi * R_l, P_{l-1}))
    int lo = node_lo, hi = node_hi;
    for (int l = level - 1; l >= 0; --l) {
        P.own_lo[l] = lo;
        P.own_hi[l] = hi;
        if (l > 0) {
            lo = lo * P.lv[l].R;
            hi = std::min(hi * P.lv[l].R, P.lv[l - 1].P);
        }
    }
    for (int l = level; l < P.nlevels; ++l) { P.own_lo[l] = 0; P.own_hi[l] = P.lv[l].P; }
    P.seg_lo = P.own_lo[0];
    P.seg_hi = P.own_hi[0];
    return 0;
}

int mfgm_plan_set_shard(mfgm_plan* plan, int seg_lo, int seg_hi) {
    // level-0 segments [seg_lo, seg_hi) = nodes of level 1: the exchange happens at level 1
    return mfgm_plan_set_shard_level(plan, 1, seg_lo, seg_hi);
}

int mfgm_plan_exchange_region(const mfgm_plan* plan, size_t* offset_doubles, size_t* count_doubles) {
    if (!plan || !offset_doubles || !count_doubles) return 1;
    const Plan& P = plan->p;
    if (P.nlevels < 2) return 1;
    const int l = P.shard_level > 0 ? P.shard_level : 1;
    *offset_doubles = P.off_Dhat[l];
    *count_doubles = P.off_L[l] - P.off_Dhat[l];
    return 0;
}

int mfgm_plan_decode_info(const mfgm_plan* plan, int info_value, int* out4) {
    if (!plan || !out4) return 1;
    out4[0] = out4[1] = out4[2] = out4[3] = -1;
    if (info_value == 0) return 0;
    const Plan& P = plan->p;
    const int code = 0x7fffffff - info_value;              // flag_not_pd (mfgm_math.h): 1 + lane + (level << 27), smallest wins
    const int level = code >> 27, lane = (code & ((1 << 27) - 1)) - 1;
    if (info_value < (1 << 20) || level < 0 || level >= P.nlevels || lane < 0 || lane >= P.lv[level].L) return 2;   // flagged without a location
    const LevelDesc& lv = P.lv[level];
    const int b = lane / lv.P, p = lane - b * lv.P;
    long long span = 1;                                     // level-0 nodes one node of this level stands for
    for (int l = 0; l < level; ++l) span *= P.lv[l].R;
    const long long lo = (long long)p * lv.R * span, hi = std::min<long long>((long long)(p + 1) * lv.R * span, P.T);
    out4[0] = b; out4[1] = (int)lo; out4[2] = (int)hi; out4[3] = level;
    return 2;
}

int mfgm_plan_check_info(const mfgm_plan* plan, const int* info, int* out4, void* stream) {
    if (!plan || !info || !out4) return 1;
    int v = 0;
    if (hipMemcpyAsync(&v, info, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 3;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 3;
    return mfgm_plan_decode_info(plan, v, out4);
}

int mfgm_plan_level(const mfgm_plan* plan, int level, int* out4) {
    if (!plan || !out4 || level < 0 || level >= plan->p.nlevels) return 1;
    const LevelDesc& lv = plan->p.lv[level];
    out4[0] = lv.n; out4[1] = lv.R; out4[2] = lv.P; out4[3] = lv.Lpad;
    return 0;
}

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

int mfgm_unpack_moments(const mfgm_plan* plan, const double* packed_mom, double* natural, void* stream) {
    // the moment array is a per-node vector of 3d doubles: same re-layout as a vector of a chain with state dimension 3d
    if (!plan || !packed_mom || !natural) return 1;
    const Plan& P = plan->p;
    if (P.wide) return 1;
    const LevelDesc& lv = P.lv[0];
    const int En = 3 * P.d;
    int CH = std::min(std::max(1, 64 / En), lv.R);
    dim3 grid(lv.Lpad / 64, ceil_div(lv.R, CH)), block(256);
    size_t shmem = (size_t)64 * (CH * En + 1) * sizeof(double);
    hipLaunchKernelGGL((k_repack<false>), grid, block, shmem, (hipStream_t)stream, packed_mom, natural, lv, En, (int)PK_VEC, P.T, CH);
    MFGM_CHECK_LAUNCH();
    return 0;
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

namespace {
// x <- x + w (g - x) on two arrays in one launch (blockIdx.y picks the array)
__global__ __launch_bounds__(256) void k_site_lerp(double* __restrict__ x1, const double* __restrict__ g1, size_t n1, double* __restrict__ x2,
                                                   const double* __restrict__ g2, size_t n2, double w) {
    double* x = blockIdx.y ? x2 : x1;
    const double* g = blockIdx.y ? g2 : g1;
    const size_t n = blockIdx.y ? n2 : n1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = __builtin_fma(w, g[i] - x[i], x[i]);
}
}  // namespace

int mfgm_site_lerp(double* nat1, const double* g1, size_t n1, double* nat2, const double* g2, size_t n2, double w, void* stream) {
    if (!nat1 || !g1 || !nat2 || !g2) return 1;
    if (n1 == 0 && n2 == 0) return 0;
    const size_t n = std::max(n1, n2);
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_site_lerp, dim3(blocks, 2), dim3(256), 0, (hipStream_t)stream, nat1, g1, n1, nat2, g2, n2, w);
    MFGM_CHECK_LAUNCH();
    return 0;
}

namespace {
// out <- x + w (g - x) on two arrays in one launch (blockIdx.y picks the array); out may be x
__global__ __launch_bounds__(256) void k_site_lerp_to(double* o1, const double* x1, const double* __restrict__ g1, size_t n1, double* o2,
                                                      const double* x2, const double* __restrict__ g2, size_t n2, double w) {
    double* o = blockIdx.y ? o2 : o1;
    const double* x = blockIdx.y ? x2 : x1;
    const double* g = blockIdx.y ? g2 : g1;
    const size_t n = blockIdx.y ? n2 : n1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = __builtin_fma(w, g[i] - x[i], x[i]);
}
// per-chain ELBO of the CVI-DP model on the structured state: sum_j ve_part[b, j] - (kl_part[b] + logdet[b] + c)
__global__ __launch_bounds__(256) void k_cq_elbo(int B, int nblk, const double* __restrict__ ve_part, const double* __restrict__ kl_part,
                                                 const double* __restrict__ logdet, double c, const int* __restrict__ info,
                                                 double* __restrict__ out, double* __restrict__ total) {
    __shared__ double sh[4];
    const bool bad = info && *info != 0;
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        double v = 0.0;
        for (int j = 0; j < nblk; ++j) v += ve_part[(size_t)b * nblk + j];
        v -= kl_part[b] + logdet[b] + c;
        if (bad) v = __builtin_nan("");
        if (out) out[b] = v;
        acc += v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && total) *total = sh[0] + sh[1] + sh[2] + sh[3];
}
}  // namespace

int mfgm_site_lerp_to(double* out1, const double* nat1, const double* g1, size_t n1, double* out2, const double* nat2, const double* g2,
                      size_t n2, double w, void* stream) {
    if (!out1 || !nat1 || !g1 || !out2 || !nat2 || !g2) return 1;
    if (n1 == 0 && n2 == 0) return 0;
    const size_t n = std::max(n1, n2);
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_site_lerp_to, dim3(blocks, 2), dim3(256), 0, (hipStream_t)stream, out1, nat1, g1, n1, out2, nat2, g2, n2, w);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_cq_elbo(int B, int nblk, const double* ve_part, const double* kl_part, const double* logdet, double c, const int* info,
                 double* elbo, double* total, void* stream) {
    if (B < 1 || nblk < 1 || !ve_part || !kl_part || !logdet || (!elbo && !total)) return 1;
    hipLaunchKernelGGL(k_cq_elbo, dim3(1), dim3(256), 0, (hipStream_t)stream, B, nblk, ve_part, kl_part, logdet, c, info, elbo, total);
    MFGM_CHECK_LAUNCH();
    return 0;
}

namespace {
struct CombineW { double w[8]; };
// out[i] = c + ce extra[i] + sum_k w[k] terms[k][i]  (NaN when *info != 0: a pivot block was not positive definite), total = sum_i out[i]
__global__ void k_combine_terms(int n_terms, int n, const double* __restrict__ terms, CombineW w, double c, const double* __restrict__ extra,
                                double ce, const int* __restrict__ info, double* __restrict__ out, double* __restrict__ total) {
    __shared__ double sh[4];
    const bool bad = info && *info != 0;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double v = c + (extra ? ce * extra[i] : 0.0);
        for (int k = 0; k < n_terms; ++k) v = __builtin_fma(w.w[k], terms[(size_t)k * n + i], v);
        if (bad) v = __builtin_nan("");
        if (out) out[i] = v;
        acc += v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && total) *total = sh[0] + sh[1] + sh[2] + sh[3];
}
}  // namespace

int mfgm_combine_terms(int n_terms, int n, const double* terms, const double* w, double c, const double* extra, double ce, const int* info,
                       double* out, double* total, void* stream) {
    if (n_terms < 0 || n_terms > 8 || n < 1 || (n_terms > 0 && (!terms || !w)) || (!out && !total)) return 1;
    CombineW cw;
    for (int k = 0; k < 8; ++k) cw.w[k] = k < n_terms ? w[k] : 0.0;
    hipLaunchKernelGGL(k_combine_terms, dim3(1), dim3(256), 0, (hipStream_t)stream, n_terms, n, terms, cw, c, extra, ce, info, out, total);
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
    if (total >= (1ull << 32) || (size_t)P.B * P.T >= (1ull << 32)) return 1;
    int blocks = (int)std::min<size_t>((total + 255) / 256, 16384);
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

int mfgm_node_io_pair(const mfgm_plan* plan, double* packed_vec, double* packed_sym, const long long* node_ids, int n,
                      double* values_vec, double* values_sym, int mode, double scale, void* stream) {
    if (!plan || !packed_vec || !packed_sym || mode < 0 || mode > 2 || n < 0) return 1;
    if (n == 0) return 0;
    if (!node_ids || !values_vec || !values_sym) return 1;
    const Plan& P = plan->p;
    if (P.wide) return 1;
    const size_t total = (size_t)n * (P.d + P.d * P.d);
    if (total >= (1ull << 32) || (size_t)P.B * P.T >= (1ull << 32)) return 1;
    int blocks = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(k_node_io_pair, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P.lv[0], P.T, P.d, packed_vec, packed_sym,
                       node_ids, n, values_vec, values_sym, mode, scale);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_site_update_pair(const mfgm_plan* plan, double* packed_vec, double* packed_sym, const long long* node_ids, int n,
                          double* sites_vec, double* sites_sym, const double* g_vec, const double* g_sym, double lr, void* stream) {
    if (!plan || !packed_vec || !packed_sym || n < 0) return 1;
    if (n == 0) return 0;
    if (!node_ids || !sites_vec || !sites_sym || !g_vec || !g_sym) return 1;
    const Plan& P = plan->p;
    if (P.wide) return 1;
    const size_t total = (size_t)n * (P.d + P.d * P.d);
    if (total >= (1ull << 32) || (size_t)P.B * P.T >= (1ull << 32)) return 1;
    int blocks = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(k_site_update_pair, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P.lv[0], P.T, P.d, packed_vec, packed_sym,
                       node_ids, n, sites_vec, sites_sym, g_vec, g_sym, lr);
    MFGM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"

namespace {
template <int D>
int mvn_obs_ve_impl(const Plan& P, const double* mu, const double* Sig, const long long* node_ids, int n_per, const double* y,
                    const double* Sinv, double cst, double* out_mu, double* out_cov, double* ve, hipStream_t st) {
    hipLaunchKernelGGL((k_mvn_obs_ve<D>), dim3((n_per + 255) / 256, P.B), dim3(256), 0, st, P.lv[0], P.T, mu, Sig, node_ids, n_per, y,
                       Sinv, cst, out_mu, out_cov, ve);
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" {

int mfgm_mvn_obs_ve(const mfgm_plan* plan, const double* mu, const double* Sig, const long long* node_ids, int n_per, const double* y,
                    const double* Sinv, double cst, double* out_mu, double* out_cov, double* ve, void* stream) {
    if (!plan || !mu || !Sig || !node_ids || n_per < 1 || !y || !Sinv || !ve) return 1;
    const Plan& P = plan->p;
    if (P.wide || (size_t)P.B * P.T >= (1ull << 32)) return 1;
    hipStream_t st = (hipStream_t)stream;
    MFGM_DISPATCH_D(P.d, (mvn_obs_ve_impl<DD>(P, mu, Sig, node_ids, n_per, y, Sinv, cst, out_mu, out_cov, ve, st)));
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
    if (d > 8 && d <= 16) hipLaunchKernelGGL((k_batched_chol_wave<16>), dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, d, A, L, info);
    else if (d > 16) hipLaunchKernelGGL((k_batched_chol_wave<32>), dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, d, A, L, info);
    else hipLaunchKernelGGL(k_batched_chol, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, N, d, A, L, info);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_batched_trsm(int N, int d, int m, int lbatch, const double* L, const double* B, double* X, int mode, void* stream) {
    if (N < 0 || d < 1 || d > 32 || m < 1 || mode < 1 || mode > 3 || (lbatch != 1 && lbatch != N)) return 1;
    if (N == 0) return 0;
    if (!L || !B || !X) return 1;
    if (m >= 8) {
        hipStream_t st = (hipStream_t)stream;
#define TRSM_COLS(DM_, CP_) hipLaunchKernelGGL((k_batched_trsm_cols<DM_, CP_>), dim3((N + 64 / CP_ - 1) / (64 / CP_)), dim3(64), 0, st, N, d, m, lbatch, L, B, X, mode)
#define TRSM_DM(DM_) do { if (m <= 16) TRSM_COLS(DM_, 16); else if (m <= 32) TRSM_COLS(DM_, 32); else TRSM_COLS(DM_, 64); } while (0)
        if (d <= 8) TRSM_DM(8);
        else if (d <= 16) TRSM_DM(16);
        else TRSM_DM(32);
#undef TRSM_DM
#undef TRSM_COLS
        MFGM_CHECK_LAUNCH();
        return 0;
    }
    const long long total = (long long)N * m;
    hipLaunchKernelGGL(k_batched_trsm, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream, N, d, m, lbatch, L, B,
                       X, mode);
    MFGM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
