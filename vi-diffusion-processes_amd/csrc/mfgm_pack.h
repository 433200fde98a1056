// Natural [B, T, ...] row-major  <->  packed segment-interleaved layout (mfgm_layout.h).
// Replaces the reference's block_to_band / band_to_block re-layout calls
// (block_tri_diag.py:206-237, 553-596) -- done once at the boundary instead of on every property access.
#pragma once
#include "mfgm_layout.h"
#include "mfgm_math.h"

namespace mfgm {

enum PackKind { PK_VEC = 0, PK_FULL = 1, PK_SYM = 2, PK_TRI = 3 };

__host__ __device__ inline int kind_enat(int kind, int d) { return kind == PK_VEC ? d : d * d; }
__host__ __device__ inline int kind_epack(int kind, int d) {
    return kind == PK_VEC ? d : (kind == PK_FULL ? d * d : d * (d + 1) / 2);
}

// tile: 64 lanes x CH steps.  Natural side is read/written in per-lane contiguous runs of CH*E_nat doubles,
// packed side in 512-byte lane-contiguous rows; the transpose goes through LDS (odd row stride, conflict free).
template <bool PACK>
__global__ __launch_bounds__(256) void k_repack(const double* __restrict__ src, double* __restrict__ dst, LevelDesc lv,
                                               int d, int kind, int n_nat, int CH) {
    extern __shared__ double tile[];
    const int En = kind_enat(kind, d), Ep = kind_epack(kind, d);
    const int Emax = En;  // En >= Ep
    const int stride = CH * Emax + 1;
    const int lane0 = blockIdx.x * 64, s0 = blockIdx.y * CH;
    const int tid = threadIdx.x;
    const int P = lv.P, R = lv.R, Lpad = lv.Lpad;

    if (PACK) {
        for (int idx = tid; idx < 64 * CH * En; idx += 256) {
            const int li = idx / (CH * En), off = idx - li * (CH * En);
            const int lane = lane0 + li, sl = off / En, ne = off - sl * En;
            double v = 0.0;
            if (lane < lv.L) {
                const int b = lane / P, p = lane - b * P, s = s0 + sl, t = p * R + s;
                if (s < R && t < n_nat) v = src[((size_t)b * n_nat + t) * En + ne];
            }
            tile[li * stride + off] = v;
        }
        __syncthreads();
        for (int idx = tid; idx < CH * Ep * 64; idx += 256) {
            const int li = idx & 63, se = idx >> 6, sl = se / Ep, e = se - sl * Ep;
            if (s0 + sl < R && lane0 + li < Lpad) {
                int ne = e;
                if (kind >= PK_SYM) {
                    int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
                    while (tix(i + 1, 0) <= e) ++i;
                    while (tix(i, 0) > e) --i;
                    ne = i * d + (e - tix(i, 0));
                }
                dst[(((size_t)blockIdx.x * R + (s0 + sl)) * Ep + e) * 64 + li] = tile[li * stride + sl * En + ne];
            }
        }
    } else {
        for (int idx = tid; idx < CH * Ep * 64; idx += 256) {
            const int li = idx & 63, se = idx >> 6, sl = se / Ep, e = se - sl * Ep;
            double v = 0.0;
            if (s0 + sl < R && lane0 + li < Lpad) v = src[(((size_t)blockIdx.x * R + (s0 + sl)) * Ep + e) * 64 + li];
            tile[li * stride + sl * Ep + e] = v;
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * CH * En; idx += 256) {
            const int li = idx / (CH * En), off = idx - li * (CH * En);
            const int lane = lane0 + li, sl = off / En, ne = off - sl * En;
            if (lane < lv.L) {
                const int b = lane / P, p = lane - b * P, s = s0 + sl, t = p * R + s;
                if (s < R && t < n_nat) {
                    double v;
                    if (kind < PK_SYM) {
                        v = tile[li * stride + sl * Ep + ne];
                    } else {
                        const int i = ne / d, j = ne - i * d;
                        if (kind == PK_SYM) v = tile[li * stride + sl * Ep + six(i, j)];
                        else v = (j <= i) ? tile[li * stride + sl * Ep + tix(i, j)] : 0.0;
                    }
                    dst[((size_t)b * n_nat + t) * En + ne] = v;
                }
            }
        }
    }
}

}  // namespace mfgm
