// Re-layout and node gather / scatter for the wide (8 < d <= 32) arrays, which are natural-layout [B, T, E].
#pragma once
#include <cstddef>

namespace mfgm {

// natural [B, n_nat, E] <-> wide [B, T, E] (nodes >= n_nat are zero-filled on pack).  kind 2 symmetrises from the lower
// triangle on pack; kind 3 zeroes the strict upper triangle in both directions.
static __global__ __launch_bounds__(256) void kw_copy(const double* __restrict__ src, double* __restrict__ dst, int B, int T, int d,
                                              int kind, int n_nat, int pack) {
    const int E = (kind == 0) ? d : d * d;
    const int nout = pack ? T : n_nat;
    const size_t total = (size_t)B * nout * E;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx % E);
        const size_t bt = idx / E;
        const int t = (int)(bt % nout), b = (int)(bt / nout);
        int es = e;
        bool zero = false;
        if (kind >= 2) {
            const int r = e / d, c = e - r * d;
            if (c > r) {
                if (kind == 3) zero = true;
                else if (pack) es = c * d + r;
            }
        }
        double v = 0.0;
        if (!zero && t < n_nat) v = src[((size_t)b * (pack ? n_nat : T) + t) * E + es];
        dst[idx] = v;
    }
}

// gather / scatter of listed nodes (k_node_io semantics) on wide arrays
static __global__ __launch_bounds__(256) void kw_node_io(int d, int kind, double* packed, double* packed2,
                                                 const long long* __restrict__ node_ids, int n, double* values, int mode,
                                                 double scale) {
    const unsigned E = (kind == 0) ? d : d * d;
    const unsigned total = (unsigned)n * E;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned i = idx / E, e = idx - i * E;
        const size_t off = (size_t)node_ids[i] * E + e;
        unsigned es = e;
        bool zero = false;
        if (kind >= 2) {
            const unsigned r = e / d, c = e - r * d;
            if (c > r) {
                if (kind == 3) zero = true;
                else if (mode != 0) es = c * d + r;      // symmetric scatters read the lower triangle
            }
        }
        if (mode == 0) values[idx] = zero ? 0.0 : packed[off];
        else {
            const double v = zero ? 0.0 : values[i * E + es];
            if (mode == 1) packed[off] = v;
            else {
                packed[off] += scale * v;
                if (packed2) packed2[off] += scale * v;
            }
        }
    }
}

}  // namespace mfgm
