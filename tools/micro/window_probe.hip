// Micro-benchmark of the sweep kernels' memory pattern: one wave per SIMD; per step a wave prefetches E doubles per lane for the
// next step, does NF dependent-free f64 FMAs per lane ("factorisation"), stores ES doubles per lane, then consumes the prefetch.
// Compares 8-byte per-lane accesses (E + ES instructions per step; more than the 64 the vmcnt window holds) with 16-byte ones.
// Build: hipcc --offload-arch=gfx950 -O3 window_probe.hip -o window_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int E, int ES, int W, int NF>
__global__ __launch_bounds__(64) void k_sweep(const double* __restrict__ src, double* __restrict__ dst, int R) {
    const int tile = blockIdx.x, l = threadIdx.x;
    constexpr int EP = E / W, SP = ES / W;
    double cur[E], nxt[E];
    const double* p0 = src + (size_t)tile * R * (E * 64);
    double* q0 = dst + (size_t)tile * R * (ES * 64);
    auto load = [&](int s, double (&b)[E]) {
        const double* p = p0 + (size_t)s * (E * 64);
        if (W == 1) {
#pragma unroll
            for (int e = 0; e < EP; ++e) b[e] = p[e * 64 + l];
        } else {
#pragma unroll
            for (int e = 0; e < EP; ++e) {
                double2 v = reinterpret_cast<const double2*>(p)[e * 64 + l];
                b[2 * e] = v.x; b[2 * e + 1] = v.y;
            }
        }
    };
    load(0, nxt);
    double carry = 0.0;
    for (int s = 0; s < R; ++s) {
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = nxt[e] + carry;
        if (s + 1 < R) load(s + 1, nxt);
        // NF FMAs per lane over the E values (E independent chains)
#pragma unroll
        for (int k = 0; k < NF / E; ++k)
#pragma unroll
            for (int e = 0; e < E; ++e) cur[e] = __builtin_fma(cur[e], 1.0000001, 1e-9);
        // fold every loaded value into what is stored (or carried), so that no load is dead
        if (ES > 0) {
#pragma unroll
            for (int e = ES; e < E; ++e) cur[e % ES] += cur[e];
        } else {
#pragma unroll
            for (int e = 0; e < E - 1; ++e) cur[E - 1] += cur[e];
        }
        double* q = q0 + (size_t)s * (ES * 64);
        if (W == 1) {
#pragma unroll
            for (int e = 0; e < SP; ++e) q[e * 64 + l] = cur[e];
        } else {
#pragma unroll
            for (int e = 0; e < SP; ++e) reinterpret_cast<double2*>(q)[e * 64 + l] = make_double2(cur[2 * e], cur[2 * e + 1]);
        }
        carry = cur[ES > 0 ? 0 : E - 1] * 1e-30;
    }
    if (carry == 12345.0) dst[l] = carry;      // keeps the read-only variant alive
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

template <int E, int ES, int NF>
void run(const double* src, double* dst, int tiles, int R) {
    const double gb = (double)tiles * R * 64 * (E + ES) * 8 / 1e9;
    float t1 = timeit([&] { hipLaunchKernelGGL((k_sweep<E, ES, 1, NF>), dim3(tiles), dim3(64), 0, 0, src, dst, R); }, 10);
    float t2 = timeit([&] { hipLaunchKernelGGL((k_sweep<E, ES, 2, NF>), dim3(tiles), dim3(64), 0, 0, src, dst, R); }, 10);
    printf("E=%d ES=%d NF=%4d :  8B/lane %.3f ms %.2f TB/s | 16B/lane %.3f ms %.2f TB/s\n", E, ES, NF, t1, gb / t1, t2, gb / t2);
}

int main() {
    const int tiles = 1021, R = 98;
    const size_t n = (size_t)tiles * R * 64 * 96;
    double *src, *dst;
    (void)hipMalloc(&src, n * 8); (void)hipMalloc(&dst, n * 8);
    (void)hipMemset(src, 0, n * 8);
    run<64, 28, 0>(src, dst, tiles, R);       // forward-like bytes, no arithmetic
    run<64, 28, 512>(src, dst, tiles, R);     // ~ forward: 500 f64 ops per step
    run<64, 28, 1024>(src, dst, tiles, R);    // ~ backward: 1000 f64 ops per step
    run<64, 0, 1024>(src, dst, tiles, R);     // ~ reduce: reads only
    run<64, 44, 1024>(src, dst, tiles, R);    // ~ backward with moments
    run<90, 64, 1280>(src, dst, tiles, R);    // ~ fused Girsanov backward
    printf("two and four waves per SIMD, segments of R/2 and R/4 (what the sweeps could use with <= 256 / 128 registers):\n");
    run<64, 28, 512>(src, dst, 2 * tiles, R / 2);
    run<64, 28, 512>(src, dst, 4 * tiles, R / 4);
    run<64, 0, 1024>(src, dst, 2 * tiles, R / 2);
    run<64, 44, 1024>(src, dst, 2 * tiles, R / 2);
    return 0;
}
