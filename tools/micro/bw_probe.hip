// Micro-benchmark: streaming-read bandwidth of the wave-tiled packed layout with 8-byte vs 16-byte per-lane loads,
// one wave per SIMD (1024 waves), E doubles per node, R steps.  Build: hipcc --offload-arch=gfx950 -O3 bw_probe.hip -o bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int E, int W>   // W = doubles per lane per load (1 or 2)
__global__ __launch_bounds__(64) void k_read(const double* __restrict__ base, double* __restrict__ out, int R) {
    const int tile = blockIdx.x, l = threadIdx.x;
    double acc = 0.0;
    constexpr int EP = (E + W - 1) / W;
    double buf[EP * W];
    const double* p0 = base + (size_t)tile * R * (EP * W * 64);
    // prefetch step 0
    for (int s = 0; s < R; ++s) {
        const double* p = p0 + (size_t)s * (EP * W * 64);
        if (W == 1) {
#pragma unroll
            for (int e = 0; e < EP; ++e) buf[e] = p[e * 64 + l];
        } else {
#pragma unroll
            for (int e = 0; e < EP; ++e) {
                double2 v = reinterpret_cast<const double2*>(p)[e * 64 + l];
                buf[2 * e] = v.x; buf[2 * e + 1] = v.y;
            }
        }
#pragma unroll
        for (int e = 0; e < EP * W; ++e) acc += buf[e];
    }
    out[tile * 64 + l] = acc;
}

template <int E, int W>
__global__ __launch_bounds__(64) void k_copy(const double* __restrict__ base, double* __restrict__ dst, int R) {
    const int tile = blockIdx.x, l = threadIdx.x;
    constexpr int EP = (E + W - 1) / W;
    const size_t off0 = (size_t)tile * R * (EP * W * 64);
    for (int s = 0; s < R; ++s) {
        const double* p = base + off0 + (size_t)s * (EP * W * 64);
        double* q = dst + off0 + (size_t)s * (EP * W * 64);
        if (W == 1) {
            double buf[EP];
#pragma unroll
            for (int e = 0; e < EP; ++e) buf[e] = p[e * 64 + l];
#pragma unroll
            for (int e = 0; e < EP; ++e) q[e * 64 + l] = buf[e] * 1.0000001;
        } else {
            double2 buf[EP];
#pragma unroll
            for (int e = 0; e < EP; ++e) buf[e] = reinterpret_cast<const double2*>(p)[e * 64 + l];
#pragma unroll
            for (int e = 0; e < EP; ++e) { buf[e].x *= 1.0000001; reinterpret_cast<double2*>(q)[e * 64 + l] = buf[e]; }
        }
    }
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    constexpr int E = 63;          // doubles per node (L + G + y)
    const int tiles = 1021, R = 98;
    const size_t n = (size_t)tiles * R * 64 * 64;   // room for EP*W up to 64
    double *src, *dst, *out;
    hipMalloc(&src, n * 8); hipMalloc(&dst, n * 8); hipMalloc(&out, tiles * 64 * 8);
    hipMemset(src, 0, n * 8);
    const double gb = (double)tiles * R * 64 * E * 8 / 1e9;
    float t;
    t = timeit([&] { hipLaunchKernelGGL((k_read<E, 1>), dim3(tiles), dim3(64), 0, 0, src, out, R); }, 10);
    printf("read  8B/lane : %.3f ms  %.2f TB/s\n", t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL((k_read<E, 2>), dim3(tiles), dim3(64), 0, 0, src, out, R); }, 10);
    printf("read 16B/lane : %.3f ms  %.2f TB/s (payload incl. 1 pad double: %.2f)\n", t, gb / t, gb * 64 / 63 / t);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<E, 1>), dim3(tiles), dim3(64), 0, 0, src, dst, R); }, 10);
    printf("copy  8B/lane : %.3f ms  %.2f TB/s (r+w)\n", t, 2 * gb / t);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<E, 2>), dim3(tiles), dim3(64), 0, 0, src, dst, R); }, 10);
    printf("copy 16B/lane : %.3f ms  %.2f TB/s (r+w)\n", t, 2 * gb / t);
    // more waves: 4 per SIMD
    t = timeit([&] { hipLaunchKernelGGL((k_copy<E, 1>), dim3(tiles * 4), dim3(64), 0, 0, src, dst, R / 4); }, 10);
    printf("copy  8B/lane, 4x waves, R/4: %.3f ms  %.2f TB/s (r+w)\n", t, 2 * gb * (R / 4 * 4) / R / t);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<E, 2>), dim3(tiles * 4), dim3(64), 0, 0, src, dst, R / 4); }, 10);
    printf("copy 16B/lane, 4x waves, R/4: %.3f ms  %.2f TB/s (r+w)\n", t, 2 * gb * (R / 4 * 4) / R / t);
    return 0;
}
