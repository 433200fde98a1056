// Micro-benchmark: cost of the in-row broadcast primitives a lone wavefront can use (cycles per operation, dependent chain and
// independent stream), next to a plain v_fma_f64.  hipcc --offload-arch=gfx950 -O3 -o dpp_rate dpp_rate.hip && ./dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int J> __device__ inline double b64(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + J, 0xF, 0xF, true); }
template <int J> __device__ inline double b32x2(double x) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + J, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + J, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double swz(double x) {   // ds_swizzle bit mode: lane (l & 0x18) | 3 of each group of 8
    int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), 0x18 | (3 << 5));
    int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), 0x18 | (3 << 5));
    return __hiloint2double(hi, lo);
}
__device__ inline double rdl(double x) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 3);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k(double* out, long long* cyc, int n) {
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c = 1e-9;
    double b0 = a0 * 0.5, b1 = a1 * 0.5, b2 = a2 * 0.5, b3 = a3 * 0.5, d0 = 1e-7 * threadIdx.x;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7, fc = 1e-6f;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (MODE == 0) { a0 = __builtin_fma(a0, c, a1); a1 = __builtin_fma(a1, c, a2); a2 = __builtin_fma(a2, c, a3); a3 = __builtin_fma(a3, c, a0); }
        if (MODE == 1) { a0 = b64<3>(a0) + c; a1 = b64<5>(a1) + c; a2 = b64<1>(a2) + c; a3 = b64<2>(a3) + c; }
        if (MODE == 2) { a0 = b32x2<3>(a0) + c; a1 = b32x2<5>(a1) + c; a2 = b32x2<1>(a2) + c; a3 = b32x2<2>(a3) + c; }
        if (MODE == 3) { a0 = swz(a0) + c; a1 = swz(a1) + c; a2 = swz(a2) + c; a3 = swz(a3) + c; }
        if (MODE == 4) { a0 = rdl(a0) + c; a1 = rdl(a1) + c; a2 = rdl(a2) + c; a3 = rdl(a3) + c; }
        if (MODE == 5) { a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; }
        if (MODE == 7) {
            a0 = __builtin_fma(a0, c, d0); a1 = __builtin_fma(a1, c, d0); a2 = __builtin_fma(a2, c, d0); a3 = __builtin_fma(a3, c, d0);
            b0 = __builtin_fma(b0, c, d0); b1 = __builtin_fma(b1, c, d0); b2 = __builtin_fma(b2, c, d0); b3 = __builtin_fma(b3, c, d0);
        }
        if (MODE == 8) {   // 8 independent chains, three distinct register operands each
            a0 = __builtin_fma(a0, b0, d0); a1 = __builtin_fma(a1, b1, d0); a2 = __builtin_fma(a2, b2, d0); a3 = __builtin_fma(a3, b3, d0);
            b0 = __builtin_fma(b0, a1, d0); b1 = __builtin_fma(b1, a2, d0); b2 = __builtin_fma(b2, a3, d0); b3 = __builtin_fma(b3, a0, d0);
        }
        if (MODE == 9) {   // fp32 for reference
            f0 = __builtin_fmaf(f0, fc, f1); f1 = __builtin_fmaf(f1, fc, f2); f2 = __builtin_fmaf(f2, fc, f3); f3 = __builtin_fmaf(f3, fc, f0);
            f4 = __builtin_fmaf(f4, fc, f5); f5 = __builtin_fmaf(f5, fc, f6); f6 = __builtin_fmaf(f6, fc, f7); f7 = __builtin_fmaf(f7, fc, f4);
        }
        if (MODE == 6) { a0 = __builtin_fma(a0, c, a0); a0 = __builtin_fma(a0, c, a0); a0 = __builtin_fma(a0, c, a0); a0 = __builtin_fma(a0, c, a0); }
    }
      }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* what, int waves_per_block) {
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 1024 * 8);
    const int n = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves_per_block), 0, 0, out, cyc, n);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves_per_block), 0, 0, out, cyc, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[64]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-52s waves/WG %2d: %7.2f ns, %6.1f shader cycles per group (16 groups per loop trip)\n", what, waves_per_block,
           1e6 * ms / n / 16, (double)h[0] / n / 16);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 4, 8, 16}) {
        run<7>("8 independent v_fma_f64 (a*c+d, 2 reg + 1 shared)", w);
        run<8>("8 independent v_fma_f64 (3 distinct regs)", w);
        run<9>("8 v_fma_f32", w);
        run<0>("4 independent v_fma_f64", w);
        run<6>("4 DEPENDENT v_fma_f64", w);
        run<5>("4 v_add_f64", w);
        run<1>("4 x (v_mov_b64_dpp row_newbcast + add)", w);
        run<2>("4 x (2 v_mov_b32_dpp row_newbcast + add)", w);
        run<3>("4 x (2 ds_swizzle + add)", w);
        run<4>("4 x (2 v_readlane + add)", w);
    }
    return 0;
}
