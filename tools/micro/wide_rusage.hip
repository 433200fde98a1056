// resource-usage probe for the wide kernels: hipcc -c -Rpass-analysis=kernel-resource-usage wide_rusage.hip
#include <hip/hip_runtime.h>
#include "../../vi-diffusion-processes_amd/csrc/mfgm_wide.h"
namespace mfgm {
template __global__ void kw_reduce<16, true, true>(WideArgs);
template __global__ void kw_forward<16, true, true, true>(WideArgs);
template __global__ void kw_backward<16, true, true, true>(WideArgs);
template __global__ void kw_reduce<32, true, true>(WideArgs);
template __global__ void kw_forward<32, true, true, true>(WideArgs);
template __global__ void kw_backward<32, true, true, true>(WideArgs);
}
