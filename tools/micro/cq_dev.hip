// Development harness: instantiates the level-0 cq kernels for d = 6 alone, so that `hipcc -S --cuda-device-only` on this file gives
// their ISA / register usage in seconds instead of the minutes the whole sweeps unit takes (tools/isa_regs.py reads the listing).
#include "../../vi-diffusion-processes_amd/csrc/mfgm_internal.h"
#include "../../vi-diffusion-processes_amd/csrc/mfgm_cq.h"
using namespace mfgm;
void inst(SweepArgs a, CqArgs q, SdeParams pr, double* fix) {
    hipLaunchKernelGGL((k_reduce_cq<6>), dim3(1), dim3(64), 0, 0, a, q);
    hipLaunchKernelGGL((k_reduce_cq_lean<6>), dim3(1), dim3(64), 0, 0, a, q);
    hipLaunchKernelGGL((k_forward_cq<6>), dim3(1), dim3(64), 0, 0, a, q);
    hipLaunchKernelGGL((k_forward_cq<6, 2>), dim3(1), dim3(64), 0, 0, a, q);
    hipLaunchKernelGGL((k_backward_kl_cq<6, 2>), dim3(1), dim3(64), 0, 0, a, pr, q);
    hipLaunchKernelGGL((k_forward_reduce_cq<6>), dim3(1), dim3(128), 0, 0, a, q);
    hipLaunchKernelGGL((k_backward_girsanov_cq<6>), dim3(1), dim3(64), 0, 0, a, pr, q, fix);
    hipLaunchKernelGGL((k_backward_kl_cq<6>), dim3(1), dim3(64), 0, 0, a, pr, q);
}
