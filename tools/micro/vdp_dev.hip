// Development harness: the final VDP Lagrange sweeps for d = 6 alone (ISA / register usage in seconds; tools/isa_regs.py, isa_hist.py).
#include "../../vi-diffusion-processes_amd/csrc/mfgm_internal.h"
#include "../../vi-diffusion-processes_amd/csrc/mfgm_sweeps.h"
#include "../../vi-diffusion-processes_amd/csrc/mfgm_vdp.h"
using namespace mfgm;
void inst(LevelDesc lv, VdpParams pr, double* p, int* c) {
    hipLaunchKernelGGL((k_vdp_lagrange<6, 4>), dim3(1), dim3(64), 0, 0, lv, pr, p, p, p, p, p, p, p, p, p, c, p);
    hipLaunchKernelGGL((k_vdp_lagrange<6, 5>), dim3(1), dim3(64), 0, 0, lv, pr, p, p, p, p, p, p, p, p, p, c, p);
}
void inst2(LevelDesc lv, VdpParams pr, double* p, int* c) {
    hipLaunchKernelGGL((k_vdp_marginals<6, 1>), dim3(1), dim3(64), 0, 0, lv, pr, p, p, p, p, p, p, p);
    hipLaunchKernelGGL((k_vdp_marginals<6, 1, 1>), dim3(1), dim3(64), 0, 0, lv, pr, p, p, p, p, p, p, p);
    hipLaunchKernelGGL((k_vdp_marginals<6, 3, 2>), dim3(1), dim3(64), 0, 0, lv, pr, p, p, p, p, p, p, p, p, p, c, p);
    hipLaunchKernelGGL((k_vdp_marginals<6, 3>), dim3(1), dim3(64), 0, 0, lv, pr, p, p, p, p, p, p, p);
    hipLaunchKernelGGL((k_vdp_lagrange<6, 1>), dim3(1), dim3(64), 0, 0, lv, pr, p, p, p, p, p, p, p, p, p, c, p);
    hipLaunchKernelGGL((k_vdp_lagrange_products<6>), dim3(1), dim3(64), 0, 0, lv, pr, p, p);
}
