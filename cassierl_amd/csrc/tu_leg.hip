// tu_leg.hip -- translation unit of the two-lanes-per-environment PD / torque / record-command kernels (cassie_kernels_leg.hip).
#include "cassie_kernels_leg.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

void step_leg(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending) {
  dim3 grid((n_envs + 31) / 32), block(64);
  if (mode == 0) hipLaunchKernelGGL((leg::env_step_leg_kernel<0>), grid, block, 0, s, p, pending);
  else if (mode == 1) hipLaunchKernelGGL((leg::env_step_leg_kernel<1>), grid, block, 0, s, p, pending);
  else hipLaunchKernelGGL((leg::env_step_leg_kernel<2>), grid, block, 0, s, p, pending);
}

}  // namespace launch
}  // namespace cassie
