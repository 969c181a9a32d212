// tu_g16.hip -- translation unit of the four-environments-per-wavefront PD / torque kernels (cassie_kernels_g16.hip).
#define CASSIE_TU_G16
#include "cassie_kernels.hip"
#include "cassie_kernels_g16.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

void step_g16(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending) {
  dim3 grid((n_envs + 3) / 4), block(64);
  if (mode == 0) hipLaunchKernelGGL((g16::env_step_g16_kernel<0>), grid, block, 0, s, p, pending);
  else if (mode == 1) hipLaunchKernelGGL((g16::env_step_g16_kernel<1>), grid, block, 0, s, p, pending);
  else hipLaunchKernelGGL((g16::env_step_g16_kernel<2>), grid, block, 0, s, p, pending);
}

void classify_pending(int n_envs, hipStream_t s, const VecParams& p, int* pending) {
  hipLaunchKernelGGL(g16::classify_pending_kernel, dim3((n_envs + 3) / 4), dim3(64), 0, s, p, pending);
}

}  // namespace launch
}  // namespace cassie
