// tu_duo.hip -- translation unit of the 64-environments-per-wavefront PD / torque / record-command kernels (cassie_kernels_duo.hip).
#include "cassie_kernels_duo.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

size_t duo_workspace_bytes(int n_envs) { return (size_t)(((n_envs + 63) / 64 + DUO_WAVES - 1) / DUO_WAVES * DUO_WAVES) * leg::duo_workspace_doubles_per_wave * sizeof(double); }

void step_duo(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending, double* workspace) {
  const int waves = (n_envs + 63) / 64;
  dim3 grid((waves + DUO_WAVES - 1) / DUO_WAVES), block(64 * DUO_WAVES);
  if (mode == 0) hipLaunchKernelGGL((leg::env_step_duo_kernel<0>), grid, block, 0, s, p, pending, workspace);
  else if (mode == 1) hipLaunchKernelGGL((leg::env_step_duo_kernel<1>), grid, block, 0, s, p, pending, workspace);
  else hipLaunchKernelGGL((leg::env_step_duo_kernel<2>), grid, block, 0, s, p, pending, workspace);
}

}  // namespace launch
}  // namespace cassie
