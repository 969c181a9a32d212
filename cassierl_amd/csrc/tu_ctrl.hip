// tu_ctrl.hip -- translation unit of the wave-per-environment controller kernels (cassie_ctrl.hip).
#include "cassie_kernels.hip"
#include "cassie_ctrl.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

void ctrl_k4(int ctrl, bool scripted, int n_envs, hipStream_t s, const VecParams& p, const double* zpos, const double* zvel) {
  dim3 grid(n_envs), block(64);
  if (ctrl == 2) {
    if (scripted) hipLaunchKernelGGL((env_ctrl_kernel<2, true>), grid, block, 0, s, p, zpos, zvel);
    else hipLaunchKernelGGL((env_ctrl_kernel<2, false>), grid, block, 0, s, p, zpos, zvel);
  } else {
    if (scripted) hipLaunchKernelGGL((env_ctrl_kernel<3, true>), grid, block, 0, s, p, zpos, zvel);
    else hipLaunchKernelGGL((env_ctrl_kernel<3, false>), grid, block, 0, s, p, zpos, zvel);
  }
}

}  // namespace launch
}  // namespace cassie
