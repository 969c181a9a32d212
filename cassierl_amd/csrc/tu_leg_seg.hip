// tu_leg_seg.hip -- translation unit of the SEGMENT form of the two-lanes-per-environment kernel (cassie_kernels_leg.hip,
// env_step_leg_seg_kernel): its own unit so that tu_leg.hip -- the headline kernel -- compiles to exactly what it was.
#define CASSIE_LEG_SEGMENT 1
#include "cassie_kernels_leg.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

void step_leg_segment(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending, int* gone, bool first, int later) {
  dim3 grid((n_envs + 31) / 32), block(64);
  const leg::Segment seg{first ? 1 : 0, later};
  if (mode == 0) hipLaunchKernelGGL((leg::env_step_leg_seg_kernel<0>), grid, block, 0, s, p, pending, gone, seg);
  else if (mode == 1) hipLaunchKernelGGL((leg::env_step_leg_seg_kernel<1>), grid, block, 0, s, p, pending, gone, seg);
  else hipLaunchKernelGGL((leg::env_step_leg_seg_kernel<2>), grid, block, 0, s, p, pending, gone, seg);
}

void reset_leg(int n_envs, hipStream_t s, const VecParams& p, const uint8_t* mask, const double* qpos, const double* qvel, uint8_t* need_slow) {
  hipLaunchKernelGGL(leg::env_reset_leg_kernel, dim3((n_envs + 31) / 32), dim3(64), 0, s, p, mask, qpos, qvel, need_slow);
}

}  // namespace launch
}  // namespace cassie
