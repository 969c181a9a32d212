// tu_hf.hip -- translation unit of the height-field instantiations: the PD / torque / record-command step kernels (the latter = the mj_step of StepOsc / StepJacobian) and the
// reset kernel with the terrain collision stage compiled in.  Separate from tu_base / tu_g16 so that the flat-floor
// kernels (the headline) are byte-for-byte what they were and everything builds in parallel.
#include "cassie_kernels.hip"
#include "cassie_kernels_g16.hip"
#define CASSIE_LEG_HF
#include "cassie_kernels_leg.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

void step_k1_hf(int mode, int n_envs, hipStream_t s, const VecParams& p) {
  dim3 grid(p.pending ? (n_envs + 63) / 64 : n_envs, 1), block(64);
  if (mode == 0) hipLaunchKernelGGL((env_step_kernel<0, 1, K1_MAXACT, true>), grid, block, 0, s, p);
  else if (mode == 1) hipLaunchKernelGGL((env_step_kernel<1, 1, K1_MAXACT, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((env_step_kernel<2, 1, K1_MAXACT, true>), grid, block, 0, s, p);
}
void step_g16_hf(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending) {
  dim3 grid((n_envs + 3) / 4), block(64);
  if (mode == 0) hipLaunchKernelGGL((g16::env_step_g16_kernel<0, true>), grid, block, 0, s, p, pending);
  else if (mode == 1) hipLaunchKernelGGL((g16::env_step_g16_kernel<1, true>), grid, block, 0, s, p, pending);
  else hipLaunchKernelGGL((g16::env_step_g16_kernel<2, true>), grid, block, 0, s, p, pending);
}
void step_leg_hf(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending) {
  dim3 grid((n_envs + 31) / 32), block(64);
  if (mode == 0) hipLaunchKernelGGL((leg::env_step_leg_hf_kernel<0>), grid, block, 0, s, p, pending);
  else if (mode == 1) hipLaunchKernelGGL((leg::env_step_leg_hf_kernel<1>), grid, block, 0, s, p, pending);
  else hipLaunchKernelGGL((leg::env_step_leg_hf_kernel<2>), grid, block, 0, s, p, pending);
}
void reset_hf(int n_envs, hipStream_t s, const VecParams& p, const uint8_t* mask, const double* qpos, const double* qvel) {
  hipLaunchKernelGGL(env_reset_kernel<true>, dim3(n_envs), dim3(64), 0, s, p, mask, qpos, qvel);
}

}  // namespace launch
}  // namespace cassie
