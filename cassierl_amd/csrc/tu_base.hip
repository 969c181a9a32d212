// tu_base.hip -- translation unit of the wave-per-environment kernels (cassie_kernels.hip) and the small state kernels.
#define CASSIE_TU_BASE
#include "cassie_kernels.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

void step_k1(int mode, K1Variant variant, int n_envs, hipStream_t s, const VecParams& p, int split) {
  dim3 grid(p.pending ? (n_envs + 63) / 64 : n_envs, p.pending ? split : 1), block(64);  // hand-over pass: one workgroup scans 64 pending counts
  if (variant == K1_DEBUG) {
    if (mode == 0) hipLaunchKernelGGL((env_step_kernel<0, 2, K1_MAXACT_DBG>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((env_step_kernel<1, 2, K1_MAXACT_DBG>), grid, block, 0, s, p);
  } else {
    if (mode == 0) hipLaunchKernelGGL((env_step_kernel<0, 1, K1_MAXACT>), grid, block, 0, s, p);
    else if (mode == 1) hipLaunchKernelGGL((env_step_kernel<1, 1, K1_MAXACT>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((env_step_kernel<2, 1, K1_MAXACT>), grid, block, 0, s, p);
  }
}
void reset(int n_envs, hipStream_t s, const VecParams& p, const uint8_t* mask, const double* qpos, const double* qvel) {
  hipLaunchKernelGGL(env_reset_kernel<false>, dim3(n_envs), dim3(64), 0, s, p, mask, qpos, qvel);
}
void opstate(int n_envs, hipStream_t s, const VecParams& p, double* out18) {
  hipLaunchKernelGGL(env_opstate_kernel, dim3(n_envs), dim3(64), 0, s, p, out18);
}
void init_state(int n_envs, hipStream_t s, double* state) {
  hipLaunchKernelGGL(env_init_kernel, dim3((n_envs * ENV_STRIDE + 255) / 256), dim3(256), 0, s, state, n_envs);
}
void accumulate_returns(int n_envs, hipStream_t s, const double* reward, const uint8_t* done, double* returns, unsigned long long* episodes) {
  hipLaunchKernelGGL(accumulate_returns_kernel, dim3((n_envs + 1023) / 1024), dim3(1024), 0, s, reward, done, returns, episodes, n_envs);
}
void get_state(int n_envs, hipStream_t s, const double* state, double* qpos, double* qvel) {
  hipLaunchKernelGGL(get_state_kernel, dim3((n_envs * 13 + 255) / 256), dim3(256), 0, s, state, n_envs, qpos, qvel);
}

}  // namespace launch
}  // namespace cassie
