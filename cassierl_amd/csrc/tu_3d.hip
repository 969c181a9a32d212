// tu_3d.hip -- translation unit of the Cassie3d kernels (cassie3d_kernels.hip).
#include "cassie_kernels.hip"
#include "cassie3d_kernels.hip"
#include "cassie3d_pair.hip"
#include "cassie3d_leg.hip"
#include "cassie_launch.h"

#ifndef C3_FAST_WPS
#define C3_FAST_WPS 3   // wavefronts per SIMD the <= 32-row kernel's register budget is sized for (LDS allows 12 per CU)
#endif

namespace cassie3d {
namespace launch {

void step3d(int variant, int n_envs, hipStream_t s, const Params3& p) {
  dim3 grid(n_envs), block(64);
  if (variant == 3) hipLaunchKernelGGL(leg::env_step3d_leg_kernel<32>, dim3((n_envs + 15) / 16), block, 0, s, p);   // one lane per leg, 16 environments per wavefront (half of it idle: four wavefronts per CU)
  else if (variant == 4) hipLaunchKernelGGL(leg::env_step3d_leg_kernel<64>, dim3((n_envs + 31) / 32), block, 0, s, p);   // ... 32 environments per wavefront (two per CU): CASSIE3D_LEG=64
  else if (variant == 2) hipLaunchKernelGGL(env_step3d_pair_kernel, dim3((n_envs + 1) / 2), block, 0, s, p);   // two environments per wavefront
  else if (variant == 0) hipLaunchKernelGGL((env_step3d_kernel<MAXR_FAST, C3_FAST_WPS>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((env_step3d_kernel<MAXR, 1>), grid, block, 0, s, p);
}
void init3d(int n_envs, hipStream_t s, double* state, const double* qpos, const double* qvel) {
  hipLaunchKernelGGL(env_init3d_kernel, dim3(n_envs), dim3(64), 0, s, state, n_envs, qpos, qvel);
}

}  // namespace launch
}  // namespace cassie3d

#ifdef CASSIE3D_PHASE_TIMING
// profiling builds only; not part of the public ABI: the phase clocks of env_step3d_leg_kernel since the last call
extern "C" int Cassie3dDebugPhaseCycles(unsigned long long* out16) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(cassie3d::leg::k5c_phase), 16 * sizeof(unsigned long long)) != hipSuccess) return -2;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(cassie3d::leg::k5c_phase), z, sizeof z) == hipSuccess ? 0 : -3;
}
#endif
