// tu_duo_hf.hip -- translation unit of the height-field instantiation of the 64-environments-per-wavefront kernels (cassie_kernels_duo.hip,
// env_step_duo_hf_kernel): its own unit, so that tu_duo.hip -- the headline kernel -- compiles to exactly what it was.
#define CASSIE_LEG_HF
#include "cassie_kernels_duo.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

void step_duo_hf(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending, double* workspace, int table_slots, bool flat_hint) {
  leg::DuoSlots sl;   // (as step_duo, tu_duo.hip)
  sl.busy = table_slots ? reinterpret_cast<unsigned*>(workspace + (size_t)table_slots * leg::duo_workspace_doubles_per_wave) : nullptr;
  sl.mask = table_slots ? (unsigned)table_slots - 1u : 0u;
  sl.flat_hint = flat_hint ? 1u : 0u;
  const int waves = (n_envs + 63) / 64;
  dim3 grid((waves + DUO_WAVES - 1) / DUO_WAVES), block(64 * DUO_WAVES);
  if (mode == 0) hipLaunchKernelGGL((leg::env_step_duo_hf_kernel<0>), grid, block, 0, s, p, pending, workspace, sl);
  else if (mode == 1) hipLaunchKernelGGL((leg::env_step_duo_hf_kernel<1>), grid, block, 0, s, p, pending, workspace, sl);
  else hipLaunchKernelGGL((leg::env_step_duo_hf_kernel<2>), grid, block, 0, s, p, pending, workspace, sl);
}

}  // namespace launch
}  // namespace cassie
