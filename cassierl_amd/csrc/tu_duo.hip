// tu_duo.hip -- translation unit of the 64-environments-per-wavefront PD / torque / record-command kernels (cassie_kernels_duo.hip).
#include "cassie_kernels_duo.hip"
#include "cassie_launch.h"

namespace cassie {
namespace launch {

// Workspace of a handle: [slots][W_N][64] doubles + the claim table (slots words, zero = free; zeroed once by the owner, every launch leaves
// it zero).  Up to DuoSlots::DIRECT_MAX tasks: one slot per task, no table.  Above: a table of a power of two >= 2 x simds slots.
int duo_table_slots(int n_envs, int simds) {
  const int tasks = ((n_envs + 63) / 64 + DUO_WAVES - 1) / DUO_WAVES * DUO_WAVES;
  if (tasks <= leg::DuoSlots::DIRECT_MAX) return 0;
  int t = 64;
  while (t < 2 * simds) t *= 2;
  return t;
}
int duo_workspace_slots_per_wave() { return leg::DDuo::W_N; }
size_t duo_workspace_bytes(int n_envs, int table_slots) {
  const int tasks = ((n_envs + 63) / 64 + DUO_WAVES - 1) / DUO_WAVES * DUO_WAVES;
  const size_t slots = table_slots ? (size_t)table_slots : (size_t)tasks;
  return slots * leg::duo_workspace_doubles_per_wave * sizeof(double) + (size_t)table_slots * sizeof(unsigned);
}

static leg::DuoSlots duo_slots_of(double* workspace, int table_slots, bool flat_hint) {
  leg::DuoSlots sl;
  sl.busy = table_slots ? reinterpret_cast<unsigned*>(workspace + (size_t)table_slots * leg::duo_workspace_doubles_per_wave) : nullptr;
  sl.mask = table_slots ? (unsigned)table_slots - 1u : 0u;
  sl.flat_hint = flat_hint ? 1u : 0u;
  return sl;
}

void step_duo(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending, double* workspace, int table_slots, bool flat_hint) {
  const leg::DuoSlots sl = duo_slots_of(workspace, table_slots, flat_hint);
  const int waves = (n_envs + 63) / 64;
  dim3 grid((waves + DUO_WAVES - 1) / DUO_WAVES), block(64 * DUO_WAVES);
  if (mode == 0) hipLaunchKernelGGL((leg::env_step_duo_kernel<0>), grid, block, 0, s, p, pending, workspace, sl);
  else if (mode == 1) hipLaunchKernelGGL((leg::env_step_duo_kernel<1>), grid, block, 0, s, p, pending, workspace, sl);
  else hipLaunchKernelGGL((leg::env_step_duo_kernel<2>), grid, block, 0, s, p, pending, workspace, sl);
}

}  // namespace launch
}  // namespace cassie
