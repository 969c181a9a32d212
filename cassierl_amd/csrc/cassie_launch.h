// cassie_launch.h -- host-callable launchers of the kernels, one translation unit per kernel family so that the
// extension builds in parallel (tu_*.hip); cassie_cabi.hip (the C-ABI) only sees these declarations.
#ifndef CASSIE_LAUNCH_H_
#define CASSIE_LAUNCH_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cassie_vec_layout.h"
#include "cassie3d_layout.h"

namespace cassie {
namespace launch {

// K1 instantiations of env_step_kernel<MODE, WPS, MAXACT>.  The general kernel is sized for ONE wavefront per SIMD (312 registers,
// no scratch): its product role is the hand-over pass, where nearly every wavefront exits at once and occupancy buys nothing,
// while the 168-register build it replaced spilled 608 B per lane (r01/r02 PMC).
enum K1Variant { K1_DEEP = 0 /* <.,1,32> */, K1_DEBUG = 2 /* <.,2,8>: forces the workspace path */ };
constexpr int K1_MAXACT = 32, K1_MAXACT_DBG = 8;
constexpr int K1_HANDOVER_SPLIT = 8;  // workgroups that share the pending environments of one 64-environment block (hand-over pass while robots are down)

// tu_base.hip: wave-per-environment kernels (mode: 0 PD, 1 torque, 2 motor commands from the state record; K1_DEBUG: modes 0, 1)
void step_k1(int mode, K1Variant variant, int n_envs, hipStream_t s, const VecParams& p, int split = 1);  // split: workgroups sharing the pending envs of a 64-block (hand-over pass)
void reset(int n_envs, hipStream_t s, const VecParams& p, const uint8_t* mask, const double* qpos, const double* qvel);
// tu_hf.hip: the same kernels with the height-field collision stage (p.hf.h != null); PD / torque modes
void step_k1_hf(int mode, int n_envs, hipStream_t s, const VecParams& p);
void step_g16_hf(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending);
void reset_hf(int n_envs, hipStream_t s, const VecParams& p, const uint8_t* mask, const double* qpos, const double* qvel);
void opstate(int n_envs, hipStream_t s, const VecParams& p, double* out18);
void init_state(int n_envs, hipStream_t s, double* state);
void get_state(int n_envs, hipStream_t s, const double* state, double* qpos, double* qvel);
void accumulate_returns(int n_envs, hipStream_t s, const double* reward, const uint8_t* done, double* returns, unsigned long long* episodes);
// tu_g16.hip: four environments per wavefront
void step_g16(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending);
// tu_leg.hip: two lanes per environment (one per leg), 32 environments per wavefront; environments that need more than 8 rows on
// a leg are handed on through `pending` (step_g16 with p.pending = that array, then step_k1)
void step_leg(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending);
// tu_duo.hip: the same tier with 64 environments per wavefront (two groups set up lane-per-leg, one joint sweep with a lane per
// environment; cassie_duo_core.h): bit-identical results, same hand-over through `pending`
// workspace: per wavefront SLOT.  table_slots == 0: slot = task (batches of up to one round of the chip); else every wavefront claims a slot from a
// table of table_slots busy words behind the workspaces, first probe = hash of its physical place (DuoSlots, cassie_kernels_duo.hip): the
// workspace is sized by the chip, not by the batch.  flat_hint (tests): every wavefront starts probing at word 0.
void step_duo(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending, double* workspace, int table_slots, bool flat_hint);
void step_duo_hf(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending, double* workspace, int table_slots, bool flat_hint);   // tu_duo_hf.hip: ... on the height field (p.hf)
int duo_table_slots(int n_envs, int simds);                  // 0 up to 1024 tasks, else a power of two >= 2 x simds
int duo_workspace_slots_per_wave();                          // Duo::W_N: slots of 64 doubles per wavefront slot (diagnosis: CassieVecTierInfo)
size_t duo_workspace_bytes(int n_envs, int table_slots);     // workspaces + claim table; to be zeroed once by the owner
// ... for a SEGMENT of the Env.step's substeps (p.n_sub = its length; `later` = substeps of the segments behind it; gone[env]: the
// environment left this tier in an earlier segment; see env_step_leg_seg_kernel)
void step_leg_segment(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending, int* gone, bool first, int later);
// CassieVecReset through the same core: need_slow[env] = 1 where the state needs the wave-per-environment reset kernel (its mask)
void reset_leg(int n_envs, hipStream_t s, const VecParams& p, const uint8_t* mask, const double* qpos, const double* qvel, uint8_t* need_slow);
// tags the pending environments the 4-envs-per-wave kernel could not hold either (PENDING_DEEP): they go straight to step_k1
void classify_pending(int n_envs, hipStream_t s, const VecParams& p, int* pending);
void step_leg_hf(int mode, int n_envs, hipStream_t s, const VecParams& p, int* pending);   // ... on the height field (p.hf)
// tu_ctrl.hip / tu_ctrl_g16.hip: controllers (ctrl: 2 OSC, 3 Jacobian): they write the motor commands into the state record
void ctrl_k4(int ctrl, bool scripted, int n_envs, hipStream_t s, const VecParams& p, const double* zpos, const double* zvel);
// (the packed controller kernel only writes the motor commands; step_g16 / step_k1 with mode 2 then do the mj_step)
void ctrl_g16(int ctrl, bool scripted, int n_envs, hipStream_t s, const VecParams& p, const double* zpos, const double* zvel);

}  // namespace launch
}  // namespace cassie

namespace cassie3d {
namespace launch {
// tu_3d.hip
void step3d(int variant /*0: <MAXR_FAST,3>, 1: <MAXR,1>, 2: two environments per wavefront, 3: one lane per leg (32 per wavefront)*/, int n_envs, hipStream_t s, const Params3& p);
void init3d(int n_envs, hipStream_t s, double* state, const double* qpos, const double* qvel);
}  // namespace launch
}  // namespace cassie3d
#endif
