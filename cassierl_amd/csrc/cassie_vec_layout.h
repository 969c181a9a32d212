// cassie_vec_layout.h -- HBM layout of one environment and the kernel parameter block
// (shared by cassie_kernels.hip and cassie_cabi.hip; not a public header).
#ifndef CASSIE_VEC_LAYOUT_H_
#define CASSIE_VEC_LAYOUT_H_
#include <stdint.h>

namespace cassie {

// One environment = 88 doubles (704 B), array-of-structures so that the owning wavefront reads
// and writes it with two coalesced wave accesses (lanes 0..63, then lanes 0..23).
//   [ 0..12] qpos   [13..25] qvel   [26..38] qacc_warmstart
//   [39..51] qpos at the last DynamicModel::setState   [52..64] qvel at the last setState   (quirk Q1/Q2)
//   [65..77] self.qstate positions (written by reset only; quirk Q3)
//   [78..83] last mj_data->ctrl (pre-clamp)   [84] env time   [85] PGS iterations of the last call
//   [86] OSC QP working set of the last StepOsc (hot start, 0 = cold; written by the OSC kernels only)   [87] pad
constexpr int ENV_STRIDE = 88;
enum { ES_Q = 0, ES_V = 13, ES_WS = 26, ES_KQ = 39, ES_KV = 52, ES_QSTATE = 65, ES_CTRL = 78, ES_TIME = 84, ES_NITER = 85, ES_QPWSET = 86 };

// debug record (doubles) written by substep() when VecParams.debug != nullptr (tests only)
enum {
  DBG_M = 0, DBG_BIAS = 169, DBG_QS = 182, DBG_F0 = 195, DBG_B = 241, DBG_R = 287, DBG_AREF = 333, DBG_ADIAG = 379,
  DBG_F = 425, DBG_QACC = 471, DBG_QACCH = 484, DBG_STRIDE = 512
};

// 4 = CASSIE_WAVE_PER_ENV (host side: skip the 4-envs-per-wave kernels); 8 = tests only: take the literal SVD / eigen-decomposition
// route of the controllers' pseudo-inverses even where the certified shortcut applies
enum { FLAG_FIX_STALE_KIN = 1, FLAG_FIX_STALE_QSTATE = 2, FLAG_NO_PINV_SHORTCUT = 8 };

// Hand-over between the kernel tiers: pending[env] = substeps the tier above left undone (0 normally), plus PENDING_DEEP when
// the environment has more rows than the 4-environments-per-wavefront kernel holds (classify_pending_kernel), so that the two lower tiers can take
// their environments at the same time (VecParams::pending_pick) instead of one after the other.
enum { PENDING_DEEP = 1 << 30, PENDING_COUNT = PENDING_DEEP - 1 };
enum { PICK_ALL = 0, PICK_DEEP = 1, PICK_SHALLOW = 2 };
#ifdef __HIPCC__
__host__ __device__
#endif
inline int pending_count(int v, int pick) {
  const bool deep = (v & PENDING_DEEP) != 0;
  return ((pick == PICK_DEEP && !deep) || (pick == PICK_SHALLOW && deep)) ? 0 : (v & PENDING_COUNT);
}

// Event counters (CassieVecGetCounters).  Only rare paths touch them, so the common case issues no atomics.
//   STAT_CLEANUP_SUBSTEPS  env-substeps the packed fast-path kernels handed to a slower general kernel (any tier)
//   STAT_K1_SUBSTEPS       ... of those, env-substeps that went all the way to the wave-per-environment kernel
//   STAT_NONFINITE         environments whose state left the finite range (|q|,|v| <= 1e10, NaN) and were force-terminated
//   STAT_WS_PROBES         extra probes of the 64-environments kernel's workspace claims (DuoSlots::claim): 0 while the physical-place hash is collision-free
enum { STAT_CLEANUP_SUBSTEPS = 0, STAT_K1_SUBSTEPS = 1, STAT_NONFINITE = 2, STAT_WS_PROBES = 3, STAT_N = 4 };
constexpr double FINITE_BOUND = 1e10;  // mjMAXVAL of MuJoCo's mj_checkPos / mj_checkVel

// Height-field terrain (rllab/envs/terrain_random.py): heights in metres, [nrow][ncol] row-major in HBM
// (row r at y = -sy + r * 2 sy / (nrow - 1), column c likewise in x).  h == null: flat floor.
struct Terrain {
  const double* h;
  int nrow, ncol;
  double sx, sy;
  double hmax;   // largest height of the field: a sphere whose lowest point is above it touches nothing (early exit of terrain_sphere)
};

struct VecParams {
  double* state;          // [n_envs][ENV_STRIDE]
  const double* actions;  // [n_envs][adim] device
  double* obs;            // [n_envs][26] or null (physics only)
  double* reward;         // [n_envs]
  uint8_t* done;          // [n_envs]
  double* terminal_obs;   // [n_envs][26] or null
  const double* traj_qpos;  // [traj_n][13]
  double traj_tmax;
  int traj_n;
  double* debug;          // [n_envs][DBG_STRIDE] or null
  double* ovf;            // [n_envs][ovf_stride]: A columns beyond the register-resident ones (rare slow path)
  int ovf_stride;
  const int* pending;     // [n_envs] substeps left per env (hand-over input of a lower kernel tier) or null
  int pending_pick;       // which entries of `pending` this launch takes: PICK_ALL, PICK_DEEP (flagged PENDING_DEEP only), PICK_SHALLOW
  int* deep_hint;         // host-visible word or null: `serial` is stored there whenever an environment needs the wave-per-environment kernel
  int serial;             // launch counter of the handle (scheduling hint only, see launch_physics_tiers)
  unsigned* pend_count;   // device [130]: [0] running sum of the current classify_pending launch, [1] arrival ticket of its sampling workgroups (both zero between launches), [2..65] per-serial sums, [66..129] their serials
  unsigned* pend_hint;    // host-visible [64]: estimated hand-overs of launch `serial` in word serial & 63 (classify_pending_kernel's last sampling workgroup stores the launch's total there: a plain store; scheduling hint only) or null
  Terrain hf;             // terrain under the robots (PD / torque modes); hf.h == null: the flat floor of the MJCF
  unsigned long long* phase;  // profiling builds only (-DCASSIE_PHASE_TIMING): [16] shader cycles accumulated per code phase
  unsigned* qp_stats;         // [3 n_envs] or null: per environment sum / maximum of the OSC QP's active-set iterations and StepOsc calls (CassieVecQpIterations)
  unsigned long long* stats;  // [STAT_N] event counters of this handle (rare-path atomics only), see STAT_*
  int n_envs, adim, n_sub, flags, env_kind, auto_reset;
};

}  // namespace cassie
#endif
