/* cassie_vec.h -- batched C-ABI of libcassie2d.so (new entry points).
 *
 * The reference steps ONE robot per ctypes call (rllab/envs/cassie2d.py:115-122 crosses the
 * boundary ten times per Env.step).  These entry points are what a vectorised caller binds
 * instead: N independent Cassie2d instances resident in HBM, one launch per Env.step.
 * They replace, for a whole batch:
 *   CassieVecReset     Cassie2dEnv.reset  (cassie2d.py:78-95)  -> Reset + GetOperationalSpaceState
 *   CassieVecStep      Cassie2dEnv.step   (cassie2d.py:97-225; cassie_stand2d.py:86-137)
 *   CassieVecSubstep   lib.StepPd / lib.StepTorque called n times (cassie2d.py:115-122)
 *   CassieVecGetState / CassieVecGetOpState   lib.GetGeneralState / lib.GetOperationalSpaceState
 * Plain pointers and sizes only.  Pointers named *_dev are DEVICE pointers (HBM-resident
 * tensors of the caller, e.g. torch.Tensor.data_ptr()); *_host are host pointers and imply a
 * synchronous copy.  All calls return 0 on success or a negative CASSIE_E* code;
 * CassieVecLastError gives the message.  Work is enqueued on the handle's HIP stream.
 */
#ifndef CASSIE_VEC_H_
#define CASSIE_VEC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CASSIE_OK 0
#define CASSIE_EINVAL (-1)
#define CASSIE_EHIP (-2)
#define CASSIE_ENODEVICE (-3)

/* control_mode (module-level `control_mode` string of cassie2d.py:52 / cassie_stand2d.py:50) */
#define CASSIE_CTRL_PD 0
#define CASSIE_CTRL_TORQUE 1
#define CASSIE_CTRL_OSC 2
#define CASSIE_CTRL_JACOBIAN 3 /* lib.StepJacobian (cassie2d.py:331); only through CassieVecSubstep / CassieVecStandingStep */
/* env_kind */
#define CASSIE_ENV_WALK 0  /* rllab/envs/cassie2d.py       */
#define CASSIE_ENV_STAND 1 /* rllab/envs/cassie_stand2d.py */
/* flags: 0 reproduces the reference bit-for-bit including its stale-state quirks */
#define CASSIE_FIX_STALE_KIN 1
#define CASSIE_FIX_STALE_QSTATE 2
#define CASSIE_WAVE_PER_ENV 4 /* use only the wave-per-environment kernels (A/B and cross-check of the 4-envs-per-wave path) */
#define CASSIE_NO_PINV_SHORTCUT 8 /* tests: controllers evaluate pseudoinverse(.,tol) by SVD / eigen-decomposition even where the
                                    certified inverse / normal-equations shortcut applies (same result) */
#define CASSIE_LEG_TIER_OFF 16 /* never start with the two-lanes-per-environment kernel (A/B, cross-check) */
#define CASSIE_LEG_TIER_ON 32  /* always start with it, also below the batch size where it pays (tests) */
#define CASSIE_DUO_TIER_OFF 64 /* never run that tier in its 64-environments-per-wavefront form (A/B, cross-check) */
#define CASSIE_DUO_TIER_ON 128 /* always run it in that form when it is the first tier (tests) */

#define CASSIE_NQ 13
#define CASSIE_NOBS 26
#define CASSIE_STATE_STRIDE 88 /* doubles per environment in the resident state block:
                                 *  0..12 qpos | 13..25 qvel | 26..38 qacc_warmstart | 39..64 qpos,qvel at the last setState (what
                                 *  GetOperationalSpaceState sees) | 65..77 self.qstate of the env | 78..83 last mj_data->ctrl | 84 env time |
                                 *  85 PGS iterations of the last call | 86 OSC QP working set (hot start; 0 = cold) | 87 unused */

typedef struct CassieVec CassieVec;

typedef struct {
  int env_kind;      /* CASSIE_ENV_*  */
  int control_mode;  /* CASSIE_CTRL_* */
  int n_substeps;    /* physics substeps per Env.step (reference default n=10) */
  int flags;         /* CASSIE_FIX_*  */
  int auto_reset;    /* 1: envs that terminate are reset inside the step and return the reset observation */
} CassieVecConfig;

int CassieVecCreate(CassieVec** out, int n_envs, int device, const CassieVecConfig* cfg);
void CassieVecFree(CassieVec* h);
const char* CassieVecLastError(const CassieVec* h);
int CassieVecNumEnvs(const CassieVec* h);
int CassieVecActionDim(const CassieVec* h);
int CassieVecSetStream(CassieVec* h, void* hip_stream); /* synchronises the previous stream; NULL = default stream */
int CassieVecSynchronize(CassieVec* h);

/* Event counters since create / the last CassieVecResetCounters (synchronises the stream):
 *   out4[0] env-substeps requested (n_envs x substeps of every step call)
 *   out4[1] env-substeps the packed fast-path kernels could not do (more constraint rows than they hold) and handed over
 *   out4[2] ... of those, env-substeps done by the wave-per-environment kernel (the slowest tier)
 *   out4[3] environments force-terminated by the failure guard: a state with NaN or |q|,|v| > 1e10 ends the episode
 *           (done = 1, reward 0, observation 0 / the reset observation) instead of poisoning the batch.  The reference
 *           has no such guard (it exits on load failure only, Cassie2d.cpp:49-52); MuJoCo's mj_checkPos/Vel is the model. */
int CassieVecGetCounters(CassieVec* h, uint64_t* out4);
int CassieVecResetCounters(CassieVec* h);

/* Active-set iterations of the OSC QP (StepOsc / the scripted standing controller; BASELINE.md C3 "QP iterations / step"; the reference's
 * qpOASES call has a budget of 100 working-set changes / 500 us on a hot start, OSC_RBDL.cpp:245-246 -- here the active set runs to KKT
 * convergence, at most 60 iterations).  The FIRST call starts the counting (the controllers store nothing before it) and returns zeros; every
 * later call returns the statistics since the previous one and clears them:
 *   out4[0] mean iterations per StepOsc call and environment   out4[1] maximum over all calls   out4[2] calls counted
 *   out4[3] largest per-environment mean */
int CassieVecQpIterations(CassieVec* h, double* out4);

/* Which kernel tier this handle runs first, and the state of its hand-over workspace (diagnostics; bench.py labels its roofline with it):
 *   out8[0] first physics tier: 0 one wavefront per environment, 1 four environments per wavefront, 2 two lanes per environment
 *           (env_step_leg_kernel), 3 64 environments per wavefront (env_step_duo_kernel) -- chosen at create by batch size / flags / environment
 *   out8[1] claim-table slots of the 64-environments kernel's workspace (0: one slot per 64-environment task, batches of up to one round of the chip)
 *   out8[2] bytes of that workspace
 *   out8[3] extra probes of its workspace claims since create / CassieVecResetCounters (0 while the physical-place hash is collision-free)
 *   out8[4] hand-overs per launch, as the segment scheduler currently estimates them
 *   out8[5] slots (of 64 doubles) per wavefront slot of that workspace (layout: Duo::W_* in csrc/cassie_duo_core.h)
 *   out8[6] environments [0, out8[6]) run in the 64-environments kernel, the rest in the two-lanes kernel (0, n, or the whole rounds of a batch the size
 *           rule splits: the two kernels give bit-identical results);  out8[7] reserved (0) */
int CassieVecTierInfo(CassieVec* h, uint64_t* out8);

/* reference-gait table of cassie2d_trajectory.py (time[n], qpos[n][13]); host pointers, copied once */
int CassieVecSetTrajectory(CassieVec* h, const double* time_host, const double* qpos_host, int n);

/* Height-field terrain under every robot of the batch (what rllab/envs/terrain_random.py:38-76 adds to the MJCF as
 * <hfield size="sx sy sz base" file=...> + <geom type="hfield">): heights_host[nrow][ncol] in METRES above the floor (row r at
 * y = -size_y + r * 2 size_y / (nrow - 1), column c likewise in x; host pointer, copied once).  NULL restores the flat floor.
 * With a field set only PD / torque modes step (CassieVecStep / CassieVecSubstep return CASSIE_EINVAL otherwise).
 * Collision model: sphere vs the triangle of the grid cell under its centre, in the robot's sagittal plane. */
int CassieVecSetHeightField(CassieVec* h, const double* heights_host, int nrow, int ncol, double size_x, double size_y);

/* masked reset to the Cassie2dEnv.reset pose; mask_dev == NULL resets every env; obs_dev may be NULL */
int CassieVecReset(CassieVec* h, const uint8_t* mask_dev, double* obs_dev);
/* Cassie2d::Reset with caller-provided states ([n][13] each, device) */
int CassieVecResetTo(CassieVec* h, const uint8_t* mask_dev, const double* qpos_dev, const double* qvel_dev, double* obs_dev);

/* one Env.step for every env: actions [n][adim] -> obs [n][26], reward [n], done [n]; terminal_obs_dev may be NULL */
int CassieVecStep(CassieVec* h, const double* actions_dev, double* obs_dev, double* reward_dev, uint8_t* done_dev,
                  double* terminal_obs_dev);
/* n_sub raw Step{Pd,Torque,Osc,Jacobian} calls per env with a constant action ([n][6], OSC [n][7]), no observation */
int CassieVecSubstep(CassieVec* h, int control_mode, const double* actions_dev, int n_sub);
/* n_sub calls of standing_controller_osc / standing_controller_jacobian (cassie2d.py:263-331) per env:
 * zpos_dev / zvel_dev [n] are the CoM height / vertical velocity targets held during the call (squatting.py:14-16
 * changes them every call, so a squat is n_sub = 1 with new targets per call) */
int CassieVecStandingStep(CassieVec* h, int control_mode, const double* zpos_dev, const double* zvel_dev, int n_sub);

/* Rollout bookkeeping of one Env.step for the whole batch, one launch on the handle's stream: returns_dev[e] += reward_dev[e]
 * (the undiscounted return a sampler sums per path -- rllab's `rollout`, external to the reference tree; the north star gathers
 * these once per rollout batch) and *episodes_dev += number of non-zero done_dev flags.  Either accumulator may be NULL. */
int CassieVecAccumulate(CassieVec* h, const double* reward_dev, const uint8_t* done_dev, double* returns_dev, unsigned long long* episodes_dev);

int CassieVecGetState(CassieVec* h, double* qpos_dev, double* qvel_dev);      /* [n][13] each */
int CassieVecGetOpState(CassieVec* h, double* x18_dev);                      /* [n][18], operational_state_to_array order */
void* CassieVecStatePtr(CassieVec* h);                                       /* device pointer to [n][CASSIE_STATE_STRIDE] */

/* ---- NOT PART OF THE DROP-IN CONTRACT -------------------------------------------------------------------------------------
 * Everything below this line is test / bench plumbing exported from the same library: host-pointer conveniences (synchronous
 * copies around the device entry points above), a per-stage debug record and a timing helper.  A maintainer binding the
 * reference against this library needs none of them (INTEGRATION.md lists the contract); they may change between rounds. */
/* host-pointer conveniences (synchronous); tests only */
int CassieVecStepHost(CassieVec* h, const double* actions_host, double* obs_host, double* reward_host, uint8_t* done_host);
int CassieVecGetStateHost(CassieVec* h, double* qpos_host, double* qvel_host);
int CassieVecSetStateHost(CassieVec* h, const double* state_host /*[n][88]*/);
int CassieVecGetFullStateHost(CassieVec* h, double* state_host /*[n][88]*/);

/* test hook: one substep for every env with a per-stage debug record ([n][512] doubles, host) */
int CassieVecDebugSubstepHost(CassieVec* h, int control_mode, const double* actions_host, double* debug_host);
/* test / diagnosis hook: the hand-over workspace of the 64-environments-per-wavefront kernel as the last launch left it ([slot pair][lane][2]
 * doubles per wavefront slot, Duo::W_* in csrc/cassie_duo_core.h): *n_doubles = its size; out_host may be null (size query) */
int CassieVecDebugWorkspaceHost(CassieVec* h, double* out_host, uint64_t max_doubles, uint64_t* n_doubles);
/* kernel timing helper for bench.py: launches `steps` Env.steps back to back on the handle's stream and
 * returns the average kernel time in milliseconds measured with HIP events on that stream */
int CassieVecTimeSteps(CassieVec* h, const double* actions_dev, int steps, double* obs_dev, double* reward_dev,
                       uint8_t* done_dev, float* avg_ms);

#ifdef __cplusplus
}
#endif
#endif
