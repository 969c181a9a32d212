// cassie_cabi.hip -- host side of libcassie2d.so: the C-ABI of include/cassie2d.h (legacy, batch-of-one),
// include/cassie_vec.h (batched Cassie2d) and include/cassie3d_vec.h (batched Cassie3d) on top of the kernel launchers
// declared in cassie_launch.h (one translation unit per kernel family, tu_*.hip).
// There is no CPU code path here: every entry point launches HIP kernels or fails.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/cassie2d.h"
#include "../../include/cassie_vec.h"
#include "../../include/cassie3d_vec.h"
#include "cassie2d_planar.h"
#include "cassie3d_tables.h"
#include "cassie_launch.h"

namespace cassie { constexpr int NSLOT = CP_NSLOT; }
namespace L2 = cassie::launch;
namespace L3 = cassie3d::launch;

static_assert(CASSIE_STATE_STRIDE == cassie::ENV_STRIDE, "public stride must match the kernel layout");
static_assert(sizeof(StateGeneral) == 208 && sizeof(StateOperationalSpace) == 144 && sizeof(ControllerOsc) == 56 &&
                  sizeof(ControllerPd) == 48 && sizeof(ControllerTorque) == 48 && sizeof(ControllerForce) == 48,
              "ABI struct sizes (RobotInterface.h:14-50)");

struct CassieVec {
  int n = 0, device = 0;
  CassieVecConfig cfg{};
  hipStream_t stream = nullptr;
  double* state = nullptr;
  double* traj_qpos = nullptr;
  double traj_tmax = 0.0;
  int traj_n = 0;
  cassie::Terrain hf{};  // device height field (N4) or {null}
  // scratch for the host-pointer conveniences
  double *d_act = nullptr, *d_obs = nullptr, *d_rew = nullptr, *d_q = nullptr, *d_v = nullptr, *d_dbg = nullptr;
  double *ovf = nullptr, *ovf_dbg = nullptr;  // workspace for constraint columns beyond the register-resident ones
  int* pending = nullptr;                    // substeps left per env after the 4-envs-per-wave kernel
  int* pending_leg = nullptr;                // substeps left per env after the two-lanes-per-env kernel (input of the 4-envs-per-wave kernel)
  bool leg = true;                           // first tier = the two-lanes-per-environment kernel (CASSIE2D_LEG=0/1 overrides the size rule)
  double* duo_ws = nullptr;                  // workspace of the 64-environments-per-wavefront kernel (L2::duo_workspace_bytes)
  int duo_table = 0;                         // claim-table slots of its workspace (L2::duo_table_slots; 0: one slot per task).  CASSIE2D_DUO_TABLE=<slots> forces a table (tests)
  bool duo_flat_hint = false;                // CASSIE2D_DUO_FLAT_HINT=1 (tests): every wavefront's first probe is word 0
  size_t duo_ws_bytes = 0;
  int duo_envs = 0;                          // environments [0, duo_envs) take that form, the rest the two-lanes kernel (0 or n unless the size rule splits the batch)
  bool duo = false;                          // ... in its 64-environments-per-wavefront form (cassie_kernels_duo.hip; CASSIE2D_DUO=0/1 overrides the size rule)
  unsigned long long* phase = nullptr;       // profiling builds (-DCASSIE_PHASE_TIMING): 16 cycle accumulators
  unsigned long long* stats = nullptr;       // device event counters (cassie::STAT_*)
  unsigned long long substeps_requested = 0; // host: env-substeps asked for since the counters were last cleared
  bool g16 = true;                           // CASSIE2D_G16=0 selects the wave-per-environment kernel only (A/B)
  uint8_t* d_done = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t side = nullptr;                // second stream: the two lower physics tiers run side by side behind the first
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int* deep_hint = nullptr;                  // pinned host word the first tier writes (launch serial of the last deep hand-over)
  int* deep_hint_dev = nullptr;              // ... its device address
  unsigned serial = 64;                      // launches of the physics tiers so far (starts past the hint window); wraps (compared modulo 2^32)
  int side_mode = -1;                        // CASSIE2D_SIDE_BY_SIDE=0/1 (tests): never / always run the lower tiers side by side; -1: by the hint
  // the Env.step in segments while robots are down (launch_physics_tiers): per segment the hand-over lists, two streams, three events
  static constexpr int NSEG = 4;
  int seg_mode = -1;                         // CASSIE2D_SEGMENTS=0/1 (A/B, tests): never / always in segments while robots are down; -1: by the count below
  unsigned* pend_hint = nullptr;             // pinned host words [64]: estimated environments that left the first tier in launch serial & 63 (classify_pending_kernel)
  unsigned* pend_hint_dev = nullptr;
  unsigned* qp_stats = nullptr;              // [3 n]: OSC QP iteration statistics (allocated by the first CassieVecQpIterations call: the controllers only count when someone reads)
  unsigned* pend_count = nullptr;            // device [130]: the launch's running sum and arrival ticket, per-serial sums and tags (classify_pending_kernel)
  unsigned pend_rate = 0;                    // hand-overs per launch, estimated from the last 48 launches' words
  bool reset_packed = true;                  // CASSIE2D_RESET_PACKED=0: CassieVecReset with one wavefront per environment only (A/B, tests)
  uint8_t* need_slow = nullptr;              // [n] written by the packed reset kernel: environments the wave-per-environment reset takes
  bool seg_ready = false, seg_failed = false;
  int* seg_pend[NSEG] = {};                  // [n] substeps left per env after segment j of the two-lanes-per-env kernel
  int* seg_pend2[NSEG] = {};                 // [n] ... after the 4-envs-per-wave kernel took segment j's environments
  int* gone = nullptr;                       // [n] the environment left the first tier in an earlier segment of this Env.step
  hipStream_t seg_deep[NSEG] = {}, seg_shal[NSEG] = {};
  hipEvent_t seg_fork[NSEG] = {}, seg_join_a[NSEG] = {}, seg_join_b[NSEG] = {};
  std::string err;
};

namespace {

int fail(CassieVec* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  return code;
}

#define HIPCHK(h, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return fail(h, CASSIE_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

int adim_of(int mode) { return mode == CASSIE_CTRL_OSC ? 7 : 6; }

constexpr unsigned SEG_MIN_HANDOVERS = 96;   // hand-overs per launch from which the Env.step runs in segments (launch_physics_tiers)
constexpr int DUO_MIN_ENVS = 32768;  // 64 environments per wavefront: never below one full round of the pair form (CassieVecCreate weighs whole rounds above it)
constexpr int LEG_MIN_ENVS = 6144;   // measured crossover (r03, bench workload): 4096 envs 0.63 ms (g16 tier) vs 0.75 ms (leg tier), 8192 envs 0.83 vs 0.74 ms
constexpr int MAXACT = L2::K1_MAXACT;                   // register-resident active constraint columns per row lane
constexpr int OVF_STRIDE = (cassie::NSLOT - MAXACT) * 64;  // doubles per env in the overflow workspace
constexpr int MAXACT_DBG = L2::K1_MAXACT_DBG;           // debug build of the substep: forces the overflow path in tests
constexpr int OVF_STRIDE_DBG = (cassie::NSLOT - MAXACT_DBG) * 64;

cassie::VecParams make_params(CassieVec* h) {
  cassie::VecParams p{};
  p.state = h->state;
  p.n_envs = h->n;
  p.adim = adim_of(h->cfg.control_mode);
  p.n_sub = h->cfg.n_substeps;
  p.flags = h->cfg.flags;
  p.env_kind = h->cfg.env_kind;
  p.auto_reset = h->cfg.auto_reset;
  p.traj_qpos = h->traj_qpos;
  p.traj_tmax = h->traj_tmax;
  p.traj_n = h->traj_n;
  p.ovf = h->ovf;
  p.ovf_stride = OVF_STRIDE;
  p.stats = h->stats;
  p.qp_stats = h->qp_stats;
  p.hf = h->hf;
  p.phase = h->phase;
  return p;
}

// lists, streams and events of the segmented Env.step (first use; false: not available, the one-launch order is used)
bool seg_resources(CassieVec* h) {
  if (h->seg_ready) return true;
  if (h->seg_failed) return false;
  const size_t bytes = (size_t)h->n * sizeof(int);
  bool ok = hipMalloc(&h->gone, bytes) == hipSuccess;
  for (int j = 0; j < CassieVec::NSEG && ok; j++)
    ok = hipMalloc(&h->seg_pend[j], bytes) == hipSuccess && hipMalloc(&h->seg_pend2[j], bytes) == hipSuccess &&
         hipStreamCreateWithFlags(&h->seg_deep[j], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&h->seg_shal[j], hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&h->seg_fork[j], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&h->seg_join_a[j], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&h->seg_join_b[j], hipEventDisableTiming) == hipSuccess;
  if (!ok) { h->seg_failed = true; (void)hipGetLastError(); return false; }
  h->seg_ready = true;
  return true;
}

// The first tier of a launch (flat floor or height field): the first `duo_envs` environments in the 64-environments-per-wavefront kernel, the rest in
// the two-lanes kernel (r06: a batch of whole rounds of the chip plus a remainder of at most one short round takes BOTH -- 65 537 .. 98 304 envs: 1.0 + 0.62 ms
// instead of three short rounds; the two kernels are bit-identical, so which environment runs where is invisible in the results).
void launch_first_tier(CassieVec* h, int mode, const cassie::VecParams& p, bool hf) {
  const int nd = h->duo_envs;
  if (nd > 0) {
    if (hf) L2::step_duo_hf(mode, nd, h->stream, p, h->pending_leg, h->duo_ws, h->duo_table, h->duo_flat_hint);
    else L2::step_duo(mode, nd, h->stream, p, h->pending_leg, h->duo_ws, h->duo_table, h->duo_flat_hint);
  }
  if (nd < h->n) {
    cassie::VecParams q = p;
    if (nd > 0) {   // the remainder: every per-environment array moved on by nd environments
      const size_t o = (size_t)nd;
      q.n_envs = h->n - nd;
      q.state = p.state + o * cassie::ENV_STRIDE;
      if (p.actions) q.actions = p.actions + o * (size_t)p.adim;
      if (p.obs) q.obs = p.obs + o * 26;
      if (p.terminal_obs) q.terminal_obs = p.terminal_obs + o * 26;
      if (p.reward) q.reward = p.reward + o;
      if (p.done) q.done = p.done + o;
    }
    if (hf) L2::step_leg_hf(mode, h->n - nd, h->stream, q, h->pending_leg + nd); else L2::step_leg(mode, h->n - nd, h->stream, q, h->pending_leg + nd);
  }
}

// Physics tiers on the flat floor (mode 0 PD, 1 torque, 2 commands from the record).  Each tier leaves an environment it cannot
// hold untouched from that substep on and says how many substeps are left; the next tier finishes it (results are what that
// tier alone would give: an environment's arithmetic is a function of its own state in every kernel).
//   tier 1  two lanes per environment (<= 8 rows per leg)        cassie_kernels_leg.hip   [h->leg]
//   tier 2  four environments per wavefront (<= 16 rows)          cassie_kernels_g16.hip
//   tier 3  one wavefront per environment (any number of rows)    cassie_kernels.hip
void launch_physics_tiers(CassieVec* h, int mode, const cassie::VecParams& pin) {
  cassie::VecParams p = pin;
  p.deep_hint = h->deep_hint_dev; p.serial = (int)++h->serial;   // the kernels only store the word; the comparison below is unsigned
  p.pend_hint = h->pend_hint_dev; p.pend_count = h->pend_count;
  if (h->pend_hint) {
    // estimated hand-overs of the recent launches, one pinned word per launch (ring of 64, read without synchronisation: a stale value
    // changes the schedule, never a result).  The host runs many launches ahead of the device, so the words of the newest launches
    // are still empty: the statistic is the SUM over the last 48 launches (the ones executed by now carry it), compared with what
    // 48 launches of SEG_MIN_HANDOVERS / 3 would give -- i.e. it tolerates two thirds of the window not having run yet.
    unsigned sum = 0;
    for (unsigned k = 1; k <= 48; k++) sum += ((volatile unsigned*)h->pend_hint)[(h->serial - k) & 63];
    h->pend_rate = sum / 16;
    ((volatile unsigned*)h->pend_hint)[h->serial & 63] = 0;   // this launch's word (last used 64 launches ago)
  }
  cassie::VecParams p2 = p, p3 = p;
  p3.pending = h->pending; p3.pending_pick = cassie::PICK_ALL;
  // A handed-down environment costs its remaining substeps end to end whatever the batch size (~0.09 ms per substep in the middle
  // tier, 0.13-0.18 ms in the last).  While robots are down the lower tiers therefore take their environments AT THE SAME TIME on
  // two streams: a small kernel tags the environments the middle tier could not hold either (PENDING_DEEP), those go straight to
  // the wave-per-environment kernel on the side stream, and the middle tier plus the pass behind it (what it passed on after all)
  // run on the caller's stream -- disjoint sets of environments.  r03 trace of fallen robots: 0.9 + 1.7 ms one after the other
  // before, max(1.7, 0.9 + <= 1.2) ms now.  The tagging kernel and the fork / join cost ~70 us per launch (r03_i: headline 1.43 ->
  // 1.50 ms when they were unconditional), so this order is only used while the wave-per-environment kernel has had work in one
  // of the last 32 launches (a word in pinned host memory it writes, read here without synchronisation: a stale value changes the
  // schedule, never a result -- in the one-stream order the middle tier looks at the deep environments first, finds them too
  // large at the same substep, and passes them on untouched).
  // (unsigned difference: wrap-safe; a launch counter that wrapped past a stale hint turns the order on for at most 32 launches)
  const bool side_by_side = h->leg && (h->side_mode >= 0 ? h->side_mode == 1 : (h->deep_hint && h->serial - (unsigned)*(volatile int*)h->deep_hint <= 32u));
  // fork: the side stream waits for the first tier.  If the event cannot be recorded / waited for, fall back to the one-stream order
  // below (same results) rather than let the side stream's kernel run concurrently with the first tier on the same records.
  // in segments only while MANY robots are down: the three extra launches of the first tier cost ~3 % of a step, and a handful of
  // hand-overs is a short tail (r04, alternating A/B at 65 536 envs: all-fallen floor 15.5 -> 17.0 M with ~270 hand-overs per step;
  // stand_torque_random 22.4 -> 21.6 M when forced; by this rule 22.7 M / 16.9 M)
  // r06: only where the first tier is the two-lanes kernel.  The segments are launches of that kernel (step_leg_segment); since the 64-environments
  // kernel sweeps its eight-row groups with a lane per environment it is ahead of four segments of the pair sweep on both falling-robot workloads
  // (tools/ab_segments.py, 65 536 envs: stand_torque_random 25.9 M never / 22.7 M always / 22.4 M by the estimate; all-fallen floor 17.5 / 17.2 / 17.2)
  const bool segments = h->seg_mode >= 0 ? h->seg_mode == 1 : (!h->duo && h->pend_rate >= SEG_MIN_HANDOVERS);
  if (side_by_side && segments && !h->seg_failed && p.n_sub >= 4 && !p.debug && seg_resources(h)) {
    // ---- the Env.step in segments (r04).  A robot that is down costs its substeps end to end (~0.13 ms each) in the lower tiers,
    // and in the order below these only start when the first tier has finished ALL its substeps: 1.4 ms of first tier + up to
    // 1.3 ms of tail.  Here the first tier runs the step as a first segment of ONE substep and up to three more of equal length;
    // after each, the environments that left it in that segment go to the lower tiers on the segment's own two streams (deep ones
    // straight to the wave-per-environment kernel, the others to the 4-environments-per-wavefront kernel and the pass behind it)
    // and are finished there to the END of the Env.step, while the first tier goes on with the next segment for everyone else
    // (`gone`).  A robot that was already down starts its ten substeps 0.2 ms into the step instead of 1.4; one that goes down in
    // the last segment has at most three substeps left.  Results: every environment is stepped by the same kernels on the same
    // data as in the one-launch order (an environment's arithmetic is a function of its own state in every kernel).
    int len[CassieVec::NSEG], nseg = 0, left = p.n_sub - 1;
    len[nseg++] = 1;
    for (int parts = CassieVec::NSEG - 1; parts > 0; parts--) { const int l = (left + parts - 1) / parts; if (l > 0) { len[nseg++] = l; left -= l; } }
    int later = p.n_sub;
    bool forked[CassieVec::NSEG] = {};
    for (int j = 0; j < nseg; j++) {
      later -= len[j];
      cassie::VecParams ps = p;
      ps.n_sub = len[j];
      if (j != nseg - 1) { ps.obs = nullptr; ps.terminal_obs = nullptr; }
      L2::step_leg_segment(mode, h->n, h->stream, ps, h->seg_pend[j], h->gone, j == 0, later);
      L2::classify_pending(h->n, h->stream, p, h->seg_pend[j]);
      // the segment's hand-overs on its own two streams; if the fork cannot be set up (or one failed before), on the caller's stream,
      // one after the other: the same kernels on the same data, only without the overlap
      const bool fork = !h->seg_failed && hipEventRecord(h->seg_fork[j], h->stream) == hipSuccess &&
                        hipStreamWaitEvent(h->seg_deep[j], h->seg_fork[j], 0) == hipSuccess && hipStreamWaitEvent(h->seg_shal[j], h->seg_fork[j], 0) == hipSuccess;
      if (!fork) h->seg_failed = true;
      hipStream_t sd = fork ? h->seg_deep[j] : h->stream, ss = fork ? h->seg_shal[j] : h->stream;
      cassie::VecParams pd = p, pa = p, pb = p;
      pd.pending = h->seg_pend[j]; pd.pending_pick = cassie::PICK_DEEP;
      L2::step_k1(mode, L2::K1_DEEP, h->n, sd, pd, L2::K1_HANDOVER_SPLIT);
      pa.pending = h->seg_pend[j]; pa.pending_pick = cassie::PICK_SHALLOW;
      L2::step_g16(mode, h->n, ss, pa, h->seg_pend2[j]);
      pb.pending = h->seg_pend2[j]; pb.pending_pick = cassie::PICK_ALL;
      L2::step_k1(mode, L2::K1_DEEP, h->n, ss, pb, L2::K1_HANDOVER_SPLIT);
      forked[j] = fork;
    }
    for (int j = 0; j < nseg; j++) {   // join: the caller's stream continues behind every segment's lower tiers
      if (!forked[j]) continue;
      if (hipEventRecord(h->seg_join_a[j], h->seg_deep[j]) != hipSuccess || hipEventRecord(h->seg_join_b[j], h->seg_shal[j]) != hipSuccess ||
          hipStreamWaitEvent(h->stream, h->seg_join_a[j], 0) != hipSuccess || hipStreamWaitEvent(h->stream, h->seg_join_b[j], 0) != hipSuccess) {
        hipStreamSynchronize(h->seg_deep[j]); hipStreamSynchronize(h->seg_shal[j]);   // the hard way
        h->seg_failed = true;
      }
    }
    return;
  }
  if (side_by_side) {
    launch_first_tier(h, mode, p, false);
    L2::classify_pending(h->n, h->stream, p, h->pending_leg);
    const bool forked = hipEventRecord(h->ev_fork, h->stream) == hipSuccess && hipStreamWaitEvent(h->side, h->ev_fork, 0) == hipSuccess;
    if (forked) {
      cassie::VecParams pd = p;
      pd.pending = h->pending_leg; pd.pending_pick = cassie::PICK_DEEP;
      L2::step_k1(mode, L2::K1_DEEP, h->n, h->side, pd, L2::K1_HANDOVER_SPLIT);
      const bool joined = hipEventRecord(h->ev_join, h->side) == hipSuccess;
      p2.pending = h->pending_leg; p2.pending_pick = cassie::PICK_SHALLOW;
      L2::step_g16(mode, h->n, h->stream, p2, h->pending);
      L2::step_k1(mode, L2::K1_DEEP, h->n, h->stream, p3, L2::K1_HANDOVER_SPLIT);
      if (!joined || hipStreamWaitEvent(h->stream, h->ev_join, 0) != hipSuccess) hipStreamSynchronize(h->side);   // join the hard way
      return;
    }
    // no fork: the tagged environments are finished on the caller's stream (the deep ones first, then the others)
    cassie::VecParams pd = p;
    pd.pending = h->pending_leg; pd.pending_pick = cassie::PICK_DEEP;
    L2::step_k1(mode, L2::K1_DEEP, h->n, h->stream, pd, L2::K1_HANDOVER_SPLIT);
    p2.pending = h->pending_leg; p2.pending_pick = cassie::PICK_SHALLOW;
    L2::step_g16(mode, h->n, h->stream, p2, h->pending);
    L2::step_k1(mode, L2::K1_DEEP, h->n, h->stream, p3, L2::K1_HANDOVER_SPLIT);
    return;
  }
  if (h->leg) {
    launch_first_tier(h, mode, p, false);
    p2.pending = h->pending_leg; p2.pending_pick = cassie::PICK_ALL;
  }
  L2::step_g16(mode, h->n, h->stream, p2, h->pending);
  L2::step_k1(mode, L2::K1_DEEP, h->n, h->stream, p3);
}

// The same tiers on the height field (one stream: robots on terrain have not been profiled lying down).
void launch_physics_tiers_hf(CassieVec* h, int mode, const cassie::VecParams& p) {
  cassie::VecParams p2 = p, p3 = p;
  if (h->leg) {
    launch_first_tier(h, mode, p, true);
    p2.pending = h->pending_leg; p2.pending_pick = cassie::PICK_ALL;
  }
  L2::step_g16_hf(mode, h->n, h->stream, p2, h->pending);
  p3.pending = h->pending; p3.pending_pick = cassie::PICK_ALL;
  L2::step_k1_hf(mode, h->n, h->stream, p3);
}

// controller-in-the-loop modes; zpos/zvel != null selects the scripted standing controllers
int launch_ctrl_step(CassieVec* h, int mode, const cassie::VecParams& p, const double* zpos, const double* zvel) {
  const bool scripted = zpos != nullptr;
  if (mode != CASSIE_CTRL_OSC && mode != CASSIE_CTRL_JACOBIAN)
    return fail(h, CASSIE_EINVAL, "controller-in-the-loop stepping exists for OSC and Jacobian modes only");
  // On terrain (rllab/envs/terrain_random.py rewrites the one MJCF every Step* variant loads) the controllers are what they are on
  // the floor -- OSC_RBDL / StepJacobian work from the RBDL model and the foot SITES, they never see MuJoCo's contacts
  // (OSC_RBDL.cpp:41-71, Cassie2d.cpp:119-209) -- and only the mj_step behind them collides with the height field.
  const int ctrl = mode == CASSIE_CTRL_OSC ? 2 : 3;
  if (h->g16 && !p.debug) {
    // Per StepOsc / StepJacobian: the packed controller kernel writes the motor commands into the state record, the packed
    // physics kernel (mode 2: commands from the record) does the mj_step, and the wave-per-environment physics kernel finishes the
    // (rare) environments with more than 16 constraint rows.  Observation / reward / reset ride on the last substep.
    for (int sub = 0; sub < p.n_sub; sub++) {
      cassie::VecParams ps = p;
      ps.n_sub = 1;
      if (sub != p.n_sub - 1) { ps.obs = nullptr; ps.terminal_obs = nullptr; }
      L2::ctrl_g16(ctrl, scripted, h->n, h->stream, ps, zpos, zvel);
      if (h->hf.h) {
        launch_physics_tiers_hf(h, 2, ps);
      } else {
        launch_physics_tiers(h, 2, ps);
      }
    }
  } else {
    // wave-per-environment kernels only (CASSIE_WAVE_PER_ENV cross-check, debug record): same split, one wavefront per environment
    for (int sub = 0; sub < p.n_sub; sub++) {
      cassie::VecParams ps = p;
      ps.n_sub = 1; ps.pending = nullptr;
      if (sub != p.n_sub - 1) { ps.obs = nullptr; ps.terminal_obs = nullptr; }
      L2::ctrl_k4(ctrl, scripted, h->n, h->stream, ps, zpos, zvel);
      ps.debug = nullptr;
      if (h->hf.h) L2::step_k1_hf(2, h->n, h->stream, ps);
      else L2::step_k1(2, L2::K1_DEEP, h->n, h->stream, ps);
    }
  }
  HIPCHK(h, hipGetLastError());
  return CASSIE_OK;
}

int launch_step(CassieVec* h, int mode, const cassie::VecParams& p) {
  const bool pdtq = mode == CASSIE_CTRL_PD || mode == CASSIE_CTRL_TORQUE;
  if (h->hf.h) {
    // height-field terrain: the 4-environments-per-wavefront and wave-per-environment kernels with the terrain collision stage
    if (!pdtq) return launch_ctrl_step(h, mode, p, nullptr, nullptr);
    if (p.debug) return fail(h, CASSIE_EINVAL, "the debug substep has no height-field variant");
    if (h->g16) {
      launch_physics_tiers_hf(h, mode, p);
    } else {
      L2::step_k1_hf(mode, h->n, h->stream, p);
    }
  } else if (h->g16 && !p.debug && pdtq) {
    // fast path: 4 environments per wavefront; environments with more than 16 active constraint rows are finished
    // by the wave-per-environment kernel, which returns immediately for every other environment
    launch_physics_tiers(h, mode, p);
  } else if (p.debug && pdtq) {
    // test hook: same code with only MAXACT_DBG register-resident columns, so that the workspace path is exercised
    L2::step_k1(mode, L2::K1_DEBUG, h->n, h->stream, p);
  } else if (pdtq) {
    L2::step_k1(mode, L2::K1_DEEP, h->n, h->stream, p);
  }
  else if (mode == CASSIE_CTRL_OSC || mode == CASSIE_CTRL_JACOBIAN) {
    return launch_ctrl_step(h, mode, p, nullptr, nullptr);
  } else return fail(h, CASSIE_EINVAL, "unknown control mode %d", mode);
  HIPCHK(h, hipGetLastError());
  return CASSIE_OK;
}

// reset_mode: 0 = Cassie2dEnv.reset pose, 1 = states from arrays, 2 = keep state (mj_forward only)
int launch_reset(CassieVec* h, const uint8_t* mask, const double* q, const double* v, double* obs, bool keep) {
  cassie::VecParams p = make_params(h);
  p.obs = obs;
  if (keep) {
    // forward only: feed the current state back in as the "new" state
    HIPCHK(h, hipSetDevice(h->device));
    L2::get_state(h->n, h->stream, h->state, h->d_q, h->d_v);
    q = h->d_q; v = h->d_v;
  }
  if (h->hf.h) L2::reset_hf(h->n, h->stream, p, mask, q, v);
  else if (h->g16 && h->reset_packed && h->need_slow) {
    // two lanes per environment, 32 environments per wavefront; a state with more than 8 rows on a leg is left to the
    // wave-per-environment kernel through the mask the packed kernel writes
    L2::reset_leg(h->n, h->stream, p, mask, q, v, h->need_slow);
    L2::reset(h->n, h->stream, p, h->need_slow, q, v);
  } else L2::reset(h->n, h->stream, p, mask, q, v);
  HIPCHK(h, hipGetLastError());
  return CASSIE_OK;
}

}  // namespace

extern "C" {

int CassieVecCreate(CassieVec** out, int n_envs, int device, const CassieVecConfig* cfg) {
  if (!out || n_envs <= 0) return CASSIE_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    fprintf(stderr, "libcassie2d: no HIP device available; this library has no CPU path\n");
    return CASSIE_ENODEVICE;
  }
  if (device < 0 || device >= ndev) return CASSIE_EINVAL;
  CassieVec* h = new CassieVec();
  h->n = n_envs; h->device = device;
  if (cfg) h->cfg = *cfg;
  else { h->cfg.env_kind = CASSIE_ENV_WALK; h->cfg.control_mode = CASSIE_CTRL_PD; h->cfg.n_substeps = 10; h->cfg.flags = 0; h->cfg.auto_reset = 1; }
  if (h->cfg.n_substeps <= 0) h->cfg.n_substeps = 10;
  auto bail = [&](int code) { CassieVecFree(h); return code; };
  if (hipSetDevice(device) != hipSuccess) return bail(CASSIE_EHIP);
  size_t n = (size_t)n_envs;
  if (hipMalloc(&h->state, n * cassie::ENV_STRIDE * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->d_act, n * 7 * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->d_obs, n * 26 * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->d_rew, n * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->d_done, n) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->d_q, n * 18 * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->d_v, n * 13 * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->ovf, n * OVF_STRIDE * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->pending, n * sizeof(int)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMemset(h->pending, 0, n * sizeof(int)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->pending_leg, n * sizeof(int)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMemset(h->pending_leg, 0, n * sizeof(int)) != hipSuccess) return bail(CASSIE_EHIP);
#ifdef CASSIE_PHASE_TIMING
  if (hipMalloc(&h->phase, 16 * sizeof(unsigned long long)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMemset(h->phase, 0, 16 * sizeof(unsigned long long)) != hipSuccess) return bail(CASSIE_EHIP);
#endif
  if (hipMalloc(&h->stats, cassie::STAT_N * sizeof(unsigned long long)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMemset(h->stats, 0, cassie::STAT_N * sizeof(unsigned long long)) != hipSuccess) return bail(CASSIE_EHIP);
  { const char* e = getenv("CASSIE2D_G16"); if (e && e[0] == '0') h->g16 = false; }
  if (h->cfg.flags & CASSIE_WAVE_PER_ENV) h->g16 = false;
  // Two lanes per environment = 32 environments per wavefront, one wavefront per SIMD: a launch of up to 32 768 environments takes
  // one wavefront's time (0.74 ms per Env.step of ten substeps).  Below LEG_MIN_ENVS the 4-environments-per-wavefront kernel (8x
  // more wavefronts, two per SIMD, 0.45-0.6 ms for a lone pair of wavefronts) is faster.
  h->leg = h->g16 && n_envs >= LEG_MIN_ENVS;
  { const char* e = getenv("CASSIE2D_LEG"); if (e && (e[0] == '0' || e[0] == '1')) h->leg = h->g16 && e[0] == '1'; }
  if (h->cfg.flags & CASSIE_LEG_TIER_OFF) h->leg = false;
  if (h->cfg.flags & CASSIE_LEG_TIER_ON) h->leg = h->g16;
  // the joint-sweep form of that tier (64 environments per wavefront) where the batch fills the chip with it: one wavefront per SIMD is
  // 65 536 environments; below DUO_MIN_ENVS the 32-environment wavefronts of the pair form finish earlier (twice as many SIMDs busy)
  int simds = 1024;
  {
    // both forms run one wavefront per SIMD, so a launch takes whole ROUNDS of the chip's SIMDs: per round the pair form's wavefront
    // (32 environments) lives ~0.62 ms, the joint form's (64 environments) ~1.0 ms (r06 sweep, MI355X: profiles/r06_size_sweep.jsonl).  Three
    // candidates: the pair form for everyone; the joint form for everyone; the joint form for the whole rounds of the batch and the pair form for the
    // remainder (r06).  32 768 envs: one round either way, the pair form's is shorter; 65 536: 1 x 1.0 against 2 x 0.62; 98 304: 1.0 + 0.62 against
    // 3 x 0.62 or 2 x 1.0; 131 072: 2 x 1.0.
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) simds = 4 * prop.multiProcessorCount;
    const double T_PAIR = 0.62, T_JOINT = 1.0;
    const int round_pair = 32 * simds, round_joint = 64 * simds;
    auto rounds = [](int n, int per) { return (n + per - 1) / per; };
    const double cost_pair = T_PAIR * rounds(n_envs, round_pair), cost_joint = T_JOINT * rounds(n_envs, round_joint);
    const int whole = n_envs / round_joint * round_joint, rest = n_envs - whole;
    const double cost_split = T_JOINT * (whole / round_joint) + T_PAIR * rounds(rest, round_pair);
    h->duo_envs = 0;
    if (h->leg && n_envs > DUO_MIN_ENVS) {
      if (whole > 0 && rest > 0 && cost_split < cost_pair && cost_split < cost_joint) h->duo_envs = whole;
      else if (1.05 * cost_joint < cost_pair) h->duo_envs = n_envs;
    }
  }
  { const char* e = getenv("CASSIE2D_DUO"); if (e && (e[0] == '0' || e[0] == '1')) h->duo_envs = (h->leg && e[0] == '1') ? n_envs : 0; }
  if (h->cfg.flags & CASSIE_DUO_TIER_OFF) h->duo_envs = 0;
  if (h->cfg.flags & CASSIE_DUO_TIER_ON) h->duo_envs = h->leg ? n_envs : 0;
  h->duo = h->duo_envs > 0;
  if (h->duo) {
    h->duo_table = L2::duo_table_slots(h->duo_envs, simds);
    { const char* e = getenv("CASSIE2D_DUO_TABLE"); if (e && atoi(e) >= 2) { int t = 2; while (t < atoi(e)) t *= 2; h->duo_table = t; } }   // tests: a small batch through the claim path
    { const char* e = getenv("CASSIE2D_DUO_FLAT_HINT"); h->duo_flat_hint = e && e[0] == '1'; }
    h->duo_ws_bytes = L2::duo_workspace_bytes(h->duo_envs, h->duo_table);
    if (hipMalloc(&h->duo_ws, h->duo_ws_bytes) != hipSuccess) return bail(CASSIE_EHIP);
    if (hipMemset(h->duo_ws, 0, h->duo_ws_bytes) != hipSuccess) return bail(CASSIE_EHIP);
  }
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) return bail(CASSIE_EHIP);
  { const char* e = getenv("CASSIE2D_SIDE_BY_SIDE"); if (e && (e[0] == '0' || e[0] == '1')) h->side_mode = e[0] - '0'; }
  { const char* e = getenv("CASSIE2D_SEGMENTS"); if (e && (e[0] == '0' || e[0] == '1')) h->seg_mode = e[0] - '0'; }
  if (hipHostMalloc((void**)&h->pend_hint, 64 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return bail(CASSIE_EHIP);
  for (int k = 0; k < 64; k++) h->pend_hint[k] = 0;
  if (hipHostGetDevicePointer((void**)&h->pend_hint_dev, h->pend_hint, 0) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->pend_count, 130 * sizeof(unsigned)) != hipSuccess || hipMemset(h->pend_count, 0xFF, 130 * sizeof(unsigned)) != hipSuccess || hipMemset(h->pend_count, 0, 2 * sizeof(unsigned)) != hipSuccess) return bail(CASSIE_EHIP);
  { const char* e = getenv("CASSIE2D_RESET_PACKED"); if (e && e[0] == '0') h->reset_packed = false; }
  if (hipMalloc(&h->need_slow, n) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipHostMalloc((void**)&h->deep_hint, sizeof(int), hipHostMallocMapped) != hipSuccess) return bail(CASSIE_EHIP);
  *h->deep_hint = 0;
  if (hipHostGetDevicePointer((void**)&h->deep_hint_dev, h->deep_hint, 0) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) return bail(CASSIE_EHIP);
  // Cassie2d::Cassie2d: ctor pose, mj_forward, setState (Cassie2d.cpp:56-64)
  L2::init_state(n_envs, h->stream, h->state);
  if (launch_reset(h, nullptr, nullptr, nullptr, nullptr, true) != CASSIE_OK) return bail(CASSIE_EHIP);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(CASSIE_EHIP);
  *out = h;
  return CASSIE_OK;
}

void CassieVecFree(CassieVec* h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipFree(h->state); hipFree(h->traj_qpos); hipFree((void*)h->hf.h); hipFree(h->d_act); hipFree(h->d_obs); hipFree(h->d_rew);
  hipFree(h->d_done); hipFree(h->d_q); hipFree(h->d_v); hipFree(h->d_dbg); hipFree(h->ovf); hipFree(h->ovf_dbg); hipFree(h->pending); hipFree(h->pending_leg); hipFree(h->duo_ws); hipFree(h->qp_stats); hipFree(h->pend_count); hipFree(h->stats); hipFree(h->phase);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  if (h->side) { hipStreamSynchronize(h->side); hipStreamDestroy(h->side); }
  for (int j = 0; j < CassieVec::NSEG; j++) {
    if (h->seg_deep[j]) { hipStreamSynchronize(h->seg_deep[j]); hipStreamDestroy(h->seg_deep[j]); }
    if (h->seg_shal[j]) { hipStreamSynchronize(h->seg_shal[j]); hipStreamDestroy(h->seg_shal[j]); }
    if (h->seg_fork[j]) hipEventDestroy(h->seg_fork[j]);
    if (h->seg_join_a[j]) hipEventDestroy(h->seg_join_a[j]);
    if (h->seg_join_b[j]) hipEventDestroy(h->seg_join_b[j]);
    hipFree(h->seg_pend[j]); hipFree(h->seg_pend2[j]);
  }
  hipFree(h->gone); hipFree(h->need_slow);
  if (h->deep_hint) hipHostFree(h->deep_hint);
  if (h->pend_hint) hipHostFree(h->pend_hint);
  delete h;
}

const char* CassieVecLastError(const CassieVec* h) { return h ? h->err.c_str() : "null handle"; }
int CassieVecNumEnvs(const CassieVec* h) { return h ? h->n : 0; }
int CassieVecActionDim(const CassieVec* h) { return h ? adim_of(h->cfg.control_mode) : 0; }
int CassieVecSetStream(CassieVec* h, void* s) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // work queued on the old stream must not race with launches on the new one
  h->stream = (hipStream_t)s;                  // NULL: back to the default stream
  return CASSIE_OK;
}

int CassieVecGetCounters(CassieVec* h, uint64_t* out4) {
  if (!h || !out4) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  unsigned long long host[cassie::STAT_N];
  HIPCHK(h, hipMemcpyAsync(host, h->stats, sizeof host, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  out4[0] = h->substeps_requested;
  out4[1] = host[cassie::STAT_CLEANUP_SUBSTEPS];
  out4[2] = host[cassie::STAT_K1_SUBSTEPS];
  out4[3] = host[cassie::STAT_NONFINITE];
  return CASSIE_OK;
}

int CassieVecQpIterations(CassieVec* h, double* out4) {
  if (!h || !out4) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = (size_t)h->n;
  out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
  if (!h->qp_stats) {   // first call: start counting
    HIPCHK(h, hipMalloc(&h->qp_stats, 3 * n * sizeof(unsigned)));
    HIPCHK(h, hipMemsetAsync(h->qp_stats, 0, 3 * n * sizeof(unsigned), h->stream));
    return CASSIE_OK;
  }
  std::vector<unsigned> host(3 * n);
  HIPCHK(h, hipMemcpyAsync(host.data(), h->qp_stats, 3 * n * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemsetAsync(h->qp_stats, 0, 3 * n * sizeof(unsigned), h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double sum = 0.0, calls = 0.0, mx = 0.0, env_mean_max = 0.0;
  for (size_t e = 0; e < n; e++) {
    sum += host[e]; calls += host[2 * n + e];
    if (host[n + e] > mx) mx = host[n + e];
    if (host[2 * n + e]) { const double m = (double)host[e] / host[2 * n + e]; if (m > env_mean_max) env_mean_max = m; }
  }
  out4[0] = calls > 0 ? sum / calls : 0.0; out4[1] = mx; out4[2] = calls; out4[3] = env_mean_max;
  return CASSIE_OK;
}

int CassieVecTierInfo(CassieVec* h, uint64_t* out8) {
  if (!h || !out8) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  unsigned long long host[cassie::STAT_N];
  HIPCHK(h, hipMemcpyAsync(host, h->stats, sizeof host, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  out8[0] = !h->g16 ? 0 : !h->leg ? 1 : !h->duo ? 2 : 3;
  out8[1] = (uint64_t)h->duo_table;
  out8[2] = (uint64_t)h->duo_ws_bytes;
  out8[3] = host[cassie::STAT_WS_PROBES];
  out8[4] = h->pend_rate;
  out8[5] = (uint64_t)L2::duo_workspace_slots_per_wave();
  out8[6] = (uint64_t)h->duo_envs;
  out8[7] = 0;
  return CASSIE_OK;
}

#ifdef CASSIE_PHASE_TIMING
extern "C" int CassieVecPhaseCycles(CassieVec* h, unsigned long long* out16) {  // profiling builds only; not part of the public ABI
  if (!h || !h->phase) return CASSIE_EINVAL;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return CASSIE_EHIP;
  if (hipMemcpy(out16, h->phase, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return CASSIE_EHIP;
  hipMemset(h->phase, 0, 16 * sizeof(unsigned long long));
  return CASSIE_OK;
}
#endif

int CassieVecAccumulate(CassieVec* h, const double* reward_dev, const uint8_t* done_dev, double* returns_dev, unsigned long long* episodes_dev) {
  if (!h || (returns_dev && !reward_dev) || (episodes_dev && !done_dev)) return fail(h, CASSIE_EINVAL, "CassieVecAccumulate: an accumulator without its input");
  if (!returns_dev && !episodes_dev) return CASSIE_OK;
  HIPCHK(h, hipSetDevice(h->device));
  L2::accumulate_returns(h->n, h->stream, reward_dev, done_dev, returns_dev, episodes_dev);
  HIPCHK(h, hipGetLastError());
  return CASSIE_OK;
}

int CassieVecResetCounters(CassieVec* h) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemsetAsync(h->stats, 0, cassie::STAT_N * sizeof(unsigned long long), h->stream));
  h->substeps_requested = 0;
  return CASSIE_OK;
}
int CassieVecSynchronize(CassieVec* h) { if (!h) return CASSIE_EINVAL; HIPCHK(h, hipStreamSynchronize(h->stream)); return CASSIE_OK; }
void* CassieVecStatePtr(CassieVec* h) { return h ? h->state : nullptr; }

int CassieVecSetTrajectory(CassieVec* h, const double* time_host, const double* qpos_host, int n) {
  if (!h || !time_host || !qpos_host || n <= 0) return fail(h, CASSIE_EINVAL, "bad trajectory");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->traj_qpos) hipFree(h->traj_qpos);
  HIPCHK(h, hipMalloc(&h->traj_qpos, (size_t)n * 13 * sizeof(double)));
  HIPCHK(h, hipMemcpy(h->traj_qpos, qpos_host, (size_t)n * 13 * sizeof(double), hipMemcpyHostToDevice));
  h->traj_tmax = time_host[n - 1];  // cassie2d_trajectory.py:17
  h->traj_n = n;
  return CASSIE_OK;
}

int CassieVecSetHeightField(CassieVec* h, const double* heights_host, int nrow, int ncol, double size_x, double size_y) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->hf.h) { hipFree((void*)h->hf.h); h->hf = cassie::Terrain{}; }
  if (!heights_host) return CASSIE_OK;  // back to the flat floor
  if (nrow < 2 || ncol < 2 || !(size_x > 0) || !(size_y > 0)) return fail(h, CASSIE_EINVAL, "bad height field");
  double* d = nullptr;
  const size_t bytes = (size_t)nrow * ncol * sizeof(double);
  HIPCHK(h, hipMalloc(&d, bytes));
  HIPCHK(h, hipMemcpy(d, heights_host, bytes, hipMemcpyHostToDevice));
  double hmax = heights_host[0];
  for (size_t i = 1; i < (size_t)nrow * ncol; i++) hmax = heights_host[i] > hmax ? heights_host[i] : hmax;
  h->hf.h = d; h->hf.nrow = nrow; h->hf.ncol = ncol; h->hf.sx = size_x; h->hf.sy = size_y; h->hf.hmax = hmax;
  return CASSIE_OK;
}

int CassieVecReset(CassieVec* h, const uint8_t* mask_dev, double* obs_dev) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  return launch_reset(h, mask_dev, nullptr, nullptr, obs_dev, false);
}

int CassieVecResetTo(CassieVec* h, const uint8_t* mask_dev, const double* qpos_dev, const double* qvel_dev, double* obs_dev) {
  if (!h || !qpos_dev) return fail(h, CASSIE_EINVAL, "qpos_dev is required");
  HIPCHK(h, hipSetDevice(h->device));
  return launch_reset(h, mask_dev, qpos_dev, qvel_dev, obs_dev, false);
}

int CassieVecStep(CassieVec* h, const double* actions_dev, double* obs_dev, double* reward_dev, uint8_t* done_dev, double* terminal_obs_dev) {
  if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev) return fail(h, CASSIE_EINVAL, "null argument");
  if (h->cfg.env_kind == CASSIE_ENV_WALK && !h->traj_qpos) return fail(h, CASSIE_EINVAL, "walk env needs CassieVecSetTrajectory first");
  HIPCHK(h, hipSetDevice(h->device));
  cassie::VecParams p = make_params(h);
  p.actions = actions_dev; p.obs = obs_dev; p.reward = reward_dev; p.done = done_dev; p.terminal_obs = terminal_obs_dev;
  h->substeps_requested += (unsigned long long)h->n * p.n_sub;
  return launch_step(h, h->cfg.control_mode, p);
}

int CassieVecSubstep(CassieVec* h, int control_mode, const double* actions_dev, int n_sub) {
  if (!h || !actions_dev || n_sub <= 0) return fail(h, CASSIE_EINVAL, "bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  cassie::VecParams p = make_params(h);
  p.actions = actions_dev; p.adim = adim_of(control_mode); p.n_sub = n_sub; p.obs = nullptr;
  h->substeps_requested += (unsigned long long)h->n * n_sub;
  return launch_step(h, control_mode, p);
}

int CassieVecStandingStep(CassieVec* h, int control_mode, const double* zpos_dev, const double* zvel_dev, int n_sub) {
  if (!h || !zpos_dev || !zvel_dev || n_sub <= 0) return fail(h, CASSIE_EINVAL, "bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  cassie::VecParams p = make_params(h);
  p.actions = nullptr; p.n_sub = n_sub; p.obs = nullptr;
  h->substeps_requested += (unsigned long long)h->n * n_sub;
  return launch_ctrl_step(h, control_mode, p, zpos_dev, zvel_dev);
}

int CassieVecGetState(CassieVec* h, double* qpos_dev, double* qvel_dev) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  L2::get_state(h->n, h->stream, h->state, qpos_dev, qvel_dev);
  HIPCHK(h, hipGetLastError());
  return CASSIE_OK;
}

int CassieVecGetOpState(CassieVec* h, double* x18_dev) {
  if (!h || !x18_dev) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  cassie::VecParams p = make_params(h);
  L2::opstate(h->n, h->stream, p, x18_dev);
  HIPCHK(h, hipGetLastError());
  return CASSIE_OK;
}

int CassieVecStepHost(CassieVec* h, const double* a, double* obs, double* rew, uint8_t* done) {
  if (!h || !a) return CASSIE_EINVAL;
  size_t n = h->n;
  int ad = adim_of(h->cfg.control_mode);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(h->d_act, a, n * ad * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int rc = CassieVecStep(h, h->d_act, h->d_obs, h->d_rew, h->d_done, nullptr);
  if (rc) return rc;
  if (obs) HIPCHK(h, hipMemcpyAsync(obs, h->d_obs, n * 26 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (rew) HIPCHK(h, hipMemcpyAsync(rew, h->d_rew, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (done) HIPCHK(h, hipMemcpyAsync(done, h->d_done, n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int CassieVecGetStateHost(CassieVec* h, double* q, double* v) {
  if (!h) return CASSIE_EINVAL;
  int rc = CassieVecGetState(h, h->d_q, h->d_v);
  if (rc) return rc;
  size_t n = h->n;
  if (q) HIPCHK(h, hipMemcpyAsync(q, h->d_q, n * 13 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (v) HIPCHK(h, hipMemcpyAsync(v, h->d_v, n * 13 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int CassieVecSetStateHost(CassieVec* h, const double* s) {
  if (!h || !s) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(h->state, s, (size_t)h->n * cassie::ENV_STRIDE * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int CassieVecGetFullStateHost(CassieVec* h, double* s) {
  if (!h || !s) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(s, h->state, (size_t)h->n * cassie::ENV_STRIDE * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int CassieVecDebugWorkspaceHost(CassieVec* h, double* out_host, uint64_t max_doubles, uint64_t* n_doubles) {
  if (!h || !n_doubles) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  *n_doubles = h->duo_ws_bytes / sizeof(double);
  if (!out_host || !h->duo_ws) return CASSIE_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const uint64_t m = *n_doubles < max_doubles ? *n_doubles : max_doubles;
  HIPCHK(h, hipMemcpy(out_host, h->duo_ws, m * sizeof(double), hipMemcpyDeviceToHost));
  return CASSIE_OK;
}

int CassieVecDebugSubstepHost(CassieVec* h, int control_mode, const double* a, double* dbg_host) {
  if (!h || !a || !dbg_host) return CASSIE_EINVAL;
  size_t n = h->n;
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->d_dbg) HIPCHK(h, hipMalloc(&h->d_dbg, n * cassie::DBG_STRIDE * sizeof(double)));
  if (!h->ovf_dbg) HIPCHK(h, hipMalloc(&h->ovf_dbg, n * OVF_STRIDE_DBG * sizeof(double)));
  HIPCHK(h, hipMemsetAsync(h->d_dbg, 0, n * cassie::DBG_STRIDE * sizeof(double), h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_act, a, n * adim_of(control_mode) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  cassie::VecParams p = make_params(h);
  p.actions = h->d_act; p.adim = adim_of(control_mode); p.n_sub = 1; p.obs = nullptr; p.debug = h->d_dbg;
  if (control_mode == CASSIE_CTRL_PD || control_mode == CASSIE_CTRL_TORQUE) { p.ovf = h->ovf_dbg; p.ovf_stride = OVF_STRIDE_DBG; }
  int rc = launch_step(h, control_mode, p);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(dbg_host, h->d_dbg, n * cassie::DBG_STRIDE * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int CassieVecTimeSteps(CassieVec* h, const double* actions_dev, int steps, double* obs_dev, double* reward_dev, uint8_t* done_dev, float* avg_ms) {
  if (!h || steps <= 0 || !avg_ms) return CASSIE_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < steps; i++) {
    int rc = CassieVecStep(h, actions_dev, obs_dev, reward_dev, done_dev, nullptr);
    if (rc) return rc;
  }
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_ms = ms / steps;
  return CASSIE_OK;
}

// ------------------------------------------------------------------------------------------------ legacy ABI
struct Cassie2d {
  CassieVec* vec;
  bool display;
};

static void legacy_die(const char* what, CassieVec* v) {
  fprintf(stderr, "libcassie2d: %s: %s\n", what, v ? CassieVecLastError(v) : "");
  abort();  // the reference exits the process on init failure too (Cassie2d.cpp:49-52)
}

Cassie2d* Cassie2dInit(void) {
  CassieVecConfig cfg{};
  cfg.env_kind = CASSIE_ENV_STAND; cfg.control_mode = CASSIE_CTRL_PD; cfg.n_substeps = 1; cfg.flags = 0; cfg.auto_reset = 0;
  CassieVec* v = nullptr;
  int rc = CassieVecCreate(&v, 1, 0, &cfg);
  if (rc != CASSIE_OK) { fprintf(stderr, "libcassie2d: Cassie2dInit failed (%d): no usable MI355X/HIP device\n", rc); abort(); }
  Cassie2d* c = new Cassie2d();
  c->vec = v; c->display = false;
  return c;
}

void Reset(Cassie2d* c, StateGeneral* s) {
  double q[13], v[13];
  for (int i = 0; i < 3; i++) { q[i] = s->base_pos[i]; v[i] = s->base_vel[i]; }
  for (int i = 0; i < 5; i++) { q[3 + i] = s->left_pos[i]; v[3 + i] = s->left_vel[i]; q[8 + i] = s->right_pos[i]; v[8 + i] = s->right_vel[i]; }
  CassieVec* h = c->vec;
  if (hipMemcpy(h->d_q, q, sizeof q, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(h->d_v, v, sizeof v, hipMemcpyHostToDevice) != hipSuccess)
    legacy_die("Reset copy", h);
  if (launch_reset(h, nullptr, h->d_q, h->d_v, nullptr, false)) legacy_die("Reset", h);
  if (CassieVecSynchronize(h)) legacy_die("Reset sync", h);
}

static void legacy_step(Cassie2d* c, int mode, const double* a, int adim) {
  CassieVec* h = c->vec;
  if (hipMemcpy(h->d_act, a, adim * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) legacy_die("Step copy", h);
  if (CassieVecSubstep(h, mode, h->d_act, 1)) legacy_die("Step", h);
  if (CassieVecSynchronize(h)) legacy_die("Step sync", h);
}

void StepTorque(Cassie2d* c, ControllerTorque* a) { legacy_step(c, CASSIE_CTRL_TORQUE, a->torques, 6); }
void StepPd(Cassie2d* c, ControllerPd* a) { legacy_step(c, CASSIE_CTRL_PD, a->angles, 6); }
void StepOsc(Cassie2d* c, ControllerOsc* a) { legacy_step(c, CASSIE_CTRL_OSC, a->body_xdd, 7); }
void StepJacobian(Cassie2d* c, ControllerForce* a) { legacy_step(c, CASSIE_CTRL_JACOBIAN, a->left_force, 6); }

void GetGeneralState(Cassie2d* c, StateGeneral* s) {
  double q[13], v[13];
  if (CassieVecGetStateHost(c->vec, q, v)) legacy_die("GetGeneralState", c->vec);
  for (int i = 0; i < 3; i++) { s->base_pos[i] = q[i]; s->base_vel[i] = v[i]; }
  for (int i = 0; i < 5; i++) { s->left_pos[i] = q[3 + i]; s->left_vel[i] = v[3 + i]; s->right_pos[i] = q[8 + i]; s->right_vel[i] = v[8 + i]; }
}

void GetOperationalSpaceState(Cassie2d* c, StateOperationalSpace* s) {
  double x[18];
  CassieVec* h = c->vec;
  if (CassieVecGetOpState(h, h->d_q)) legacy_die("GetOperationalSpaceState", h);
  if (hipMemcpy(x, h->d_q, sizeof x, hipMemcpyDeviceToHost) != hipSuccess) legacy_die("GetOperationalSpaceState copy", h);
  // only elements [0],[1] of each vector and body_x[2]/body_xd[2] are written (Cassie2d.cpp:225-235, quirk Q4)
  for (int i = 0; i < 2; i++) {
    s->body_x[i] = x[i]; s->body_xd[i] = x[3 + i];
    s->left_x[i] = x[6 + i]; s->left_xd[i] = x[9 + i];
    s->right_x[i] = x[12 + i]; s->right_xd[i] = x[15 + i];
  }
  s->body_x[2] = x[2]; s->body_xd[2] = x[5];
}

void Display(Cassie2d* c, bool display) { c->display = display; }
void Render(Cassie2d*) {}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// include/cassie3d_vec.h: batched Cassie3d physics
static_assert(CASSIE3D_STATE_STRIDE == cassie3d::ENV3_STRIDE && CASSIE3D_NQ == cassie3d::NQ && CASSIE3D_NV == cassie3d::NV &&
                  CASSIE3D_NU == cassie3d::NU && CASSIE3D_DEBUG_STRIDE == cassie3d::D3_STRIDE && CASSIE3D_OFF_QVEL == cassie3d::E3_V &&
                  CASSIE3D_OFF_WARMSTART == cassie3d::E3_WS && CASSIE3D_OFF_CTRL == cassie3d::E3_CTRL && CASSIE3D_OFF_TIME == cassie3d::E3_TIME &&
                  CASSIE3D_OFF_OVERFLOW == cassie3d::E3_OVF,
              "public Cassie3d layout must match the kernel layout");

struct Cassie3dVec {
  int n = 0, device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;  // kernels run on `stream`; `own_stream` is the one this handle created
  double *state = nullptr, *d_act = nullptr, *d_dbg = nullptr;
  int* pending = nullptr;
  int* pending_leg = nullptr;   // substeps left per env after the lane-per-leg kernel
  bool leg = true;    // first tier = the lane-per-leg kernel (cassie3d_leg.hip), 32 environments per wavefront; CASSIE3D_LEG=0: the r03 tiers only
  bool leg_full = false;   // CASSIE3D_LEG=64: 32 environments per wavefront instead of 16 (A/B)
  bool pair = false;  // CASSIE3D_PAIR=1: first pass with TWO environments per wavefront (cassie3d_pair.hip; parity-green, measured slower: kept as cross-check)
  unsigned long long* stats = nullptr;
  unsigned long long substeps_requested = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // the step in segments (launch3d): per segment the hand-over lists, a side stream and two events
  static constexpr int NSEG = 3;
  bool seg_on = true;                        // CASSIE3D_SEGMENTS=0: the whole step in one launch of the lane-per-leg kernel (A/B, tests)
  int* seg_pend[NSEG] = {};
  int* seg_pend2[NSEG] = {};
  int* gone = nullptr;
  hipStream_t seg_side[NSEG] = {};
  hipEvent_t seg_fork[NSEG] = {}, seg_join[NSEG] = {};
  std::string err;
};

namespace {
// fast kernel (<= 32 rows, 2 waves per SIMD) for everyone, then the general kernel for the environments it left pending
void launch3d(Cassie3dVec* h, cassie3d::Params3 p) {
  p.stats = h->stats;
  if (p.debug) {
    L3::step3d(1, h->n, h->stream, p);
    return;
  }
  // tiers: one lane per leg (rows in per-lane LDS slots: 3 connect rows + 25 slots per limit + 81 per contact) -> one wavefront
  // per environment with at most 32 rows -> the general kernel (64 rows, capped); each leaves an environment it cannot hold
  // untouched from that substep on and says how many substeps are left
  p.gone = nullptr; p.seg_first = 1; p.seg_later = 0;
  if (h->leg && h->seg_on && h->gone && p.n_sub >= 2 * Cassie3dVec::NSEG) {
    // ---- the step in segments (r04): the lane-per-leg kernel runs the substeps in three launches; after each, the environments that
    // left its row capacity in that segment are finished -- to the END of the step -- by the wavefront-per-environment kernels on the
    // segment's own stream, while the lane-per-leg kernel goes on with the next segment for everyone else.  A hand-over costs ~0.25 ms
    // per substep in the lower tiers against ~0.5 ms per substep of the first tier's own chain, so all but the last segment's
    // hand-overs are off the step's critical path (16 384 envs: 0.26 ms of tail per step -> 0.0x).
    int later = p.n_sub;
    bool forked[Cassie3dVec::NSEG] = {};
    for (int j = 0; j < Cassie3dVec::NSEG; j++) {
      const int len = (later + (Cassie3dVec::NSEG - j) - 1) / (Cassie3dVec::NSEG - j);
      later -= len;
      cassie3d::Params3 ps = p;
      ps.n_sub = len; ps.pending_in = nullptr; ps.pending_out = h->seg_pend[j];
      ps.gone = h->gone; ps.seg_first = j == 0; ps.seg_later = later;
      L3::step3d(h->leg_full ? 4 : 3, h->n, h->stream, ps);
      // the segment's hand-overs on its own stream; if the fork cannot be set up (or one failed before), on the caller's stream
      // behind the segment: the same kernels on the same data, only without the overlap
      const bool fork = h->seg_on && hipEventRecord(h->seg_fork[j], h->stream) == hipSuccess && hipStreamWaitEvent(h->seg_side[j], h->seg_fork[j], 0) == hipSuccess;
      if (!fork) h->seg_on = false;   // (this step is finished in segments, in order; the next ones in one launch)
      hipStream_t ss = fork ? h->seg_side[j] : h->stream;
      cassie3d::Params3 pa = p, pb = p;
      pa.pending_in = h->seg_pend[j]; pa.pending_out = h->seg_pend2[j];
      L3::step3d(0, h->n, ss, pa);
      pb.pending_in = h->seg_pend2[j]; pb.pending_out = nullptr;
      L3::step3d(1, h->n, ss, pb);
      forked[j] = fork;
    }
    for (int j = 0; j < Cassie3dVec::NSEG; j++) {
      if (!forked[j]) continue;
      if (hipEventRecord(h->seg_join[j], h->seg_side[j]) != hipSuccess || hipStreamWaitEvent(h->stream, h->seg_join[j], 0) != hipSuccess) {
        hipStreamSynchronize(h->seg_side[j]);   // join the hard way
        h->seg_on = false;
      }
    }
    return;
  }
  const int* in = nullptr;
  if (h->leg) {
    p.pending_in = nullptr; p.pending_out = h->pending_leg;
    L3::step3d(h->leg_full ? 4 : 3, h->n, h->stream, p);
    in = h->pending_leg;
  }
  p.pending_in = in; p.pending_out = h->pending;
  L3::step3d(h->pair ? 2 : 0, h->n, h->stream, p);
  p.pending_in = h->pending; p.pending_out = nullptr;
  L3::step3d(1, h->n, h->stream, p);
}
int fail3(Cassie3dVec* h, int code, const char* msg) { if (h) h->err = msg; return code; }
#define HIPCHK3(h, call)                                                                      \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) { if (h) (h)->err = std::string(#call " failed: ") + hipGetErrorString(e_); return CASSIE_EHIP; } \
  } while (0)
}  // namespace

extern "C" {

int Cassie3dVecCreate(Cassie3dVec** out, int n_envs, int device) {
  if (!out || n_envs <= 0) return CASSIE_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    fprintf(stderr, "libcassie2d: no HIP device available; this library has no CPU path\n");
    return CASSIE_ENODEVICE;
  }
  if (device < 0 || device >= ndev) return CASSIE_EINVAL;
  Cassie3dVec* h = new Cassie3dVec();
  h->n = n_envs; h->device = device;
  auto bail = [&](int code) { Cassie3dVecFree(h); return code; };
  if (hipSetDevice(device) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipStreamCreate(&h->stream) != hipSuccess) return bail(CASSIE_EHIP);
  h->own_stream = h->stream;
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->state, (size_t)n_envs * cassie3d::ENV3_STRIDE * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->d_act, (size_t)n_envs * cassie3d::NU * sizeof(double)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->pending, (size_t)n_envs * sizeof(int)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMalloc(&h->pending_leg, (size_t)n_envs * sizeof(int)) != hipSuccess) return bail(CASSIE_EHIP);
  { const char* e = getenv("CASSIE3D_LEG"); if (e && e[0] == '0') h->leg = false; if (e && e[0] == '6') h->leg_full = true; }
  if (hipMalloc(&h->stats, cassie3d::S3_N * sizeof(unsigned long long)) != hipSuccess) return bail(CASSIE_EHIP);
  if (hipMemset(h->stats, 0, cassie3d::S3_N * sizeof(unsigned long long)) != hipSuccess) return bail(CASSIE_EHIP);
  { const char* e = getenv("CASSIE3D_PAIR"); if (e && e[0] == '1') { h->pair = true; h->leg = false; } }   // the opt-in cross-check kernel is a FIRST tier (it takes no pending list)
  { const char* e = getenv("CASSIE3D_SEGMENTS"); if (e && e[0] == '0') h->seg_on = false; }
  if (h->leg && h->seg_on) {   // lists, streams and events of the segmented step (not available: one launch, same results)
    bool ok = hipMalloc(&h->gone, (size_t)n_envs * sizeof(int)) == hipSuccess;
    for (int j = 0; j < Cassie3dVec::NSEG && ok; j++)
      ok = hipMalloc(&h->seg_pend[j], (size_t)n_envs * sizeof(int)) == hipSuccess && hipMalloc(&h->seg_pend2[j], (size_t)n_envs * sizeof(int)) == hipSuccess &&
           hipStreamCreateWithFlags(&h->seg_side[j], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&h->seg_fork[j], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&h->seg_join[j], hipEventDisableTiming) == hipSuccess;
    if (!ok) { h->seg_on = false; (void)hipGetLastError(); }
  }
  if (Cassie3dVecReset(h, nullptr, nullptr) != CASSIE_OK || hipStreamSynchronize(h->stream) != hipSuccess) return bail(CASSIE_EHIP);
  *out = h;
  return CASSIE_OK;
}

void Cassie3dVecFree(Cassie3dVec* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (int j = 0; j < Cassie3dVec::NSEG; j++) {
    if (h->seg_side[j]) { hipStreamSynchronize(h->seg_side[j]); hipStreamDestroy(h->seg_side[j]); }
    if (h->seg_fork[j]) hipEventDestroy(h->seg_fork[j]);
    if (h->seg_join[j]) hipEventDestroy(h->seg_join[j]);
    hipFree(h->seg_pend[j]); hipFree(h->seg_pend2[j]);
  }
  hipFree(h->gone);
  hipFree(h->state); hipFree(h->d_act); hipFree(h->d_dbg); hipFree(h->pending); hipFree(h->pending_leg); hipFree(h->stats);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->own_stream) hipStreamDestroy(h->own_stream);
  delete h;
}

const char* Cassie3dVecLastError(const Cassie3dVec* h) { return h ? h->err.c_str() : "null handle"; }

int Cassie3dVecSetStream(Cassie3dVec* h, void* s) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK3(h, hipStreamSynchronize(h->stream));
  h->stream = s ? (hipStream_t)s : h->own_stream;  // NULL: back to the handle's own stream
  return CASSIE_OK;
}

int Cassie3dVecSynchronize(Cassie3dVec* h) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK3(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int Cassie3dVecReset(Cassie3dVec* h, const double* qpos_dev, const double* qvel_dev) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK3(h, hipSetDevice(h->device));
  L3::init3d(h->n, h->stream, h->state, qpos_dev, qvel_dev);
  cassie3d::Params3 p{};
  p.state = h->state; p.actions = nullptr; p.debug = nullptr; p.n_envs = h->n; p.n_sub = 1; p.integrate = 0;
  launch3d(h, p);  // mj_forward
  HIPCHK3(h, hipGetLastError());
  return CASSIE_OK;
}

int Cassie3dVecStep(Cassie3dVec* h, const double* torques_dev, int n_sub) {
  if (!h || !torques_dev || n_sub <= 0) return fail3(h, CASSIE_EINVAL, "bad argument");
  HIPCHK3(h, hipSetDevice(h->device));
  cassie3d::Params3 p{};
  p.state = h->state; p.actions = torques_dev; p.debug = nullptr; p.n_envs = h->n; p.n_sub = n_sub; p.integrate = 1;
  h->substeps_requested += (unsigned long long)h->n * n_sub;
  launch3d(h, p);
  HIPCHK3(h, hipGetLastError());
  return CASSIE_OK;
}

double* Cassie3dVecStatePtr(Cassie3dVec* h) { return h ? h->state : nullptr; }

int Cassie3dVecGetCounters(Cassie3dVec* h, uint64_t* out4) {
  if (!h || !out4) return CASSIE_EINVAL;
  HIPCHK3(h, hipSetDevice(h->device));
  unsigned long long host[cassie3d::S3_N];
  HIPCHK3(h, hipMemcpyAsync(host, h->stats, sizeof host, hipMemcpyDeviceToHost, h->stream));
  HIPCHK3(h, hipStreamSynchronize(h->stream));
  out4[0] = h->substeps_requested;
  out4[1] = host[cassie3d::S3_GENERAL_SUBSTEPS];
  out4[2] = host[cassie3d::S3_CAPPED_SUBSTEPS];
  out4[3] = host[cassie3d::S3_LEG_HANDOVER_SUBSTEPS];
  return CASSIE_OK;
}

int Cassie3dVecResetCounters(Cassie3dVec* h) {
  if (!h) return CASSIE_EINVAL;
  HIPCHK3(h, hipSetDevice(h->device));
  HIPCHK3(h, hipMemsetAsync(h->stats, 0, cassie3d::S3_N * sizeof(unsigned long long), h->stream));
  h->substeps_requested = 0;
  return CASSIE_OK;
}

int Cassie3dVecStepHost(Cassie3dVec* h, const double* torques, int n_sub) {
  if (!h || !torques) return CASSIE_EINVAL;
  HIPCHK3(h, hipSetDevice(h->device));
  HIPCHK3(h, hipMemcpyAsync(h->d_act, torques, (size_t)h->n * cassie3d::NU * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int rc = Cassie3dVecStep(h, h->d_act, n_sub);
  if (rc) return rc;
  HIPCHK3(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int Cassie3dVecGetStateHost(Cassie3dVec* h, double* state) {
  if (!h || !state) return CASSIE_EINVAL;
  HIPCHK3(h, hipSetDevice(h->device));
  HIPCHK3(h, hipMemcpyAsync(state, h->state, (size_t)h->n * cassie3d::ENV3_STRIDE * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK3(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int Cassie3dVecSetStateHost(Cassie3dVec* h, const double* state) {
  if (!h || !state) return CASSIE_EINVAL;
  HIPCHK3(h, hipSetDevice(h->device));
  HIPCHK3(h, hipMemcpyAsync(h->state, state, (size_t)h->n * cassie3d::ENV3_STRIDE * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK3(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int Cassie3dVecDebugForwardHost(Cassie3dVec* h, const double* torques, double* dbg) {
  if (!h || !torques || !dbg) return CASSIE_EINVAL;
  HIPCHK3(h, hipSetDevice(h->device));
  const size_t nb = (size_t)h->n * cassie3d::D3_STRIDE * sizeof(double);
  if (!h->d_dbg) HIPCHK3(h, hipMalloc(&h->d_dbg, nb));
  HIPCHK3(h, hipMemsetAsync(h->d_dbg, 0, nb, h->stream));
  HIPCHK3(h, hipMemcpyAsync(h->d_act, torques, (size_t)h->n * cassie3d::NU * sizeof(double), hipMemcpyHostToDevice, h->stream));
  cassie3d::Params3 p{};
  p.state = h->state; p.actions = h->d_act; p.debug = h->d_dbg; p.n_envs = h->n; p.n_sub = 1; p.integrate = 0;
  launch3d(h, p);
  HIPCHK3(h, hipGetLastError());
  HIPCHK3(h, hipMemcpyAsync(dbg, h->d_dbg, nb, hipMemcpyDeviceToHost, h->stream));
  HIPCHK3(h, hipStreamSynchronize(h->stream));
  return CASSIE_OK;
}

int Cassie3dVecTimeSteps(Cassie3dVec* h, const double* torques_dev, int n_sub, int steps, float* avg_ms) {
  if (!h || steps <= 0 || !avg_ms) return CASSIE_EINVAL;
  HIPCHK3(h, hipSetDevice(h->device));
  HIPCHK3(h, hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < steps; i++) {
    int rc = Cassie3dVecStep(h, torques_dev, n_sub);
    if (rc) return rc;
  }
  HIPCHK3(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK3(h, hipEventSynchronize(h->ev1));
  float ms = 0;
  HIPCHK3(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_ms = ms / steps;
  return CASSIE_OK;
}

}  // extern "C"
