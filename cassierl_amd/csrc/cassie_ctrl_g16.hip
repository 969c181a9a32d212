// cassie_ctrl_g16.hip -- the in-loop controllers with 4 environments per wavefront (one per 16-lane DPP row), as a kernel of
// their own.
//
//   Cassie2d::StepOsc / StepJacobian      src/Cassie2d/Cassie2d.cpp:119-209   (setState, controller, then mj_step)
//   standing_controller_osc / _jacobian   rllab/envs/cassie2d.py:263-331      (SCRIPTED: targets from the op-space state)
//
// One StepOsc / StepJacobian = this kernel (DynamicModel::setState + DynamicState + controller -> motor commands written into
// the state record) followed by the physics kernel of cassie_kernels_g16.hip in MODE 2 (one mj_step with the commands of the
// record; with the observation / reward / reset section on the last substep of an Env.step).
//
// Why two kernels (r02, tools/phase_profile.py): fused into one kernel the controller's ~500 registers and 8.6 KB of LDS per
// environment held the whole kernel at ONE wavefront per SIMD, and 62-71 % of its time was the physics substep running
// latency-bound at half the issue rate it reaches in its own kernel (2 wavefronts per SIMD).  The price of the split is one
// round trip of the 704-byte state record through HBM/L2 per substep (~90 MB per substep at 65 536 envs: ~15 us).
// The controllers are row-generic (cassie_ctrl.hip); this file adds the LDS layout and the staging around them.
#ifndef CASSIE_CTRL_G16_HIP_
#define CASSIE_CTRL_G16_HIP_

#include <cstddef>
namespace cassie {
namespace g16 {

// LDS of one environment for the controller kernel: 414 doubles = 3.3 KB, 13.2 KB per wavefront, so that TWELVE wavefronts
// (three per SIMD) share the 160 KB of a CU -- the kernel is latency-bound (r03 PMC: 10.7 cycles per VALU instruction at one
// wavefront per SIMD), every further wavefront hides part of that.  r02 had 1035 doubles (33 KB per wavefront, one per SIMD),
// r03..r05 582 (two per SIMD).
// The layout is also the controller's scratch (the `cs` argument of cassie_ctrl.hip is this same object).  Three things make
// it small:
//   * buffers with disjoint lifetimes share storage (the 300-double overlay below; the stale-kinematics inputs kq / kv are only
//     read by scripted_targets, before ctrl_dyn writes acc / bias);
//   * controller rows are kept compact (3 base + 5 own-leg columns instead of 13);
//   * Hinv is symmetric and stored as a packed upper triangle (91 instead of 169 doubles).
struct EnvLdsC {
  double q[16], v[16];
  union {
    struct { double kq[16], kv[16]; };      // kinematics of the LAST setState (scripted targets only)
    struct { double acc[16], bias[16]; };   // JdotQdot of each controller row; NonlinearEffects + damping*qvel
  };
  double y[16], act[8], u[8], s18[18];
  // 300 doubles shared by lifetime (r06: 414 doubles = 3.3 KB per environment, 13.2 KB per wavefront -- TWELVE wavefronts, three per SIMD, share
  // the 160 KB of a CU; r03..r05: 582 doubles, two per SIMD).  Inside one wavefront LDS accesses are in program order, so a block may be overwritten as
  // soon as the code that reads its previous tenant lies behind:
  //   [  0,  84)  lc ls lw lox loz lax laz     read until the controller rows exist          then  JH, S4 (born behind the rows)
  //   [ 84, 252)  lcx lcz lfx lfz s1* lv* site  dead behind mass_rows                         then  Hinv [84, 180) (born behind the Gauss-Jordan)
  //   [180, 300)                                                                               then  the controller rows Jc (born behind the last FK read)
  //   [ 84, 264)  T, t0 / U: stored when every read of Hinv, Jc for the T (U) columns lies behind (ctrl_osc computes a whole column in registers first)
  union {
    struct {                                // forward kinematics and mass-matrix exchange
      double lc[12], ls[12], lw[12], lox[12], loz[12], lax[12], laz[12];
      double lcx[12], lcz[12], lfx[12], lfz[12], s1x[16], s1z[16], s2[16], lvx[12], lvz[12], site[2][6][4];
    };
    struct {                                // constraint projector, Hinv, controller rows
      double JH[4][NV], S4[16], pad0_[16];
      double minv[NV * (NV + 1) / 2 + 5];   // Hinv, packed upper triangle
      double Jc[NCR][8];                    // compact controller rows
    };
    struct {                                // the controller's own matrices
      double pad1_[84];
      union {
        struct { double T[NZ][12]; double t0[12]; };
        struct { double U[6][NV]; };
      };
    };
  };
  __host__ __device__ static constexpr int hidx(int r, int c) {
    const int lo = r <= c ? r : c, hi = r <= c ? c : r;
    return lo * NV - lo * (lo - 1) / 2 + (hi - lo);
  }
};
static_assert(sizeof(EnvLdsC) == 414 * sizeof(double), "layout of the controller's LDS block");
static_assert(offsetof(EnvLdsC, minv) - offsetof(EnvLdsC, lc) == 84 * sizeof(double) && offsetof(EnvLdsC, lcx) == offsetof(EnvLdsC, minv) &&
              offsetof(EnvLdsC, Jc) - offsetof(EnvLdsC, lc) == 180 * sizeof(double) && offsetof(EnvLdsC, T) == offsetof(EnvLdsC, minv), "overlay offsets");
static_assert(sizeof(EnvLdsC) * 4 * 12 <= 160 * 1024, "twelve controller wavefronts must fit the LDS of a CU");

// CTRL: 2 = OSC, 3 = Jacobian.  SCRIPTED: targets from standing_controller_* (zpos/zvel per env) instead of actions.
#ifndef CTRL_WAVES
#define CTRL_WAVES 3   // wavefronts per SIMD the kernel is sized for (r06; registers <= 168, LDS 13.2 KB per wavefront)
#endif
template <int CTRL, bool SCRIPTED>
__global__ void __launch_bounds__(64, CTRL_WAVES) env_ctrl_g16_kernel(VecParams p, const double* zpos, const double* zvel) {
  __shared__ EnvLdsC sm4[4];
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  const int env = blockIdx.x * 4 + g;
  const bool valid = env < p.n_envs;
  EnvLdsC& sm = sm4[g];
  EnvLdsC& cs = sm;  // the controller's scratch lives in the same per-environment block
  const size_t e = valid ? (size_t)env : 0;
  double* st = p.state + e * ENV_STRIDE;
  LaneConst c;
  load_lane_const(c, l);
  c.grp = 0; c.dvalid = l < NV;
  if (l < NV) {
    sm.q[l] = st[ES_Q + l]; sm.v[l] = st[ES_V + l];
    sm.kq[l] = st[ES_KQ + l]; sm.kv[l] = st[ES_KV + l];
  } else { sm.q[l] = 0.0; sm.v[l] = 0.0; sm.kq[l] = 0.0; sm.kv[l] = 0.0; }
  constexpr int ADIM = CTRL == 2 ? 7 : 6;
  const bool fix_kin = (p.flags & FLAG_FIX_STALE_KIN) != 0, noshort = (p.flags & FLAG_NO_PINV_SHORTCUT) != 0;
  unsigned wset = (unsigned)st[ES_QPWSET];  // OSC QP working set of the previous call (uniform inside a row)
  PhaseClock pc;
  pc.start();
  lds_sync();
  if (SCRIPTED) scripted_targets<CTRL>(sm, cs, c, l, valid, fix_kin, zpos[e], zvel[e]);  // reads the kinematics of the LAST setState
  else if (l < ADIM) cs.act[l] = p.actions[e * ADIM + l];
  if (valid && l < NV) { st[ES_KQ + l] = sm.q[l]; st[ES_KV + l] = sm.v[l]; }  // DynamicModel::setState
  lds_sync();
  PHASE_MARK(pc, 0);
  int qpit = 0;
  if (CTRL == 2) ctrl_osc(sm, cs, c, l, valid, g, wset, noshort, &pc, &qpit);
  else ctrl_jacobian(sm, cs, c, l, valid, g, nullptr, noshort);
  if (valid) {
    if (l < NU) st[ES_CTRL + l] = cs.u[l];  // mj_data->ctrl (pre-clamp), consumed by the physics kernel
    if (CTRL == 2 && l == 0) st[ES_QPWSET] = (double)wset;
    if (CTRL == 2 && l == 0 && p.qp_stats) qp_stats_add(p.qp_stats, p.n_envs, env, qpit);
  }
  pc.flush(p.phase, lane);
}

}  // namespace g16
}  // namespace cassie
#endif
