// cassie_ctrl_g16.hip -- controller-in-the-loop Env.step with 4 environments per wavefront (one per 16-lane DPP row).
//
//   Cassie2d::StepOsc / StepJacobian      src/Cassie2d/Cassie2d.cpp:119-209   (controller, then mj_step)
//   standing_controller_osc / _jacobian   rllab/envs/cassie2d.py:263-331      (SCRIPTED: targets from the op-space state)
//   Env.step of the standing task         rllab/envs/cassie_stand2d.py:86-137 (observation, reward, termination)
//
// The controllers of cassie_ctrl.hip are row-generic, the physics is g16::substep; this file only adds the LDS layout in
// which both fit and the kernel that sequences them.  An environment whose physics needs more than 16 constraint rows is
// frozen at that substep (state, kq/kv and time exactly as before the substep) and finished by the wave-per-environment
// kernel env_ctrl_step_kernel through `pending[env]`, as in cassie_kernels_g16.hip.
#ifndef CASSIE_CTRL_G16_HIP_
#define CASSIE_CTRL_G16_HIP_

namespace cassie {
namespace g16 {

// LDS of one environment (8.6 KB; 4 per wavefront).  The controller needs M^-1 together with the FK by-products (link
// accelerations for JdotQdot), so those are not overlaid here; the controller matrices are dead once u is in registers and
// share storage with the constraint rows of the physics substep.
struct EnvLdsC {
  double q[16], v[16], ws[16], ctrl[8];
  double lc[12], ls[12], lw[12], lox[12], loz[12], lcx[12], lcz[12], lfx[12], lfz[12];
  double qs[16];
  double s1x[16], s1z[16], s2[16];
  double minv[NV * NV + 7];
  double lvx[12], lvz[12], lax[12], laz[12], site[2][6][4], s18[18], kq[16], kv[16];
  union {
    struct { double rowJ[MAXR][8]; int rowleg[MAXR]; };
    CtrlSmem cs;
  };
};

// CTRL: 2 = OSC, 3 = Jacobian.  SCRIPTED: targets from standing_controller_* (zpos/zvel per env) instead of actions.
template <int CTRL, bool SCRIPTED>
__global__ void __launch_bounds__(64, 1) env_ctrl_step_g16_kernel(VecParams p, const double* zpos, const double* zvel, int* pending) {
  __shared__ EnvLdsC sm4[4];
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  const int env = blockIdx.x * 4 + g;
  const bool valid = env < p.n_envs;
  EnvLdsC& sm = sm4[g];
  CtrlSmem& cs = sm.cs;
  const size_t e = valid ? (size_t)env : 0;
  double* st = p.state + e * ENV_STRIDE;
  LaneConst c;
  load_lane_const(c, l);
  c.grp = 0; c.dvalid = l < NV;
  double qstate_l = 0.0;
  if (l < NV) {
    sm.q[l] = st[ES_Q + l]; sm.v[l] = st[ES_V + l]; sm.ws[l] = st[ES_WS + l];
    sm.kq[l] = st[ES_KQ + l]; sm.kv[l] = st[ES_KV + l];
    qstate_l = st[ES_QSTATE + l];
  } else { sm.kq[l] = 0.0; sm.kv[l] = 0.0; }
  if (l < NU) sm.ctrl[l] = st[ES_CTRL + l];
  double time = st[ES_TIME];
  constexpr int ADIM = CTRL == 2 ? 7 : 6;
  double act_l = 0.0;
  if (!SCRIPTED && l < ADIM) act_l = p.actions[e * ADIM + l];
  const double zp = SCRIPTED ? zpos[e] : 0.0, zv = SCRIPTED ? zvel[e] : 0.0;
  const bool fix_kin = (p.flags & FLAG_FIX_STALE_KIN) != 0, noshort = (p.flags & FLAG_NO_PINV_SHORTCUT) != 0;
  lds_sync();
  bool live = valid;
  int pend = 0, niter_sum = 0;
  double ctrl = 0.0;
  G16Out so; so.niter = 0; so.overflow = false;
  unsigned wset = (unsigned)st[ES_QPWSET];  // OSC QP working set of the previous call (uniform inside a row)
  for (int sub = 0; sub < p.n_sub; sub++) {
    const double kq_old = sm.kq[l], kv_old = sm.kv[l];
    if (SCRIPTED) scripted_targets<CTRL>(sm, cs, c, l, live, fix_kin, zp, zv);
    else if (l < ADIM) cs.act[l] = act_l;  // the physics substep overlays cs: re-stage the action every substep
    if (live && l < NV) { sm.kq[l] = sm.q[l]; sm.kv[l] = sm.v[l]; }  // DynamicModel::setState
    lds_sync();
    if (CTRL == 2) ctrl_osc(sm, cs, c, l, live, g, wset, noshort);
    else ctrl_jacobian(sm, cs, c, l, live, g, nullptr, noshort);
    const double cnew = c.act >= 0 ? cs.u[c.act] : 0.0;
    lds_sync();
    substep(sm, c, l, g, cnew, live, true, so);
    if (live && so.overflow) {
      // not done here: undo setState so that the clean-up pass sees the environment exactly as before this substep
      live = false; pend = p.n_sub - sub;
      sm.kq[l] = kq_old; sm.kv[l] = kv_old;
    }
    if (live) { ctrl = cnew; niter_sum += so.niter; time += 0.0005; }
    lds_sync();
    if (__ballot(live) == 0) break;
  }
  if (live && c.dvalid && c.act >= 0) sm.ctrl[c.act] = ctrl;
  lds_sync();
  if (p.obs) {
    auto opstate_regs = [&](double& oa, double& ob) {
      opstate18(sm, c, l, fix_kin, sm.s18);
      oa = sm.s18[l + 1 < 18 ? l + 1 : 17];  // obs[l] = s18[l+1]
      if (l == 5 || l == 11) oa -= sm.s18[0];
      ob = l == 0 ? sm.s18[17] : 0.0;        // obs[16 + l]
      lds_sync();
    };
    double obs_a, obs_b;
    opstate_regs(obs_a, obs_b);
    const double bodyx = sm.s18[0];  // s18 is not overlaid by anything the outputs touch
    double reward = 0.0;
    int done = 0;
    {
      const bool fixq = (p.flags & FLAG_FIX_STALE_QSTATE) != 0;
      const double qv = fixq ? sm.q[l < NV ? l : 0] : qstate_l;
      env_outputs_row(p, l, SCRIPTED ? nullptr : p.actions + e * ADIM, ADIM, time, bodyx, qv, obs_a, obs_b, reward, done);
    }
    const bool badl = l < NV && !(in_range(sm.q[l < NV ? l : 0]) && in_range(sm.v[l < NV ? l : 0]));  // failure guard
    const bool bad = live && ((((unsigned)(__ballot(badl) >> (16 * g))) & 0xFFFFu) != 0 || !in_range(reward));
    if (bad) {
      obs_a = 0.0; obs_b = 0.0; reward = 0.0; done = 1;
      if (l == 0 && p.stats) atomicAdd(p.stats + STAT_NONFINITE, 1ull);
      if (p.auto_reset) {
        sm.ws[l] = 0.0; sm.kq[l] = l < NV ? cp_env_qinit[l] : 0.0; sm.kv[l] = 0.0;
        if (l < NU) sm.ctrl[l] = 0.0;
      }
    }
    if (live && p.terminal_obs) { p.terminal_obs[e * 26 + l] = obs_a; if (l < 10) p.terminal_obs[e * 26 + 16 + l] = obs_b; }
    const bool do_reset = live && done && p.auto_reset;
    if (__ballot(do_reset) != 0) {
      if (do_reset && l < NV) { sm.q[l] = cp_env_qinit[l]; sm.v[l] = 0.0; qstate_l = cp_env_qinit[l]; }
      if (do_reset) { time = 0.0; wset = 0u; }
      lds_sync();
      G16Out ro; ro.niter = 0; ro.overflow = false;
      substep(sm, c, l, g, c.act >= 0 ? sm.ctrl[c.act] : 0.0, do_reset, false, ro);  // reset pose: 12 rows, cannot overflow
      double ra, rb;
      opstate_regs(ra, rb);
      if (do_reset) { obs_a = ra; obs_b = rb; }
    }
    if (live) {
      p.obs[e * 26 + l] = obs_a;
      if (l < 10) p.obs[e * 26 + 16 + l] = obs_b;
      if (l == 0) { p.reward[env] = reward; p.done[env] = (uint8_t)done; }
    }
  }
  if (valid) {
    if (l < NV) {
      st[ES_Q + l] = sm.q[l]; st[ES_V + l] = sm.v[l]; st[ES_WS + l] = sm.ws[l];
      st[ES_KQ + l] = sm.kq[l]; st[ES_KV + l] = sm.kv[l]; st[ES_QSTATE + l] = qstate_l;
    }
    if (l < NU) st[ES_CTRL + l] = sm.ctrl[l];
    if (l == 0) {
      st[ES_TIME] = time; st[ES_NITER] = (double)niter_sum; if (CTRL == 2) st[ES_QPWSET] = (double)wset; pending[env] = pend;
      if (pend > 0 && p.stats) atomicAdd(p.stats + STAT_CLEANUP_SUBSTEPS, (unsigned long long)pend);
    }
  }
}

}  // namespace g16
}  // namespace cassie
#endif
