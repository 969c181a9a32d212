// cassie_kernels_leg.hip -- gfx950 backend and kernel of the two-lanes-per-environment Env.step (cassie_leg_core.h): one lane
// per leg, 32 environments per wavefront, one wavefront per workgroup.
//
//   lane 2e     left leg of environment e (+ the pelvis collision sphere, + the base-dof fields of the record on write-back)
//   lane 2e + 1 right leg
// The two lanes of an environment exchange values with DPP quad_perm [1,0,3,2] (v_mov_b32_dpp x2 per double); nothing else
// crosses lanes, there is no LDS traffic between lanes and no barrier.  LDS is used as per-lane indexed storage for the
// compaction of the active set (a lane writes the descriptor of its n-th active contact / limit at slot n -- a run-time index a
// register array cannot take -- and reads the slots back at compile-time indices) and for the cold part of the lane state
// (setState snapshot, qstate, ctrl, clock, action: long-lived, touched once per substep): 29 KB per wavefront, bank = lane.
// Registers: sized for ONE wavefront per SIMD (512 VGPR + AGPR): the sweep keeps the leg's 8 x 8 symmetric block of A, the
// base-coupling vectors, residuals and forces (~100 doubles) in VGPRs; the allocator parks the rest in AGPRs, not scratch.
// Reference call sites: as cassie_kernels.hip (Cassie2d::Step/StepPd + mj_step + Cassie2dEnv.step).
#ifndef CASSIE_KERNELS_LEG_HIP_
#define CASSIE_KERNELS_LEG_HIP_
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LEG_FN __device__ __forceinline__
#include "cassie_leg_core.h"

namespace cassie {
namespace leg {

struct DevB {
  static constexpr bool SPLIT_TAIL = true;   // sub_setup: a wavefront on its feet does not build its two empty row slots (cassie_leg_core.h)
  typedef double D;
  typedef int I;
  typedef bool M;
  typedef double* P;
  typedef uint8_t* P8;
  struct OwnerScope { LEG_FN OwnerScope(bool) {} };
  typedef const double* K;   // the lane's row of cp_legk
  static LEG_FN K kbase(int leg) { return &cp_legk[0][0] + leg * LK_N; }
  static LEG_FN double kld(K k, int idx) { return k[idx]; }
  // per-lane slots: [slot][field][lane]
  struct Lds {
    double pr[3][4][64];
    double lm[4][3][64];
    double cold[49][64];   // Core::C_* slots
    int pdepth[3][64];
    int lmj[4][64];
#ifdef CASSIE_PHASE_TIMING
    unsigned long long t_last, acc[16];   // profiling builds: shader cycles per code phase of this wavefront (tools/phase_profile.py leg)
    LEG_FN void mark(int k) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long n = __builtin_readcyclecounter();
      if (threadIdx.x == 0) { acc[k] += n - t_last; t_last = n; }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    LEG_FN void mark(int) {}
#endif
    LEG_FN double cld(int i) const { return cold[i][threadIdx.x]; }
    LEG_FN void cst(int i, double v, bool m) { if (m) cold[i][threadIdx.x] = v; }
    LEG_FN void st_pair(int slot, double px, double pz, double dist, double invw, int depth, bool m) {
      if (m) {
        const int l = threadIdx.x;
        pr[slot][0][l] = px; pr[slot][1][l] = pz; pr[slot][2][l] = dist; pr[slot][3][l] = invw; pdepth[slot][l] = depth;
      }
    }
    LEG_FN void ld_pair(int s, double& px, double& pz, double& dist, double& invw, int& depth) const {
      const int l = threadIdx.x;
      px = pr[s][0][l]; pz = pr[s][1][l]; dist = pr[s][2][l]; invw = pr[s][3][l]; depth = pdepth[s][l];
    }
    LEG_FN void st_lim(int slot, double pos, double sgn, double invw, int j, bool m) {
      if (m) {
        const int l = threadIdx.x;
        lm[slot][0][l] = pos; lm[slot][1][l] = sgn; lm[slot][2][l] = invw; lmj[slot][l] = j;
      }
    }
    LEG_FN void ld_lim(int s, double& pos, double& sgn, double& invw, int& j) const {
      const int l = threadIdx.x;
      pos = lm[s][0][l]; sgn = lm[s][1][l]; invw = lm[s][2][l]; j = lmj[s][l];
    }
  };
  static LEG_FN int leg() { return (int)threadIdx.x & 1; }
#ifdef LEG_NO_FENCE
  static LEG_FN void fence() {}
#else
  static LEG_FN void fence() { __builtin_amdgcn_sched_barrier(0); }
#endif   // nothing is scheduled across this point
  static LEG_FN int opq(int x) { asm volatile("" : "+v"(x)); return x; }     // the value, unknown to the optimiser
  static LEG_FN int zs() { int z = 0; asm volatile("" : "+s"(z)); return z; }  // a wave-uniform zero, unknown to the optimiser
  static LEG_FN double sel(bool m, double a, double b) { return m ? a : b; }
  static LEG_FN int seli(bool m, int a, int b) { return m ? a : b; }
  // exchange with the partner lane (lane ^ 1): DPP quad_perm [1,0,3,2]; both lanes of a pair are always active together
  static LEG_FN int swapi(int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, false); }
  static LEG_FN double swap(double x) { return __hiloint2double(swapi(__double2hiint(x)), swapi(__double2loint(x))); }
  static LEG_FN bool swapm(bool m) { return swapi((int)m) != 0; }
  // both lanes of every pair <- the pair's lane W (0 even = left, 1 odd = right): DPP quad_perm [0,0,2,2] / [1,1,3,3]
  template <int W> static LEG_FN double pair_bcast(double x) {
    constexpr int CTRL = W == 0 ? 0xA0 : 0xF5;
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, false));
  }
  static LEG_FN bool any(bool m) { return __ballot(m) != 0ull; }
  static LEG_FN double ldc(const double* t, int i) { return t[i]; }
  static LEG_FN double ldg(const double* t, int i) { return t[i]; }
  static LEG_FN int toI(bool m) { return (int)m; }
  static LEG_FN double toD(int i) { return (double)i; }
  static LEG_FN int toint(double x) { return (int)x; }
  static LEG_FN void sincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }
  static LEG_FN double sqrt(double x) { return ::sqrt(x); }
  // 1/d to ~1 ulp: hardware seed + two Newton steps (as fast_rcp of cassie_kernels.hip)
  static LEG_FN double rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
  }
  static LEG_FN double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
  static LEG_FN double fabs(double x) { return ::fabs(x); }
  static LEG_FN double fmax(double a, double b) { return ::fmax(a, b); }
  static LEG_FN double exp(double x) { return ::exp(x); }
  static LEG_FN double fmod(double a, double b) { return ::fmod(a, b); }
  static LEG_FN double copysign(double a, double b) { return ::copysign(a, b); }
  static LEG_FN double pld(const double* p, int off) { return p[off]; }
  static LEG_FN void pst(double* p, int off, double v, bool m) { if (m) p[off] = v; }
  static LEG_FN void pst8(uint8_t* p, bool v, bool m) { if (m) *p = (uint8_t)v; }
};

typedef Core<DevB> DCore;

// MODE: 0 PD, 1 torque, 2 motor commands from the state record.  pending[env] = substeps this kernel did NOT do because the
// environment needed more than 8 constraint rows on a leg (0 normally); the packed 16-row kernel / the wave-per-environment
// kernel finish those (cassie_cabi.hip).
template <int MODE>
__global__ void __launch_bounds__(64, 1) env_step_leg_kernel(VecParams p, int* pending) {
  __shared__ DevB::Lds lds;
  const int lane = threadIdx.x;
  const int env = blockIdx.x * 32 + (lane >> 1);
  const bool valid = env < p.n_envs;
  const size_t e = valid ? (size_t)env : 0;
  EnvCfg cfg;
  cfg.n_sub = p.n_sub; cfg.flags = p.flags; cfg.env_kind = p.env_kind; cfg.auto_reset = p.auto_reset; cfg.adim = p.adim;
  cfg.want_obs = p.obs != nullptr; cfg.traj_qpos = p.traj_qpos; cfg.traj_tmax = p.traj_tmax; cfg.traj_n = p.traj_n;
  DCore::Io io;
  io.rec = p.state + e * ENV_STRIDE;
  io.has_act = p.actions != nullptr;
  io.act = const_cast<double*>(p.actions) + (io.has_act ? e * p.adim : 0);
  io.obs = p.obs + (cfg.want_obs ? e * 26 : 0);
  io.has_tobs = p.terminal_obs != nullptr;
  io.tobs = p.terminal_obs + (io.has_tobs ? e * 26 : 0);
  io.rew = p.reward + (cfg.want_obs ? e : 0);
  io.done = p.done + (cfg.want_obs ? e : 0);
#ifdef CASSIE_PHASE_TIMING
  if (lane == 0) { for (int i = 0; i < 16; i++) lds.acc[i] = 0; lds.t_last = __builtin_readcyclecounter(); }
#endif
  DCore::Out o;
  DCore::env_step<MODE>(cfg, lds, io, valid, o);
#ifdef CASSIE_PHASE_TIMING
  lds.mark(0);
  if (lane == 0 && p.phase) for (int i = 0; i < 16; i++) atomicAdd(p.phase + i, lds.acc[i]);
#endif
  if (valid && (lane & 1) == 0) {
    pending[e] = o.pend;
    if (p.stats) {
      if (o.pend > 0) atomicAdd(p.stats + STAT_CLEANUP_SUBSTEPS, (unsigned long long)o.pend);
      if (o.bad) atomicAdd(p.stats + STAT_NONFINITE, 1ull);
    }
  }
}

#ifdef CASSIE_LEG_SEGMENT   // tu_leg_seg.hip only: a separate translation unit, so that the kernel above compiles to what it was
// The same kernel for a SEGMENT of an Env.step (cassie_cabi.hip, launch_physics_tiers while robots are down: the step is cut into a
// few launches so that the lower tiers start on an overflowing environment while the rest of the batch is still being stepped here).
//   gone[env] != 0 on entry (seg.first: never): the environment left this tier in an earlier segment -- not touched, pending = 0;
//   on exit gone[env] = 1 if it left in this one (seg.first: written for everyone, so the array needs no clearing);
//   pending[env] counts to the END of the Env.step (seg.later = substeps of the segments behind this one).
// A separate kernel: the one above is not to be touched (its solver loop's register allocation reacts to anything, 0.8-2.6 %).
struct Segment { int first, later; };
template <int MODE>
__global__ void __launch_bounds__(64, 1) env_step_leg_seg_kernel(VecParams p, int* pending, int* gone, Segment seg) {
  __shared__ DevB::Lds lds;
  const int lane = threadIdx.x;
  const int env = blockIdx.x * 32 + (lane >> 1);
  const bool exists = env < p.n_envs;
  const bool valid = exists && (seg.first || gone[env] == 0);
  const size_t e = valid ? (size_t)env : 0;
  EnvCfg cfg;
  cfg.n_sub = p.n_sub; cfg.flags = p.flags; cfg.env_kind = p.env_kind; cfg.auto_reset = p.auto_reset; cfg.adim = p.adim;
  cfg.want_obs = p.obs != nullptr; cfg.traj_qpos = p.traj_qpos; cfg.traj_tmax = p.traj_tmax; cfg.traj_n = p.traj_n;
  cfg.pend_extra = seg.later; cfg.cont = !seg.first;
  DCore::Io io;
  io.rec = p.state + e * ENV_STRIDE;
  io.has_act = p.actions != nullptr;
  io.act = const_cast<double*>(p.actions) + (io.has_act ? e * p.adim : 0);
  io.obs = p.obs + (cfg.want_obs ? e * 26 : 0);
  io.has_tobs = p.terminal_obs != nullptr;
  io.tobs = p.terminal_obs + (io.has_tobs ? e * 26 : 0);
  io.rew = p.reward + (cfg.want_obs ? e : 0);
  io.done = p.done + (cfg.want_obs ? e : 0);
  DCore::Out o;
  DCore::env_step<MODE>(cfg, lds, io, valid, o);
  if (exists && (lane & 1) == 0) {
    const int pend = valid ? o.pend : 0;
    pending[env] = pend;
    if (seg.first) gone[env] = pend > 0; else if (pend > 0) gone[env] = 1;
    if (p.stats && valid) {
      if (pend > 0) atomicAdd(p.stats + STAT_CLEANUP_SUBSTEPS, (unsigned long long)pend);
      if (o.bad) atomicAdd(p.stats + STAT_NONFINITE, 1ull);
    }
  }
}

// CassieVecReset with the two-lanes-per-environment core (r04): mask [n] or null = everyone, qpos / qvel [n][13] or null = the reset
// pose.  need_slow[env] = 1: the given state needs more than 8 rows on a leg -- untouched here, env_reset_kernel takes it (need_slow as
// its mask).  65 536 environments: one wavefront per 32 of them instead of one each (5.1 ms -> see DESIGN.md K2).
__global__ void __launch_bounds__(64, 1) env_reset_leg_kernel(VecParams p, const uint8_t* mask, const double* qpos_in, const double* qvel_in, uint8_t* need_slow) {
  __shared__ DevB::Lds lds;
  const int lane = threadIdx.x;
  const int env = blockIdx.x * 32 + (lane >> 1);
  const bool exists = env < p.n_envs;
  const bool want = exists && (!mask || mask[env]);
  const size_t e = exists ? (size_t)env : 0;
  EnvCfg cfg;
  cfg.n_sub = 0; cfg.flags = p.flags; cfg.env_kind = p.env_kind; cfg.auto_reset = p.auto_reset; cfg.adim = p.adim;
  cfg.want_obs = p.obs != nullptr; cfg.traj_qpos = p.traj_qpos; cfg.traj_tmax = p.traj_tmax; cfg.traj_n = p.traj_n;
  DCore::Io io;
  io.rec = p.state + e * ENV_STRIDE;
  io.has_act = false; io.act = io.rec;
  io.obs = p.obs + (cfg.want_obs ? e * 26 : 0);
  io.has_tobs = false; io.tobs = io.obs; io.rew = io.rec; io.done = nullptr;
  const bool has_qv = qpos_in != nullptr;
  DCore::Out o;
  DCore::env_reset<false>(cfg, lds, io, want, const_cast<double*>(qpos_in) + (has_qv ? e * 13 : 0), const_cast<double*>(qvel_in) + (has_qv ? e * 13 : 0), has_qv, o);
  if (exists && (lane & 1) == 0) need_slow[env] = (want && o.pend != 0) ? 1 : 0;
}
#endif

#ifdef CASSIE_LEG_HF
// ---------------------------------------------------------------- height-field instantiation (tu_hf.hip)
// The same core with the terrain collision stage (`terrain_sphere`, cassie_kernels.hip, which the including translation unit
// provides): per contact pair one more per-lane LDS slot (the x component of the local terrain normal; nz = sqrt(1 - nx^2) > 0 for
// a height field) -- 40.7 KB per wavefront, still four per CU.  A separate backend and kernel so that the flat-floor kernel above is
// byte-for-byte what it was.
struct DevBHF : DevB {
  struct Lds : DevB::Lds {
    double nrm[3][64];
    LEG_FN void st_nrm(int slot, double nx, bool m) { if (m) nrm[slot][threadIdx.x] = nx; }
    LEG_FN double ld_nrm(int s) const { return nrm[s][threadIdx.x]; }
  };
  static LEG_FN void hf_sphere(const Terrain& t, double wx, double wy, double wz, double radius, double& dist, double& nx, double& nz) {
    cassie::terrain_sphere(t, wx, wy, wz, radius, dist, nx, nz);
  }
};
typedef Core<DevBHF> DCoreHF;

template <int MODE>
__global__ void __launch_bounds__(64, 1) env_step_leg_hf_kernel(VecParams p, int* pending) {
  __shared__ DevBHF::Lds lds;
  const int lane = threadIdx.x;
  const int env = blockIdx.x * 32 + (lane >> 1);
  const bool valid = env < p.n_envs;
  const size_t e = valid ? (size_t)env : 0;
  EnvCfg cfg;
  cfg.n_sub = p.n_sub; cfg.flags = p.flags; cfg.env_kind = p.env_kind; cfg.auto_reset = p.auto_reset; cfg.adim = p.adim;
  cfg.want_obs = p.obs != nullptr; cfg.traj_qpos = p.traj_qpos; cfg.traj_tmax = p.traj_tmax; cfg.traj_n = p.traj_n;
  DCoreHF::Io io;
  io.rec = p.state + e * ENV_STRIDE;
  io.has_act = p.actions != nullptr;
  io.act = const_cast<double*>(p.actions) + (io.has_act ? e * p.adim : 0);
  io.obs = p.obs + (cfg.want_obs ? e * 26 : 0);
  io.has_tobs = p.terminal_obs != nullptr;
  io.tobs = p.terminal_obs + (io.has_tobs ? e * 26 : 0);
  io.rew = p.reward + (cfg.want_obs ? e : 0);
  io.done = p.done + (cfg.want_obs ? e : 0);
  DCoreHF::Out o;
  DCoreHF::env_step<MODE, true>(cfg, lds, io, valid, o, &p.hf);
  if (valid && (lane & 1) == 0) {
    pending[e] = o.pend;
    if (p.stats) {
      if (o.pend > 0) atomicAdd(p.stats + STAT_CLEANUP_SUBSTEPS, (unsigned long long)o.pend);
      if (o.bad) atomicAdd(p.stats + STAT_NONFINITE, 1ull);
    }
  }
}
#endif

}  // namespace leg
}  // namespace cassie
#endif
