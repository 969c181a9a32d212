// leg_host.cpp -- CHECKER / CPU-BASELINE INFRASTRUCTURE (lives under oracle/): compiles cassierl_amd/csrc/cassie_leg_core.h (the
// two-lanes-per-environment Env.step of the HIP kernel cassie_kernels_leg.hip) for the CPU with a lane emulation backend, so that
//   (a) the CPU test-suite can check the kernel's SOURCE against the oracle before anything runs on a GPU (tests/test_leg_host.py;
//       LEG_HOST_LANES = 2: one environment per call group, operation counting for tests/count_flops.py), and
//   (b) bench.py's cpu_baseline leg can time the SAME SOURCE as the HIP kernel on the host cores (BASELINE.md section 3 / SURVEY.md
//       8(d): "-O3 -march=native, OpenMP over envs"): -DLEG_HOST_FAST -DLEG_HOST_LANES=8 puts four environments into the lanes of
//       one AVX-512 register (the loops over lanes below are what the compiler vectorises), no operation counting, OpenMP over
//       groups of environments.
// Only tests/, bench.py's cpu_baseline leg and tests/count_flops.py build and load this; the product (cassierl_amd/) has no CPU path.
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __constant__
#define __forceinline__ inline
#define LEG_FN inline
#define LEG_FP_CONTRACT_OFF   /* the whole file is compiled with -ffp-contract=off; the step functions write their FMAs out */
#include "../../cassierl_amd/csrc/cassie_leg_core.h"

#include "lane_types.h"

namespace {

struct HostB : HostOps {
  struct K { const double* p[NL]; };
  static K kbase(VI leg) { K k; LANES k.p[l] = &cp_legk[0][0] + leg.v[l] * LK_N; return k; }
  static VD kld(K k, int idx) { VD r; LANES r.v[l] = k.p[l][idx]; return r; }
  struct Lds {
    double pr[LEG_NPAIR_SLOTS][4][NL]; i64 pdepth[LEG_NPAIR_SLOTS][NL];
    double lm[4][3][NL]; i64 lmj[4][NL];
    double cold[cassie::leg::Core<HostB>::C_N][NL];
    void mark(int) {}
    VD cld(int i) const { VD r; LANES r.v[l] = cold[i][l]; return r; }
    void cst(int i, VD v, VM m) { LANES if (m.v[l]) cold[i][l] = v.v[l]; }
    void st_pair(VI slot, VD px, VD pz, VD dist, VD invw, VI depth, VM m) {
      LANES if (m.v[l]) { const int s = (int)slot.v[l]; pr[s][0][l] = px.v[l]; pr[s][1][l] = pz.v[l]; pr[s][2][l] = dist.v[l]; pr[s][3][l] = invw.v[l]; pdepth[s][l] = depth.v[l]; }
    }
    void ld_pair(int s, VD& px, VD& pz, VD& dist, VD& invw, VI& depth) {
      LANES { px.v[l] = pr[s][0][l]; pz.v[l] = pr[s][1][l]; dist.v[l] = pr[s][2][l]; invw.v[l] = pr[s][3][l]; depth.v[l] = pdepth[s][l]; }
    }
    void st_lim(VI slot, VD pos, VD sgn, VD invw, VI j, VM m) {
      LANES if (m.v[l]) { const int s = (int)slot.v[l]; lm[s][0][l] = pos.v[l]; lm[s][1][l] = sgn.v[l]; lm[s][2][l] = invw.v[l]; lmj[s][l] = j.v[l]; }
    }
    void ld_lim(int s, VD& pos, VD& sgn, VD& invw, VI& j) {
      LANES { pos.v[l] = lm[s][0][l]; sgn.v[l] = lm[s][1][l]; invw.v[l] = lm[s][2][l]; j.v[l] = lmj[s][l]; }
    }
  };
};

typedef cassie::leg::Core<HostB> HCore;

// Height-field instantiation (the counterpart of DevBHF in cassie_kernels_leg.hip): one more per-lane slot per contact pair and the
// terrain test.  `hf_sphere` calls the SAME terrain_sphere source as the kernels (cassie_terrain.h).
struct HostBHF : HostB {
  struct Lds : HostB::Lds {
    double nrm[LEG_NPAIR_SLOTS][NL];
    void st_nrm(VI slot, VD nx, VM m) { LANES if (m.v[l]) nrm[slot.v[l]][l] = nx.v[l]; }
    VD ld_nrm(int s) const { VD r; LANES r.v[l] = nrm[s][l]; return r; }
  };
  static void hf_sphere(const cassie::Terrain& t, VD wx, VD wy, VD wz, VD radius, VD& dist, VD& nx, VD& nz) {
    LANES cassie::terrain_sphere(t, wx.v[l], wy.v[l], wz.v[l], radius.v[l], dist.v[l], nx.v[l], nz.v[l]);
  }
};
typedef cassie::leg::Core<HostBHF> HCoreHF;

// lanes of the group that starts at environment e0: lane l works on environment e0 + l / 2 (a lane past the end reads the group's
// first environment and writes nothing)
template <class T> void point(T& io_p, double* base, size_t stride, int e0, int n) {
  LANES { const int e = e0 + (l >> 1); io_p.p[l] = base + (size_t)(e < n ? e : e0) * stride; }
}

template <class C, class LdsT, bool HF>
int run(double* state, const double* actions, int n, int adim, int mode, int n_sub, int flags, int env_kind, int auto_reset,
        const double* traj_qpos, double traj_tmax, int traj_n, const cassie::Terrain* hf, double* obs, double* reward, uint8_t* done,
        double* terminal_obs, int* pending, int* nonfinite, int threads) {
  cassie::leg::EnvCfg cfg;
  cfg.n_sub = n_sub; cfg.flags = flags; cfg.env_kind = env_kind; cfg.auto_reset = auto_reset; cfg.adim = adim;
  cfg.want_obs = obs != nullptr; cfg.traj_qpos = traj_qpos; cfg.traj_tmax = traj_tmax; cfg.traj_n = traj_n;
  constexpr int EPG = NL / 2;
  const int groups = (n + EPG - 1) / EPG;
  int bad = 0;
  (void)threads;
#ifdef LEG_HOST_FAST
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(threads > 0 ? threads : 1)
#endif
  for (int g = 0; g < groups; g++) {
    const int e0 = g * EPG;
    double dummy[32] = {0};
    uint8_t dummy8 = 0;
    LdsT lds;
    std::memset(&lds, 0, sizeof lds);
    if constexpr (HF) { for (int s = 0; s < LEG_NPAIR_SLOTS; s++) LANES lds.nrm[s][l] = std::nan(""); }   // an unused slot holds anything (r03: a NaN there leaked once)
    typename C::Io io;
    point(io.rec, state, cassie::ENV_STRIDE, e0, n);
    io.has_act = actions != nullptr;
    if (actions) point(io.act, const_cast<double*>(actions), adim, e0, n); else LANES io.act.p[l] = dummy;
    if (obs) point(io.obs, obs, 26, e0, n); else LANES io.obs.p[l] = dummy;
    io.has_tobs = terminal_obs != nullptr;
    if (terminal_obs) point(io.tobs, terminal_obs, 26, e0, n); else LANES io.tobs.p[l] = dummy;
    if (reward) point(io.rew, reward, 1, e0, n); else LANES io.rew.p[l] = dummy;
    LANES { const int e = e0 + (l >> 1); io.done.p[l] = done && e < n ? done + e : &dummy8; }
    VM valid; LANES valid.v[l] = e0 + (l >> 1) < n ? -1 : 0;
    typename C::Out o;
#ifdef LEG_HOST_FAST   // the timing build only carries what bench.py times: PD and torque mode on the flat floor
    if (mode == 0) C::template env_step<0, HF>(cfg, lds, io, valid, o, hf);
    else C::template env_step<1, HF>(cfg, lds, io, valid, o, hf);
#else
    if (mode == 0) C::template env_step<0, HF>(cfg, lds, io, valid, o, hf);
    else if (mode == 1) C::template env_step<1, HF>(cfg, lds, io, valid, o, hf);
    else C::template env_step<2, HF>(cfg, lds, io, valid, o, hf);
#endif
    for (int k = 0; k < EPG; k++) {
      if (e0 + k >= n) break;
      if (pending) pending[e0 + k] = (int)o.pend.v[2 * k];
      if (o.bad.v[2 * k]) bad++;
    }
  }
  if (nonfinite) *nonfinite += bad;
  return 0;
}

}  // namespace

extern "C" {

// One Env.step (or n_sub bare substeps when obs == null) of n environments on host arrays laid out like the device ones.
// pending[e] = substeps NOT done because the environment left the rows-per-leg capacity (its state is untouched from there on).
// threads: OpenMP threads over groups of environments (timing build only; the counting build is serial).
int leg_host_step(double* state, const double* actions, int n, int adim, int mode, int n_sub, int flags, int env_kind, int auto_reset,
                  const double* traj_qpos, double traj_tmax, int traj_n, double* obs, double* reward, uint8_t* done, double* terminal_obs,
                  int* pending, int* nonfinite, int threads) {
  return run<HCore, HostB::Lds, false>(state, actions, n, adim, mode, n_sub, flags, env_kind, auto_reset, traj_qpos, traj_tmax, traj_n, nullptr,
                                       obs, reward, done, terminal_obs, pending, nonfinite, threads);
}

#ifndef LEG_HOST_FAST
// The same on a height field (heights[nrow][ncol] metres over [-sx, sx] x [-sy, sy]): cassie_leg_core.h with HF = true.
int leg_host_step_hf(double* state, const double* actions, int n, int adim, int mode, int n_sub, int flags, int env_kind, int auto_reset,
                     const double* heights, int nrow, int ncol, double sx, double sy, double* obs, double* reward, uint8_t* done, int* pending,
                     int* nonfinite, int threads) {
  cassie::Terrain hf; hf.h = heights; hf.nrow = nrow; hf.ncol = ncol; hf.sx = sx; hf.sy = sy;
  hf.hmax = heights[0];
  for (size_t i = 1; i < (size_t)nrow * ncol; i++) hf.hmax = heights[i] > hf.hmax ? heights[i] : hf.hmax;
  return run<HCoreHF, HostBHF::Lds, true>(state, actions, n, adim, mode, n_sub, flags, env_kind, auto_reset, nullptr, 0.0, 0, &hf,
                                          obs, reward, done, nullptr, pending, nonfinite, threads);
}
#endif

int leg_host_lanes(void) { return NL; }

// arithmetic operations counted since the last call (see the note at g_ops); 0 in the timing build
double leg_host_ops(void) {
#ifdef LEG_HOST_FAST
  return 0.0;
#else
  double r = g_ops; g_ops = 0.0; return r;
#endif
}

}  // extern "C"
