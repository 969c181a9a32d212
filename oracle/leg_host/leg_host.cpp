// leg_host.cpp -- CHECKER / CPU-BASELINE INFRASTRUCTURE (lives under oracle/): compiles cassierl_amd/csrc/cassie_leg_core.h (the
// two-lanes-per-environment Env.step of the HIP kernel cassie_kernels_leg.hip) for the CPU with a lane emulation backend, so that
//   (a) the CPU test-suite can check the kernel's SOURCE against the oracle before anything runs on a GPU (tests/test_leg_host.py;
//       LEG_HOST_LANES = 2: one environment per call group, operation counting for tools/count_flops.py), and
//   (b) bench.py's cpu_baseline leg can time the SAME SOURCE as the HIP kernel on the host cores (BASELINE.md section 3 / SURVEY.md
//       8(d): "-O3 -march=native, OpenMP over envs"): -DLEG_HOST_FAST -DLEG_HOST_LANES=8 puts four environments into the lanes of
//       one AVX-512 register (the loops over lanes below are what the compiler vectorises), no operation counting, OpenMP over
//       groups of environments.
// Only tests/, bench.py's cpu_baseline leg and tools/count_flops.py build and load this; the product (cassierl_amd/) has no CPU path.
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __constant__
#define __forceinline__ inline
#define LEG_FN inline
#define LEG_NOUNROLL
#define LEG_FP_CONTRACT_OFF   /* the whole file is compiled with -ffp-contract=off; the step functions write their FMAs out */
#ifdef LEG_STATS   // tools/small_stats.py: per set-up of a lane group: [0] set-ups, [1] not "small", [2] lanes with a joint limit, [3] lanes with a third pair
#include <atomic>
static std::atomic<long long> g_small_stat[8];   // [4] set-ups whose worst lane has <= 1 limit and <= 2 pairs, [5] lanes with >= 2 limits, [6] set-ups with any lane over 8 rows
#define LEG_STAT_SMALL(small, go, nlim, ncon) do { g_small_stat[0]++; if (!(small)) g_small_stat[1]++; bool one_ = true, ovf_ = false; \
  for (int l_ = 0; l_ < LEG_HOST_LANES; l_++) { if ((go).v[l_] && (nlim).v[l_] > 0) g_small_stat[2]++; if ((go).v[l_] && (ncon).v[l_] > 2) g_small_stat[3]++; \
    if ((go).v[l_] && (nlim).v[l_] > 1) g_small_stat[5]++; if ((go).v[l_] && ((nlim).v[l_] > 1 || (ncon).v[l_] > 2)) one_ = false; } \
  if (one_) g_small_stat[4]++; } while (0)
#endif
#include "../../cassierl_amd/csrc/cassie_leg_core.h"
#include "../../cassierl_amd/csrc/cassie_duo_core.h"

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


// 64-environments-per-wavefront form (cassie_duo_core.h; the counterpart of DevDuoB in cassie_kernels_duo.hip): NL lanes carry NL / 2
// environments of group A and NL / 2 of group B; in the joint sweep every lane holds one environment.  The cold slots are routed as on the
// device: setState snapshot / qstate / motor commands to the record, the action from its row, the rest to per-group arrays.
struct HostDuoB : HostB {
  static constexpr bool SPLIT_TAIL = false;
  struct W { double (*p)[NL]; };
  static VD wld(W ws, int slot) { VD r; LANES r.v[l] = ws.p[slot][l]; return r; }
  static void wst(W ws, int slot, VD v) { LANES ws.p[slot][l] = v.v[l]; }
  static void wld2(W ws, int slot, VD& a, VD& b) { a = wld(ws, slot); b = wld(ws, slot + 1); }
  static void wst2(W ws, int slot, VD a, VD b) { wst(ws, slot, a); wst(ws, slot + 1, b); }
  struct Lds {
    double cold[2][20][NL];
    double pr[2][2][4][NL], pr2[4][NL], lm[4][3][NL];
    i64 pdepth[2][2][NL], pdepth2[NL], lmj[4][NL];
    int g;
    double* rec[NL];
    const double* act[NL];
    bool has_act, snap;
    void mark(int) {}
    template <class IoT> void select(int group, const IoT& io) { g = group; LANES { rec[l] = io.rec.p[l]; act[l] = io.act.p[l]; } has_act = io.has_act; }
    void snapshot(bool on) { snap = on; }
    static constexpr int slot(int i) {
      return i == 24 ? 0 : i == 28 ? 1 : (i >= 29 && i < 37) ? i - 27 : (i >= 38 && i < 43) ? i - 28 : (i >= 44 && i < 49) ? i - 29 : -1;
    }
    static int lo(int l) { return (l & 1) * 5 + 3; }
    static int ao(int l) { return (l & 1) * 3; }
    VD cld(int i) const {
      VD r;
      LANES {
        if (i < 8) r.v[l] = rec[l][cassie::ES_KQ + (i < 3 ? i : lo(l) + (i - 3))];
        else if (i < 16) r.v[l] = rec[l][cassie::ES_KV + (i - 8 < 3 ? i - 8 : lo(l) + (i - 11))];
        else if (i < 21) r.v[l] = rec[l][cassie::ES_QSTATE + lo(l) + (i - 16)];
        else if (i < 24) r.v[l] = rec[l][cassie::ES_CTRL + ao(l) + (i - 21)];
        else if (i >= 25 && i < 28) r.v[l] = has_act ? act[l][ao(l) + (i - 25)] : 0.0;
        else if (i == 37 || i == 43) r.v[l] = 0.0;
        else r.v[l] = cold[g][slot(i)][l];
      }
      return r;
    }
    void cst(int i, VD v, VM m) {
      LANES {
        if (!m.v[l]) continue;
        if (i < 8) { if (snap) rec[l][cassie::ES_KQ + (i < 3 ? i : lo(l) + (i - 3))] = v.v[l]; }
        else if (i < 16) { if (snap) rec[l][cassie::ES_KV + (i - 8 < 3 ? i - 8 : lo(l) + (i - 11))] = v.v[l]; }
        else if (i < 21) rec[l][cassie::ES_QSTATE + lo(l) + (i - 16)] = v.v[l];
        else if (i < 24) rec[l][cassie::ES_CTRL + ao(l) + (i - 21)] = v.v[l];
        else if (i >= 25 && i < 28) {}
        else if (i == 37 || i == 43) {}
        else cold[g][slot(i)][l] = v.v[l];
      }
    }
    void st_pair(VI slot_, VD px, VD pz, VD dist, VD invw, VI depth, VM m) {
      LANES if (m.v[l]) {
        const int s = (int)slot_.v[l];
        double (*q)[NL] = s < 2 ? pr[g][s] : pr2;
        q[0][l] = px.v[l]; q[1][l] = pz.v[l]; q[2][l] = dist.v[l]; q[3][l] = invw.v[l];
        (s < 2 ? pdepth[g][s] : pdepth2)[l] = depth.v[l];
      }
    }
    void ld_pair(int s, VD& px, VD& pz, VD& dist, VD& invw, VI& depth) {
      const double (*q)[NL] = s < 2 ? pr[g][s] : pr2;
      LANES { px.v[l] = q[0][l]; pz.v[l] = q[1][l]; dist.v[l] = q[2][l]; invw.v[l] = q[3][l]; depth.v[l] = (s < 2 ? pdepth[g][s] : pdepth2)[l]; }
    }
    void st_lim(VI slot_, VD pos, VD sgn, VD invw, VI j, VM m) {
      LANES if (m.v[l]) { const int s = (int)slot_.v[l]; lm[s][0][l] = pos.v[l]; lm[s][1][l] = sgn.v[l]; lm[s][2][l] = invw.v[l]; lmj[s][l] = j.v[l]; }
    }
    void ld_lim(int s, VD& pos, VD& sgn, VD& invw, VI& j) {
      LANES { pos.v[l] = lm[s][0][l]; sgn.v[l] = lm[s][1][l]; invw.v[l] = lm[s][2][l]; j.v[l] = lmj[s][l]; }
    }
  };
};
typedef cassie::leg::Duo<HostDuoB> HDuo;
// ... on a height field (the counterpart of DevDuoBHF): one more per-lane word per contact pair
struct HostDuoBHF : HostDuoB {
  struct Lds : HostDuoB::Lds {
    double nrm[2][2][NL], nrm2[NL];
    void st_nrm(VI slot_, VD nx, VM m) { LANES if (m.v[l]) { const int s = (int)slot_.v[l]; (s < 2 ? nrm[g][s] : nrm2)[l] = nx.v[l]; } }
    VD ld_nrm(int s) const { VD r; LANES r.v[l] = (s < 2 ? nrm[g][s] : nrm2)[l]; return r; }
  };
  static void hf_sphere(const cassie::Terrain& t, VD wx, VD wy, VD wz, VD radius, VD& dist, VD& nx, VD& nz) {
    LANES cassie::terrain_sphere(t, wx.v[l], wy.v[l], wz.v[l], radius.v[l], dist.v[l], nx.v[l], nz.v[l]);
  }
};
typedef cassie::leg::Duo<HostDuoBHF> HDuoHF;

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


// The 64-environments-per-wavefront form: one call group = NL lanes = NL / 2 environments of group A and the next NL / 2 of group B.
template <class DuoT, class BT, bool HF>
int run_duo(double* state, const double* actions, int n, int adim, int mode, int n_sub, int flags, int env_kind, int auto_reset,
            const double* traj_qpos, double traj_tmax, int traj_n, const cassie::Terrain* hf, double* obs, double* reward, uint8_t* done, double* terminal_obs,
            int* pending, int* nonfinite, int threads) {
  cassie::leg::EnvCfg cfg;
  cfg.n_sub = n_sub; cfg.flags = flags; cfg.env_kind = env_kind; cfg.auto_reset = auto_reset; cfg.adim = adim;
  cfg.want_obs = obs != nullptr; cfg.traj_qpos = traj_qpos; cfg.traj_tmax = traj_tmax; cfg.traj_n = traj_n;
  constexpr int EPG = NL / 2;
  const int calls = (n + 2 * EPG - 1) / (2 * EPG);
  int bad = 0;
  (void)threads;
#ifdef LEG_HOST_FAST
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(threads > 0 ? threads : 1)
#endif
  for (int c = 0; c < calls; c++) {
    double dummy[32] = {0};
    uint8_t dummy8 = 0;
    typename BT::Lds lds;
    std::memset(&lds, 0, sizeof lds);
    lds.snap = true;
    if constexpr (HF) { for (auto& a : lds.nrm) for (auto& b : a) LANES b[l] = std::nan(""); LANES lds.nrm2[l] = std::nan(""); }   // an unused slot holds anything
    typename DuoT::Io io[2];
    VM valid[2];
    for (int g = 0; g < 2; g++) {
      const int e0 = c * 2 * EPG + g * EPG;
      const int eb = e0 < n ? e0 : 0;   // a group past the end reads environment 0 and writes nothing
      point(io[g].rec, state, cassie::ENV_STRIDE, eb, n);
      io[g].has_act = actions != nullptr;
      if (actions) point(io[g].act, const_cast<double*>(actions), adim, eb, n); else LANES io[g].act.p[l] = dummy;
      if (obs) point(io[g].obs, obs, 26, eb, n); else LANES io[g].obs.p[l] = dummy;
      io[g].has_tobs = terminal_obs != nullptr;
      if (terminal_obs) point(io[g].tobs, terminal_obs, 26, eb, n); else LANES io[g].tobs.p[l] = dummy;
      if (reward) point(io[g].rew, reward, 1, eb, n); else LANES io[g].rew.p[l] = dummy;
      LANES { const int e = e0 + (l >> 1); io[g].done.p[l] = done && e < n ? done + e : &dummy8; }
      LANES valid[g].v[l] = e0 + (l >> 1) < n ? -1 : 0;
    }
    lds.select(0, io[0]);
    typename DuoT::Out o[2];
    double wsmem[DuoT::W_N][NL];
    for (int k = 0; k < DuoT::W_N; k++) LANES wsmem[k][l] = std::nan("");   // whatever a launch finds there
    typename BT::W ws; ws.p = wsmem;
    auto io_of = [&](int g) -> const typename DuoT::Io& { return io[g]; };
    if (mode == 0) DuoT::template env_step2<0, HF>(cfg, lds, ws, io_of, valid, o, hf);
    else if (mode == 1) DuoT::template env_step2<1, HF>(cfg, lds, ws, io_of, valid, o, hf);
#ifndef LEG_HOST_FAST
    else DuoT::template env_step2<2, HF>(cfg, lds, ws, io_of, valid, o, hf);
#endif
    for (int g = 0; g < 2; g++)
      for (int k = 0; k < EPG; k++) {
        const int e = c * 2 * EPG + g * EPG + k;
        if (e >= n) break;
        if (pending) pending[e] = (int)o[g].pend.v[2 * k];
        if (o[g].bad.v[2 * k]) bad++;
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

// ... through the 64-environments-per-wavefront form of the kernel (cassie_duo_core.h): same arguments, bit-identical results
int leg_host_step_duo(double* state, const double* actions, int n, int adim, int mode, int n_sub, int flags, int env_kind, int auto_reset,
                      const double* traj_qpos, double traj_tmax, int traj_n, double* obs, double* reward, uint8_t* done, double* terminal_obs,
                      int* pending, int* nonfinite, int threads) {
  return run_duo<HDuo, HostDuoB, false>(state, actions, n, adim, mode, n_sub, flags, env_kind, auto_reset, traj_qpos, traj_tmax, traj_n, nullptr, obs, reward, done,
                                        terminal_obs, pending, nonfinite, threads);
}
#ifndef LEG_HOST_FAST
int leg_host_step_duo_hf(double* state, const double* actions, int n, int adim, int mode, int n_sub, int flags, int env_kind, int auto_reset,
                         const double* heights, int nrow, int ncol, double sx, double sy, double* obs, double* reward, uint8_t* done, int* pending,
                         int* nonfinite, int threads) {
  cassie::Terrain hf; hf.h = heights; hf.nrow = nrow; hf.ncol = ncol; hf.sx = sx; hf.sy = sy;
  hf.hmax = heights[0];
  for (size_t i = 1; i < (size_t)nrow * ncol; i++) hf.hmax = heights[i] > hf.hmax ? heights[i] : hf.hmax;
  return run_duo<HDuoHF, HostDuoBHF, true>(state, actions, n, adim, mode, n_sub, flags, env_kind, auto_reset, nullptr, 0.0, 0, &hf, obs, reward, done, nullptr,
                                           pending, nonfinite, threads);
}
#endif

int leg_host_lanes(void) { return NL; }
#ifdef LEG_STATS
void leg_host_small_stats(long long* out8) { for (int i = 0; i < 8; i++) { out8[i] = g_small_stat[i]; g_small_stat[i] = 0; } }
#endif

// arithmetic operations counted since the last call (see the note at g_ops); 0 in the timing build
double leg_host_ops(void) {
#ifdef LEG_HOST_FAST
  return 0.0;
#else
  double r = g_ops; g_ops = 0.0; return r;
#endif
}

}  // extern "C"
