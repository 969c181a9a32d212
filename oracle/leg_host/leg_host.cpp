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

#ifndef LEG_HOST_LANES
#define LEG_HOST_LANES 2
#endif

namespace {

constexpr int NL = LEG_HOST_LANES;   // lanes per group: lane l = leg (l & 1) of the group's environment (l >> 1)
static_assert(NL >= 2 && NL % 2 == 0, "two lanes per environment");
#define LANES for (int l = 0; l < NL; l++)
typedef long long i64;   // every lane type is 64 bits wide, so that one AVX-512 register holds eight lanes of any of them

// Op counting (tests/count_flops.py): every arithmetic operation on a lane value adds 1 per COUNTED lane (a*b+c is written as a
// multiply and an add in the core: 2); divisions, square roots and reciprocals count 1, sincos 2, exp 1; comparisons and selects 0.
// Inside a Gauss-Seidel step only the owner leg's lane is counted (the other lane executes the same instructions on values
// that are thrown away); everywhere else both lanes are.  Compiled out of the timing build.
#ifdef LEG_HOST_FAST
inline void ops(int = 1) {}
#else
double g_ops = 0.0;
bool g_cnt[NL];
struct CntInit { CntInit() { LANES g_cnt[l] = true; } } g_cnt_init;
inline void ops(int k = 1) { int c = 0; LANES c += (int)g_cnt[l]; g_ops += k * c; }
#endif

// Lane values are GCC vector-extension types (NL x 64 bit): with NL = 8 and -march=native every operation below is one AVX-512
// instruction; with NL = 2 one SSE2 instruction.  Masks are what vector comparisons give: all-ones / zero per lane.
typedef double vdn __attribute__((vector_size(NL * 8)));
typedef i64 vin __attribute__((vector_size(NL * 8)));
struct VM {
  vin v;
  VM() {}
  VM(bool b) { v = vin{} - (i64)b; }
  VM(vin x) : v(x) {}
};
struct VI {
  vin v;
  VI() {}
  VI(int a) { v = vin{} + (i64)a; }
  VI(vin x) : v(x) {}
};
struct VD {
  vdn v;
  VD() {}
  VD(double a) { v = vdn{} + a; }
  VD(vdn x) : v(x) {}
};
#define VD_BIN(op) inline VD operator op(const VD& a, const VD& b) { ops(); return VD(a.v op b.v); }
VD_BIN(+) VD_BIN(-) VD_BIN(*) VD_BIN(/)
inline VD operator-(const VD& a) { return VD(-a.v); }
inline VD& operator+=(VD& a, const VD& b) { a = a + b; return a; }
#define VD_CMP(op) inline VM operator op(const VD& a, const VD& b) { return VM((vin)(a.v op b.v)); }
VD_CMP(<) VD_CMP(>) VD_CMP(<=) VD_CMP(>=) VD_CMP(==)
#define VI_BIN(op) inline VI operator op(const VI& a, const VI& b) { return VI(a.v op b.v); }
VI_BIN(+) VI_BIN(-) VI_BIN(*)
#define VI_CMP(op) inline VM operator op(const VI& a, const VI& b) { return VM((vin)(a.v op b.v)); }
VI_CMP(<) VI_CMP(>) VI_CMP(<=) VI_CMP(>=) VI_CMP(==) VI_CMP(!=)
inline VM operator&(const VM& a, const VM& b) { return VM(a.v & b.v); }
inline VM operator|(const VM& a, const VM& b) { return VM(a.v | b.v); }
inline VM operator!(const VM& a) { return VM(~a.v); }
inline vin lane_swap_idx() { vin r; LANES r[l] = l ^ 1; return r; }
template <int W> inline vin lane_bcast_idx() { vin r; LANES r[l] = (l & ~1) | W; return r; }

struct HostB {
  typedef VD D;
  typedef VI I;
  typedef VM M;
#ifdef LEG_HOST_FAST
  struct OwnerScope { OwnerScope(const VM&) {} };
#else
  struct OwnerScope {
    bool old[NL];
    OwnerScope(VM owner) { LANES { old[l] = g_cnt[l]; g_cnt[l] = owner.v[l] != 0; } }
    ~OwnerScope() { LANES g_cnt[l] = old[l]; }
  };
#endif
  struct K { const double* p[NL]; };
  static K kbase(VI leg) { K k; LANES k.p[l] = &cp_legk[0][0] + leg.v[l] * LK_N; return k; }
  static VD kld(K k, int idx) { VD r; LANES r.v[l] = k.p[l][idx]; return r; }
  struct P { double* p[NL]; };
  struct P8 { uint8_t* p[NL]; };
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
  static VI leg() { VI r; LANES r.v[l] = l & 1; return r; }
  static VI opq(VI x) { return x; }
  static void fence() {}
  static int zs() { return 0; }
  static VD sel(VM m, VD a, VD b) { return VD(m.v ? a.v : b.v); }
  static VI seli(VM m, VI a, VI b) { return VI(m.v ? a.v : b.v); }
  static VD swap(VD x) { return VD(__builtin_shuffle(x.v, lane_swap_idx())); }
  template <int W> static VD pair_bcast(VD x) { return VD(__builtin_shuffle(x.v, lane_bcast_idx<W>())); }
  static VM swapm(VM x) { return VM(__builtin_shuffle(x.v, lane_swap_idx())); }
  static bool any(VM m) { i64 a = 0; LANES a |= m.v[l]; return a != 0; }
  static VD ldc(const double* t, VI i) { VD r; LANES r.v[l] = t[i.v[l]]; return r; }
  static VD ldg(const double* t, VI i) { return ldc(t, i); }
  static VI toI(VM m) { return VI(m.v & 1); }
  static VD toD(VI i) { VD r; LANES r.v[l] = (double)i.v[l]; return r; }
  static VI toint(VD x) { VI r; LANES r.v[l] = (int)x.v[l]; return r; }
  static void sincos(VD x, VD& s, VD& c) { ops(2); LANES { s.v[l] = std::sin(x.v[l]); c.v[l] = std::cos(x.v[l]); } }
  static VD sqrt(VD x) { ops(); VD r; LANES r.v[l] = std::sqrt(x.v[l]); return r; }
  static VD rcp(VD x) { ops(); return VD(1.0 / x.v); }
  static VD fma(VD a, VD b, VD c) { ops(2); VD r; LANES r.v[l] = __builtin_fma(a.v[l], b.v[l], c.v[l]); return r; }
  static VD fabs(VD x) { return VD((vdn)((vin)x.v & (vin{} + (i64)0x7fffffffffffffffLL))); }
  static VD fmax(VD a, VD b) { return VD(a.v > b.v ? a.v : b.v); }   // operands are never NaN here
  static VD exp(VD x) { ops(); VD r; LANES r.v[l] = std::exp(x.v[l]); return r; }
  static VD fmod(VD a, double b) { VD r; LANES r.v[l] = std::fmod(a.v[l], b); return r; }
  static VD copysign(VD a, VD b) {
    const vin sign = vin{} + (i64)0x8000000000000000ULL;
    return VD((vdn)(((vin)a.v & ~sign) | ((vin)b.v & sign)));
  }
  static VD pld(P p, VI off) { VD r; LANES r.v[l] = p.p[l][off.v[l]]; return r; }
  static void pst(P p, VI off, VD v, VM m) { LANES if (m.v[l]) p.p[l][off.v[l]] = v.v[l]; }
  static void pst8(P8 p, VM v, VM m) { LANES if (m.v[l]) *p.p[l] = v.v[l] ? 1 : 0; }
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
