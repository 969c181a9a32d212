// leg_host.cpp -- TEST INFRASTRUCTURE: compiles cassierl_amd/csrc/cassie_leg_core.h (the two-lanes-per-environment Env.step of
// the HIP kernel cassie_kernels_leg.hip) for the CPU with a two-lane emulation backend, one environment at a time, so that the
// CPU test-suite can check the kernel's source against the oracle before anything runs on a GPU (tests/test_leg_host.py).
// Only tests/ build and load this; the product (cassierl_amd/) has no CPU path.
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __constant__
#define __forceinline__ inline
#define LEG_FN inline
#define LEG_FP_CONTRACT_OFF   /* the emulation is compiled with -ffp-contract=off */
#include "../../cassierl_amd/csrc/cassie_leg_core.h"

namespace {

// Op counting (tests/count_flops.py): every arithmetic operation on a lane value adds 1 per COUNTED lane (a*b+c is written as a
// multiply and an add in the core: 2); divisions, square roots and reciprocals count 1, sincos 2, exp 1; comparisons and selects 0.
// Inside a Gauss-Seidel step only the owner leg's lane is counted (the other lane executes the same instructions on values
// that are thrown away); everywhere else both lanes are.
double g_ops = 0.0;
bool g_cnt[2] = {true, true};
inline void ops(int k = 1) { g_ops += k * ((int)g_cnt[0] + (int)g_cnt[1]); }

struct VM {
  bool v[2];
  VM() {}
  VM(bool b) { v[0] = v[1] = b; }
};
struct VI {
  int v[2];
  VI() {}
  VI(int a) { v[0] = v[1] = a; }
};
struct VD {
  double v[2];
  VD() {}
  VD(double a) { v[0] = v[1] = a; }
};
#define VD_BIN(op) inline VD operator op(const VD& a, const VD& b) { ops(); VD r; r.v[0] = a.v[0] op b.v[0]; r.v[1] = a.v[1] op b.v[1]; return r; }
VD_BIN(+) VD_BIN(-) VD_BIN(*) VD_BIN(/)
inline VD operator-(const VD& a) { VD r; r.v[0] = -a.v[0]; r.v[1] = -a.v[1]; return r; }
inline VD& operator+=(VD& a, const VD& b) { a = a + b; return a; }
#define VD_CMP(op) inline VM operator op(const VD& a, const VD& b) { VM r; r.v[0] = a.v[0] op b.v[0]; r.v[1] = a.v[1] op b.v[1]; return r; }
VD_CMP(<) VD_CMP(>) VD_CMP(<=) VD_CMP(>=) VD_CMP(==)
#define VI_BIN(op) inline VI operator op(const VI& a, const VI& b) { VI r; r.v[0] = a.v[0] op b.v[0]; r.v[1] = a.v[1] op b.v[1]; return r; }
VI_BIN(+) VI_BIN(-) VI_BIN(*)
#define VI_CMP(op) inline VM operator op(const VI& a, const VI& b) { VM r; r.v[0] = a.v[0] op b.v[0]; r.v[1] = a.v[1] op b.v[1]; return r; }
VI_CMP(<) VI_CMP(>) VI_CMP(<=) VI_CMP(>=) VI_CMP(==) VI_CMP(!=)
inline VM operator&(const VM& a, const VM& b) { VM r; r.v[0] = a.v[0] && b.v[0]; r.v[1] = a.v[1] && b.v[1]; return r; }
inline VM operator|(const VM& a, const VM& b) { VM r; r.v[0] = a.v[0] || b.v[0]; r.v[1] = a.v[1] || b.v[1]; return r; }
inline VM operator!(const VM& a) { VM r; r.v[0] = !a.v[0]; r.v[1] = !a.v[1]; return r; }

struct HostB {
  typedef VD D;
  typedef VI I;
  typedef VM M;
  struct OwnerScope {
    bool old[2];
    OwnerScope(VM owner) { old[0] = g_cnt[0]; old[1] = g_cnt[1]; g_cnt[0] = owner.v[0]; g_cnt[1] = owner.v[1]; }
    ~OwnerScope() { g_cnt[0] = old[0]; g_cnt[1] = old[1]; }
  };
  struct K { const double* p[2]; };
  static K kbase(VI leg) { K k; k.p[0] = &cp_legk[0][0] + leg.v[0] * LK_N; k.p[1] = &cp_legk[0][0] + leg.v[1] * LK_N; return k; }
  static VD kld(K k, int idx) { VD r; r.v[0] = k.p[0][idx]; r.v[1] = k.p[1][idx]; return r; }
  struct P { double* p[2]; };
  struct P8 { uint8_t* p[2]; };
  struct Lds {
    double pr[3][4][2]; int pdepth[3][2];
    double lm[4][3][2]; int lmj[4][2];
    double cold[cassie::leg::Core<HostB>::C_N][2];
    void mark(int) {}
    VD cld(int i) const { VD r; r.v[0] = cold[i][0]; r.v[1] = cold[i][1]; return r; }
    void cst(int i, VD v, VM m) { for (int l = 0; l < 2; l++) if (m.v[l]) cold[i][l] = v.v[l]; }
    void st_pair(VI slot, VD px, VD pz, VD dist, VD invw, VI depth, VM m) {
      for (int l = 0; l < 2; l++) if (m.v[l]) { int s = slot.v[l]; pr[s][0][l] = px.v[l]; pr[s][1][l] = pz.v[l]; pr[s][2][l] = dist.v[l]; pr[s][3][l] = invw.v[l]; pdepth[s][l] = depth.v[l]; }
    }
    void ld_pair(int s, VD& px, VD& pz, VD& dist, VD& invw, VI& depth) {
      for (int l = 0; l < 2; l++) { px.v[l] = pr[s][0][l]; pz.v[l] = pr[s][1][l]; dist.v[l] = pr[s][2][l]; invw.v[l] = pr[s][3][l]; depth.v[l] = pdepth[s][l]; }
    }
    void st_lim(VI slot, VD pos, VD sgn, VD invw, VI j, VM m) {
      for (int l = 0; l < 2; l++) if (m.v[l]) { int s = slot.v[l]; lm[s][0][l] = pos.v[l]; lm[s][1][l] = sgn.v[l]; lm[s][2][l] = invw.v[l]; lmj[s][l] = j.v[l]; }
    }
    void ld_lim(int s, VD& pos, VD& sgn, VD& invw, VI& j) {
      for (int l = 0; l < 2; l++) { pos.v[l] = lm[s][0][l]; sgn.v[l] = lm[s][1][l]; invw.v[l] = lm[s][2][l]; j.v[l] = lmj[s][l]; }
    }
  };
  static VI leg() { VI r; r.v[0] = 0; r.v[1] = 1; return r; }
  static VI opq(VI x) { return x; }
  static void fence() {}
  static int zs() { return 0; }
  static VD sel(VM m, VD a, VD b) { VD r; for (int l = 0; l < 2; l++) r.v[l] = m.v[l] ? a.v[l] : b.v[l]; return r; }
  static VI seli(VM m, VI a, VI b) { VI r; for (int l = 0; l < 2; l++) r.v[l] = m.v[l] ? a.v[l] : b.v[l]; return r; }
  static VD swap(VD x) { VD r; r.v[0] = x.v[1]; r.v[1] = x.v[0]; return r; }
  template <int W> static VD pair_bcast(VD x) { VD r; r.v[0] = r.v[1] = x.v[W]; return r; }
  static VM swapm(VM x) { VM r; r.v[0] = x.v[1]; r.v[1] = x.v[0]; return r; }
  static bool any(VM m) { return m.v[0] || m.v[1]; }
  static VD ldc(const double* t, VI i) { VD r; r.v[0] = t[i.v[0]]; r.v[1] = t[i.v[1]]; return r; }
  static VD ldg(const double* t, VI i) { return ldc(t, i); }
  static VI toI(VM m) { VI r; r.v[0] = m.v[0]; r.v[1] = m.v[1]; return r; }
  static VD toD(VI i) { VD r; r.v[0] = i.v[0]; r.v[1] = i.v[1]; return r; }
  static VI toint(VD x) { VI r; r.v[0] = (int)x.v[0]; r.v[1] = (int)x.v[1]; return r; }
  static void sincos(VD x, VD& s, VD& c) { ops(2); for (int l = 0; l < 2; l++) { s.v[l] = std::sin(x.v[l]); c.v[l] = std::cos(x.v[l]); } }
  static VD sqrt(VD x) { ops(); VD r; for (int l = 0; l < 2; l++) r.v[l] = std::sqrt(x.v[l]); return r; }
  static VD rcp(VD x) { ops(); VD r; for (int l = 0; l < 2; l++) r.v[l] = 1.0 / x.v[l]; return r; }
  static VD fma(VD a, VD b, VD c) { ops(2); VD r; for (int l = 0; l < 2; l++) r.v[l] = std::fma(a.v[l], b.v[l], c.v[l]); return r; }
  static VD fabs(VD x) { VD r; for (int l = 0; l < 2; l++) r.v[l] = std::fabs(x.v[l]); return r; }
  static VD fmax(VD a, VD b) { VD r; for (int l = 0; l < 2; l++) r.v[l] = std::fmax(a.v[l], b.v[l]); return r; }
  static VD exp(VD x) { ops(); VD r; for (int l = 0; l < 2; l++) r.v[l] = std::exp(x.v[l]); return r; }
  static VD fmod(VD a, double b) { VD r; for (int l = 0; l < 2; l++) r.v[l] = std::fmod(a.v[l], b); return r; }
  static VD copysign(VD a, VD b) { VD r; for (int l = 0; l < 2; l++) r.v[l] = std::copysign(a.v[l], b.v[l]); return r; }
  static VD pld(P p, VI off) { VD r; for (int l = 0; l < 2; l++) r.v[l] = p.p[l][off.v[l]]; return r; }
  static void pst(P p, VI off, VD v, VM m) { for (int l = 0; l < 2; l++) if (m.v[l]) p.p[l][off.v[l]] = v.v[l]; }
  static void pst8(P8 p, VM v, VM m) { for (int l = 0; l < 2; l++) if (m.v[l]) *p.p[l] = (uint8_t)v.v[l]; }
};

typedef cassie::leg::Core<HostB> HCore;

// Height-field instantiation (the counterpart of DevBHF in cassie_kernels_leg.hip): one more per-lane slot per contact pair and the
// terrain test.  `hf_sphere` restates terrain_sphere of cassie_kernels.hip line by line (that one is __device__ code).
struct HostBHF : HostB {
  struct Lds : HostB::Lds {
    double nrm[3][2];
    void st_nrm(VI slot, VD nx, VM m) { for (int l = 0; l < 2; l++) if (m.v[l]) nrm[slot.v[l]][l] = nx.v[l]; }
    VD ld_nrm(int s) const { VD r; r.v[0] = nrm[s][0]; r.v[1] = nrm[s][1]; return r; }
  };
  static void hf_sphere(const cassie::Terrain& t, VD wx, VD wy, VD wz, VD radius, VD& dist, VD& nx, VD& nz) {
    for (int l = 0; l < 2; l++) {
      const int nr = t.nrow, nc = t.ncol;
      const double dx = 2.0 * t.sx / (nc - 1), dy = 2.0 * t.sy / (nr - 1);
      const double gx = (wx.v[l] + t.sx) / dx, gy = (wy.v[l] + t.sy) / dy;
      nx.v[l] = 0.0; nz.v[l] = 1.0; dist.v[l] = wz.v[l] - radius.v[l];
      if (!(gx >= 0.0 && gx <= (double)(nc - 1) && gy >= 0.0 && gy <= (double)(nr - 1))) continue;
      int ci = (int)gx, ri = (int)gy;
      ci = ci > nc - 2 ? nc - 2 : ci;
      ri = ri > nr - 2 ? nr - 2 : ri;
      const double fx = gx - ci, fy = gy - ri;
      const double* h0 = t.h + (size_t)ri * nc + ci;
      const double z00 = h0[0], z10 = h0[1], z01 = h0[nc], z11 = h0[nc + 1];
      double a, b;
      if (fy <= fx) { a = (z10 - z00) / dx; b = (z11 - z10) / dy; }
      else { a = (z11 - z01) / dx; b = (z01 - z00) / dy; }
      const double zs = z00 + a * (fx * dx) + b * (fy * dy);
      nz.v[l] = 1.0 / std::sqrt(1.0 + a * a); nx.v[l] = -a * nz.v[l];
      dist.v[l] = (wz.v[l] - zs) * nz.v[l] - radius.v[l];
    }
  }
};
typedef cassie::leg::Core<HostBHF> HCoreHF;

HostB::P both(double* p) { HostB::P r; r.p[0] = r.p[1] = p; return r; }

}  // namespace

extern "C" {

// One Env.step (or n_sub bare substeps when obs == null) of n environments on host arrays laid out like the device ones.
// pending[e] = substeps NOT done because the environment left the 8-rows-per-leg capacity (its state is untouched from there on).
int leg_host_step(double* state, const double* actions, int n, int adim, int mode, int n_sub, int flags, int env_kind, int auto_reset,
                  const double* traj_qpos, double traj_tmax, int traj_n, double* obs, double* reward, uint8_t* done, double* terminal_obs,
                  int* pending, int* nonfinite) {
  cassie::leg::EnvCfg cfg;
  cfg.n_sub = n_sub; cfg.flags = flags; cfg.env_kind = env_kind; cfg.auto_reset = auto_reset; cfg.adim = adim;
  cfg.want_obs = obs != nullptr; cfg.traj_qpos = traj_qpos; cfg.traj_tmax = traj_tmax; cfg.traj_n = traj_n;
  double dummy[32] = {0};
  uint8_t dummy8 = 0;
  for (int e = 0; e < n; e++) {
    HostB::Lds lds;
    std::memset(&lds, 0, sizeof lds);
    HCore::Io io;
    io.rec = both(state + (size_t)e * cassie::ENV_STRIDE);
    io.has_act = actions != nullptr;
    io.act = both(actions ? const_cast<double*>(actions) + (size_t)e * adim : dummy);
    io.obs = both(obs ? obs + (size_t)e * 26 : dummy);
    io.has_tobs = terminal_obs != nullptr;
    io.tobs = both(terminal_obs ? terminal_obs + (size_t)e * 26 : dummy);
    io.rew = both(reward ? reward + e : dummy);
    io.done.p[0] = io.done.p[1] = done ? done + e : &dummy8;
    VM valid; valid.v[0] = valid.v[1] = true;
    HCore::Out o;
    if (mode == 0) HCore::env_step<0>(cfg, lds, io, valid, o);
    else if (mode == 1) HCore::env_step<1>(cfg, lds, io, valid, o);
    else HCore::env_step<2>(cfg, lds, io, valid, o);
    if (pending) pending[e] = o.pend.v[0];
    if (nonfinite && o.bad.v[0]) (*nonfinite)++;
  }
  return 0;
}

// The same on a height field (heights[nrow][ncol] metres over [-sx, sx] x [-sy, sy]): cassie_leg_core.h with HF = true.
int leg_host_step_hf(double* state, const double* actions, int n, int adim, int mode, int n_sub, int flags, int env_kind, int auto_reset,
                     const double* heights, int nrow, int ncol, double sx, double sy, double* obs, double* reward, uint8_t* done, int* pending, int* nonfinite) {
  cassie::leg::EnvCfg cfg;
  cfg.n_sub = n_sub; cfg.flags = flags; cfg.env_kind = env_kind; cfg.auto_reset = auto_reset; cfg.adim = adim;
  cfg.want_obs = obs != nullptr; cfg.traj_qpos = nullptr; cfg.traj_tmax = 0.0; cfg.traj_n = 0;
  cassie::Terrain hf; hf.h = heights; hf.nrow = nrow; hf.ncol = ncol; hf.sx = sx; hf.sy = sy;
  double dummy[32] = {0};
  uint8_t dummy8 = 0;
  for (int e = 0; e < n; e++) {
    HostBHF::Lds lds;
    std::memset(&lds, 0, sizeof lds);
    for (int s = 0; s < 3; s++) lds.nrm[s][0] = lds.nrm[s][1] = std::nan("");   // an unused slot holds anything (r03: a NaN there leaked once)
    HCoreHF::Io io;
    io.rec = both(state + (size_t)e * cassie::ENV_STRIDE);
    io.has_act = actions != nullptr;
    io.act = both(actions ? const_cast<double*>(actions) + (size_t)e * adim : dummy);
    io.obs = both(obs ? obs + (size_t)e * 26 : dummy);
    io.has_tobs = false;
    io.tobs = both(dummy);
    io.rew = both(reward ? reward + e : dummy);
    io.done.p[0] = io.done.p[1] = done ? done + e : &dummy8;
    VM valid; valid.v[0] = valid.v[1] = true;
    HCoreHF::Out o;
    if (mode == 0) HCoreHF::env_step<0, true>(cfg, lds, io, valid, o, &hf);
    else if (mode == 1) HCoreHF::env_step<1, true>(cfg, lds, io, valid, o, &hf);
    else HCoreHF::env_step<2, true>(cfg, lds, io, valid, o, &hf);
    if (pending) pending[e] = o.pend.v[0];
    if (nonfinite && o.bad.v[0]) (*nonfinite)++;
  }
  return 0;
}

// arithmetic operations counted since the last call (see the note at g_ops)
double leg_host_ops(void) { double r = g_ops; g_ops = 0.0; return r; }

}  // extern "C"
