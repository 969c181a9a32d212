// leg3d_host.cpp -- CHECKER INFRASTRUCTURE (lives under oracle/): compiles cassierl_amd/csrc/cassie3d_leg_core.h (the lane-per-leg
// Cassie3d physics of the HIP kernel cassie3d_leg.hip) for the CPU with the lane emulation of lane_types.h, so that the CPU test-suite
// can check the kernel's SOURCE against the Cassie3d oracle before anything runs on a GPU (tests/test_leg3d_host.py).
// Only tests/ build and load this; the product (cassierl_amd/) has no CPU path.
#define __device__
#define __constant__
#define __forceinline__ inline
#define LEG_FN inline
#define LEG3_SUBSTEP_FN inline
#ifdef LEG3_STATS
static long long g_leg3_stat[4];
#define LEG3_STAT(k) (++g_leg3_stat[k])
#endif
#include "../../cassierl_amd/csrc/cassie3d_leg_core.h"

#include "lane_types.h"

namespace {

struct HostB3 : HostOps {
  struct K { const double* p[NL]; };
  static K kbase(VI leg) { K k; LANES k.p[l] = &::c3_legk[0][0] + leg.v[l] * LK3_N; return k; }
  static VD kld(K k, int idx) { VD r; LANES r.v[l] = k.p[l][idx]; return r; }
  struct Lds {   // per-lane slots [slot][lane]
    double a[cassie3d::leg::NSLOT3][NL];
    VD ld(int s) const { VD r; LANES r.v[l] = a[s][l]; return r; }
    void st(int s, VD v, VM m) { LANES if (m.v[l]) a[s][l] = v.v[l]; }
    VD ldv(VI s) const { VD r; LANES { const i64 k = s.v[l]; r.v[l] = a[k >= 0 && k < cassie3d::leg::NSLOT3 ? k : 0][l]; } return r; }
    void stv(VI s, VD v, VM m) { LANES if (m.v[l]) a[s.v[l]][l] = v.v[l]; }
  };
};
typedef cassie3d::leg::Core3<HostB3> HCore3;

}  // namespace

extern "C" {

// n_sub torque-mode substeps (integrate != 0) or one mj_forward (integrate == 0) of n environments on host arrays laid out like the
// device ones: state [n][80], torques [n][10] or null (the record's ctrl).  pending[e] = substeps NOT done because the environment
// left the row capacity; niter / nrows: PGS sweeps summed over the substeps done / constraint rows of the last one.
int leg3d_host_step(double* state, const double* torques, int n, int n_sub, int integrate, int* pending, int* niter, int* nrows) {
  constexpr int EPG = NL / 2;
  const int groups = (n + EPG - 1) / EPG;
  for (int g = 0; g < groups; g++) {
    const int e0 = g * EPG;
    static thread_local HostB3::Lds lds;
    for (auto& r : lds.a) LANES r[l] = 7.0;   // (the kernel clears its slots once; a finite non-zero value here shows a read of a slot that was never written as a wrong result)
    HCore3::Io io;
    LANES { const int e = e0 + (l >> 1); io.rec.p[l] = state + (size_t)(e < n ? e : e0) * cassie3d::ENV3_STRIDE; }
    io.has_act = torques != nullptr;
    LANES { const int e = e0 + (l >> 1); io.act.p[l] = torques ? const_cast<double*>(torques) + (size_t)(e < n ? e : e0) * cassie3d::NU : io.rec.p[l]; }
    VM valid; LANES valid.v[l] = e0 + (l >> 1) < n ? -1 : 0;
    HCore3::Out o;
    HCore3::env_step(lds, io, valid, n_sub, integrate != 0, o);
    for (int k = 0; k < EPG && e0 + k < n; k++) {
      if (pending) pending[e0 + k] = (int)o.pend.v[2 * k];
      if (niter) niter[e0 + k] = (int)o.niter.v[2 * k];
      if (nrows) nrows[e0 + k] = (int)o.nrows.v[2 * k];
    }
  }
  return 0;
}

int leg3d_host_lanes(void) { return NL; }
// arithmetic operations counted since the last call (lane_types.h: g_ops; 0 in the LEG_HOST_FAST builds)
double leg3d_host_ops(void) {
#ifdef LEG_HOST_FAST
  return 0.0;
#else
  double r = g_ops; g_ops = 0.0; return r;
#endif
}
#ifdef LEG3_STATS
void leg3d_host_stats(long long* out4) { for (int i = 0; i < 4; i++) { out4[i] = g_leg3_stat[i]; g_leg3_stat[i] = 0; } }
#endif

}  // extern "C"
