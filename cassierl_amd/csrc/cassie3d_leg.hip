// cassie3d_leg.hip -- gfx950 backend and kernel of the lane-per-leg Cassie3d physics (cassie3d_leg_core.h): one lane per leg, 32
// environments per wavefront, one wavefront per workgroup.
//
//   lane 2e     left leg of environment e (+ the pelvis collision sphere, + the base fields of the record on write-back)
//   lane 2e + 1 right leg
// The two lanes of an environment exchange values with DPP quad_perm [1,0,3,2]; nothing else crosses lanes and there is no barrier.
// LDS is per-lane indexed storage, [slot][lane] (bank = lane: conflict-free whatever the per-lane slot index): the constraint rows of
// the matrix-free Gauss-Seidel solver (joint limits and contacts; the three connect rows of a leg stay in registers) and, before
// them, the per-link sums of the two kinematics passes -- 80 KB per wavefront, i.e. TWO wavefronts per CU (512 resident wavefronts =
// 16 384 environments: configs[4] in one round; r04's first build kept every row in LDS, 160 KB, one wavefront per CU: 10.7 ms per
// step against 12.8 for the r03 kernels).  Registers: sized for one wavefront per SIMD (512 VGPR + AGPR): the factorisation
// (L^-1 28, Y 42, G 21 doubles), the lane state (40), the connect rows (75) and the running c, a~.
// Reference call sites: as cassie3d_kernels.hip (the MJCF model/cassie3d_stiff.xml; Cassie2d::Step semantics, Cassie2d.cpp:86-94).
#ifndef CASSIE3D_LEG_HIP_
#define CASSIE3D_LEG_HIP_
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LEG_FN __device__ __forceinline__
#define LEG3_SUBSTEP_FN __device__ __noinline__
#ifdef CASSIE3D_PHASE_TIMING   // profiling builds (tools/phase_profile_3d.py): cycles per phase of the substep, summed over wavefronts and launches
namespace cassie3d { namespace leg { __device__ unsigned long long k5c_phase[16]; } }
#define LEG3_PHASE_BEGIN unsigned long long t_last_ = __builtin_readcyclecounter(), sw_[3] = {0ull, 0ull, 0ull};
#define LEG3_MARK(k) { const unsigned long long n_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&::cassie3d::leg::k5c_phase[k], n_ - t_last_); t_last_ = n_; }
#define LEG3_SW_MARK(k) { const unsigned long long n_ = __builtin_readcyclecounter(); sw_[k] += n_ - t_last_; t_last_ = n_; }
#define LEG3_SW_FLUSH { if (threadIdx.x == 0) { atomicAdd(&::cassie3d::leg::k5c_phase[8], sw_[0]); atomicAdd(&::cassie3d::leg::k5c_phase[9], sw_[1]); atomicAdd(&::cassie3d::leg::k5c_phase[10], sw_[2]); } }
#endif
#include "cassie3d_leg_core.h"

namespace cassie3d {
namespace leg {

// LANES: active lanes per wavefront (64 = 32 environments, 32 = 16 environments with the upper half of the wavefront idle).  The kernel is
// bound by the latency of its own chains at one wavefront per SIMD, and LDS -- NSLOT3 slots per ACTIVE lane -- is what limits the
// wavefronts per CU: with 32 active lanes a wavefront needs 40 KB, so FOUR share a CU (one per SIMD) instead of two, each walks the
// row counts of the worst of 16 environments instead of 32, and configs[4]'s 16 384 environments are 1024 wavefronts = one round of
// the whole chip (r04: 7.55 -> ... ms per step).  The idle lanes cost nothing: a wave instruction takes the same time either way.
template <int LANES>
struct DevB3 {
  struct OwnerScope { LEG_FN OwnerScope(bool) {} };   // (operation counting exists in the CPU emulation only)
  typedef double D;
  typedef int I;
  typedef bool M;
  typedef double* P;
  typedef const double* K;   // the lane's row of c3_legk
  static LEG_FN K kbase(int leg) { return &c3_legk[0][0] + leg * LK3_N; }
  static LEG_FN double kld(K k, int idx) { return k[idx]; }
  struct Lds {   // (an idle lane stores nothing -- not even the unconditional stores of the kinematics passes -- and reads the slots of lane - LANES)
    double a[NSLOT3][LANES];
    static LEG_FN bool mine() { return LANES == 64 || (int)threadIdx.x < LANES; }
    LEG_FN double ld(int s) const { return a[s][threadIdx.x & (LANES - 1)]; }
    LEG_FN void st(int s, double v, bool m) { if (m && mine()) a[s][threadIdx.x & (LANES - 1)] = v; }
    LEG_FN double ldv(int s) const { return a[s][threadIdx.x & (LANES - 1)]; }
    LEG_FN void stv(int s, double v, bool m) { if (m && mine()) a[s][threadIdx.x & (LANES - 1)] = v; }
  };
  static LEG_FN int leg() { return (int)threadIdx.x & 1; }
  static LEG_FN void fence() { __builtin_amdgcn_sched_barrier(0); }
  static LEG_FN int opq(int x) { asm volatile("" : "+v"(x)); return x; }
  static LEG_FN double sel(bool m, double a, double b) { return m ? a : b; }
  static LEG_FN int seli(bool m, int a, int b) { return m ? a : b; }
  static LEG_FN int swapi(int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, false); }
  static LEG_FN double swap(double x) { return __hiloint2double(swapi(__double2hiint(x)), swapi(__double2loint(x))); }
  static LEG_FN bool swapm(bool m) { return swapi((int)m) != 0; }
  template <int W> static LEG_FN double pair_bcast(double x) {
    constexpr int CTRL = W == 0 ? 0xA0 : 0xF5;
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, false));
  }
  static LEG_FN bool any(bool m) { return __ballot(m) != 0ull; }
  static LEG_FN double ldc(const double* t, int i) { return t[i]; }
  static LEG_FN int toI(bool m) { return (int)m; }
  static LEG_FN void sincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }
  static LEG_FN double sqrt(double x) { return ::sqrt(x); }
  // 1/d: hardware seed (about 2^-26 relative) + ONE Newton step = a few ulp; the second step that K1c takes (it is compared bit for
  // bit with its host build) costs 2 dependent FMAs on each of the ~14 reciprocals of a contact step: 6.63 -> 6.32 ms per Env.step
  static LEG_FN double rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
#ifndef LEG3_RCP_STEPS
#define LEG3_RCP_STEPS 1
#endif
#pragma unroll
    for (int i = 0; i < LEG3_RCP_STEPS; i++) { const double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r); }
    return r;
  }
  static LEG_FN double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
  static LEG_FN double fabs(double x) { return ::fabs(x); }
  static LEG_FN double fmax(double a, double b) { return ::fmax(a, b); }
  static LEG_FN double pld(const double* p, int off) { return p[off]; }
  static LEG_FN void pst(double* p, int off, double v, bool m) { if (m) p[off] = v; }
};

// pending_out[env] = substeps this kernel did NOT do because the environment needed more constraint rows than a lane's LDS slots hold
// (0 normally); env_step3d_kernel finishes those (cassie_cabi.hip).
template <int LANES>
__global__ void __launch_bounds__(64, 1) env_step3d_leg_kernel(Params3 p) {
  typedef Core3<DevB3<LANES>> DCore3;
  __shared__ typename DevB3<LANES>::Lds lds;
  const int lane = threadIdx.x;
  const int env = blockIdx.x * (LANES / 2) + (lane >> 1);
  const bool exists = lane < LANES && env < p.n_envs;
  const bool first = p.gone == nullptr || p.seg_first != 0;
  const bool valid = exists && (first || p.gone[env] == 0);
  const size_t e = valid ? (size_t)env : 0;
  typename DCore3::Io io;
  io.rec = p.state + e * ENV3_STRIDE;
  io.has_act = p.actions != nullptr;
  io.act = io.has_act ? const_cast<double*>(p.actions) + e * NU : io.rec;
  if (lane < LANES) for (int s = 0; s < NSLOT3; s++) lds.a[s][lane] = 0.0;   // a lane only ever reads back what it wrote -- or this
  typename DCore3::Out o;
  DCore3::env_step(lds, io, valid, p.n_sub, p.integrate != 0, o);
  if (exists && (lane & 1) == 0) {
    const int pend = (valid && o.pend > 0) ? o.pend + (p.gone ? p.seg_later : 0) : 0;
    if (valid) {
      double* st = p.state + e * ENV3_STRIDE;
      st[E3_NITER] = (double)o.niter + (first ? 0.0 : st[E3_NITER]);
      if (pend == 0) st[E3_NEFC] = (double)o.nrows;
      if (pend > 0 && p.stats) atomicAdd(p.stats + S3_LEG_HANDOVER_SUBSTEPS, (unsigned long long)pend);
    }
    if (p.pending_out) p.pending_out[env] = pend;
    if (p.gone) { if (first) p.gone[env] = pend > 0; else if (pend > 0) p.gone[env] = 1; }
  }
}

}  // namespace leg
}  // namespace cassie3d
#endif
