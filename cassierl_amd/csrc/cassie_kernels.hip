// cassie_kernels.hip -- MI355X (gfx950) kernels for the batched Cassie2d hot path.
//
// One wavefront (64 lanes) per environment instance, one wavefront per workgroup.
// The generalised-coordinate state and every per-step intermediate live in LDS / VGPRs
// for all n_sub physics substeps of one Env.step(); HBM is touched once on entry
// (state + action, one coalesced 640-byte read) and once on exit (state, obs, reward, done).
//
// What one substep computes (reference call sites):
//   Cassie2d::StepPd / Step        src/Cassie2d/Cassie2d.cpp:86-117   (controller law + mj_step)
//   mj_step (MuJoCo Pro 1.50, PGS / elliptic / Euler; model/cassie2d_stiff.xml:5)
//   GetOperationalSpaceState       src/Cassie2d/Cassie2d.cpp:218-237
//   Cassie2dEnv.step / reset       rllab/envs/cassie2d.py:78-225, cassie_stand2d.py:72-137
//
// Formulation: sagittal-plane reduction (tables in cassie2d_planar.h, derived offline from
// model/cassie2d_stiff.xml).  Lane roles inside the wave:
//   link lanes  0..10   planar FK of the 11 links (angle, origin, COM, inertial force)
//   dof  lanes  d = lane&15 < 13 in 16-lane groups 0 and 1: group 0 carries M, group 1 carries
//               M + h*diag(damping); both are inverted by the same in-register Gauss-Jordan
//               instruction stream (DPP row broadcasts of the pivot row)
//   row  lanes  0..45   one FIXED constraint slot per lane (4 connect rows, 8 joint-limit rows,
//               17 contact (normal,tangent) pairs in MuJoCo's contact order); lane i keeps row i of
//               A = J M^-1 J' + R in registers and its own residual, so a PGS row update is
//               "owner lane computes delta -> v_readlane -> one FMA on every lane".
#ifndef CASSIE_KERNELS_HIP_
#define CASSIE_KERNELS_HIP_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cassie2d_planar.h"
#include "cassie_vec_layout.h"
#include "cassie_terrain.h"

namespace cassie {

constexpr int NV = CP_NV, NL = CP_NLINK, NU = CP_NU, NSLOT = CP_NSLOT;
constexpr int SLOT_LIM = 4, SLOT_CON = 12;
constexpr double MINVAL = 1e-15;
constexpr double H = CP_TIMESTEP;

template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}

__device__ __forceinline__ double rdlane(double x, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
// broadcast lane K of every 16-lane row to the whole row (DPP row_newbcast, gfx90a+)
// (mov_dpp: no "old" operand to materialise -- every lane of a row_newbcast / quad_perm / row_ror has a valid source)
// One v_mov_b64_dpp: row_newbcast is the DPP control the 64-bit data path supports.  Must be executed by the source lane too
// (never inside a lane-dependent branch or the lazily evaluated arm of a ?:): a source lane switched off by EXEC yields 0.
template <int K> __device__ __forceinline__ double row_bcast(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + K, 0xF, 0xF, true);
}
// exchange with lane^1 (DPP quad_perm [1,0,3,2])
__device__ __forceinline__ double swap1(double x) {
  int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, false);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// all-reduce inside every 16-lane row (DPP row rotations)
template <int CTRLCODE> __device__ __forceinline__ double dpp_mov(double x) {
  int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRLCODE, 0xF, 0xF, false);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRLCODE, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// The adds must not be contracted with a multiply in the caller: fma(a_i, b_i, x_j) on lane i and fma(a_j, b_j, x_i) on lane j
// round differently, and then the lanes of a row disagree in the last bit -- fatal for per-row decisions taken from the sum
// (Jacobi rotation signs, active-set picks).  With contraction off every lane of the row gets the bit-identical value.
__device__ __forceinline__ double row_sum(double x) {
#pragma clang fp contract(off)
  x += dpp_mov<0x128>(x); x += dpp_mov<0x124>(x); x += dpp_mov<0x122>(x); x += dpp_mov<0x121>(x);
  return x;
}
__device__ __forceinline__ double row_min(double x) {
  x = fmin(x, dpp_mov<0x128>(x)); x = fmin(x, dpp_mov<0x124>(x)); x = fmin(x, dpp_mov<0x122>(x)); x = fmin(x, dpp_mov<0x121>(x));
  return x;
}
__device__ __forceinline__ double row_max(double x) {
  x = fmax(x, dpp_mov<0x128>(x)); x = fmax(x, dpp_mov<0x124>(x)); x = fmax(x, dpp_mov<0x122>(x)); x = fmax(x, dpp_mov<0x121>(x));
  return x;
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
  return x;
}
__device__ __forceinline__ void lds_sync() { __syncthreads(); }  // single-wave workgroup: fence only

struct Smem {
  double q[16], v[16], ws[16], kq[16], kv[16], ctrl[8];
  double lc[12], ls[12], lw[12], lox[12], loz[12], lvx[12], lvz[12], lax[12], laz[12];
  double lcx[12], lcz[12], lfx[12], lfz[12];
  double s1x[16], s1z[16], s2[16];
  double tau[16], qs[16], g[16];
  double minv[NV * NV + 7], mhinv[NV * NV + 7];
  double rowJ[NSLOT + 2][8];
  double rowf[NSLOT + 2];
  double site[2][6][4];
  // index of Hinv(r, c) for the controllers (cassie_ctrl.hip): dense rows, the copy written by row min(r, c)
  __host__ __device__ static constexpr int hidx(int r, int c) { return r <= c ? r * NV + c : c * NV + r; }
};

// Loop-invariant per-lane ROLE data.  Only small integers stay in registers for the whole kernel; the double-precision
// per-lane constants are re-read from the (L2-resident) constant tables inside the phase that uses them, through an
// index the optimiser cannot see through, so that they are never hoisted out of the substep loop and spilled
// (round-1 PMC profile: 3.5 GB of scratch traffic per launch came from exactly that).
struct LaneConst {
  int ancmask;                                          // link role
  int d, grp, dvalid, dlink, submask, rel, act, kL, kR; // dof role
  int kind, leg, comp, rdof, link1, link2, pm1, pm2;    // row role
};
struct DofConst { double sigma, damping, armature, gear, clo, chi; };
struct RowConst { double d1x, d1z, d2x, d2z, radius, invw, lim_lo, lim_hi, solref0, solref1, simp0, simp1, simp2; };

__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// Phase timing (profiling builds only): PHASE_MARK(acc, k) adds the shader cycles since the previous mark to acc[k].
#ifdef CASSIE_PHASE_TIMING
struct PhaseClock {
  unsigned long long t, acc[16];
  __device__ __forceinline__ void start() { for (int i = 0; i < 16; i++) acc[i] = 0; t = __builtin_readcyclecounter(); }
  __device__ __forceinline__ void mark(int k) { unsigned long long n = __builtin_readcyclecounter(); acc[k] += n - t; t = n; }
  __device__ __forceinline__ void flush(unsigned long long* dst, int lane) { if (dst && lane == 0) for (int i = 0; i < 16; i++) atomicAdd(dst + i, acc[i]); }
};
#define PHASE_MARK(pc, k) (pc).mark(k)
#else
struct PhaseClock {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void flush(unsigned long long*, int) {}
};
#define PHASE_MARK(pc, k) ((void)0)
#endif

__device__ __forceinline__ void load_lane_const(LaneConst& c, int lane) {
  int l = lane < NL ? lane : 0;
  c.ancmask = lane < NL ? cp_link_ancmask[l] : 0;
  c.d = lane & 15; c.grp = lane >> 4; c.dvalid = (c.d < NV) && (lane < 32);
  int d = c.d < NV ? c.d : 0;
  c.dlink = cp_dof_link[d]; c.submask = cp_dof_submask[d]; c.rel = cp_dof_rel[d]; c.act = cp_dof_act[d];
  c.kL = d < 8 ? d : -1;                     // compact index of this dof in a left-leg row
  c.kR = d < 3 ? d : (d >= 8 ? d - 5 : -1);  // ... in a right-leg row
  int s = lane < NSLOT ? lane : 0;
  c.kind = lane < NSLOT ? cp_slot_kind[s] : -1; c.leg = cp_slot_leg[s]; c.comp = cp_slot_comp[s]; c.rdof = cp_slot_dof[s];
  c.link1 = cp_slot_link1[s]; c.link2 = cp_slot_link2[s];
  c.pm1 = cp_link_pathmask8[c.link1]; c.pm2 = cp_link_pathmask8[c.link2];
}
__device__ __forceinline__ void load_dof_const(DofConst& k, const LaneConst& c) {
  int d = opaque(c.d < NV ? c.d : 0);
  k.sigma = cp_dof_sigma[d]; k.damping = cp_dof_damping[d]; k.armature = cp_dof_armature[d];
  int a = opaque(c.act >= 0 ? c.act : 0);
  k.gear = c.act >= 0 ? cp_act_gear[a] : 0.0; k.clo = cp_act_ctrlrange[a][0]; k.chi = cp_act_ctrlrange[a][1];
}
__device__ __forceinline__ void load_row_const(RowConst& r, const LaneConst& c, int lane) {
  int s = opaque(lane < NSLOT ? lane : 0);
  r.d1x = cp_slot_d1[0][s][0]; r.d1z = cp_slot_d1[0][s][1]; r.d2x = cp_slot_d2[0][s][0]; r.d2z = cp_slot_d2[0][s][1];
  r.radius = cp_slot_radius[s]; r.invw = cp_slot_invweight[s];
  int rd = opaque(c.rdof >= 0 ? c.rdof : 0);
  r.lim_lo = cp_jnt_range[rd][0]; r.lim_hi = cp_jnt_range[rd][1];
  const double* sr = c.kind == 0 ? cp_eq_solref[s >> 1] : (c.kind == 1 ? cp_limit_solref : cp_contact_solref);
  const double* si = c.kind == 0 ? cp_eq_solimp[s >> 1] : (c.kind == 1 ? cp_limit_solimp : cp_contact_solimp);
  r.solref0 = sr[0]; r.solref1 = sr[1]; r.simp0 = si[0]; r.simp1 = si[1]; r.simp2 = si[2];
}

__device__ __forceinline__ double impedance(double d0, double d1, double width, double x) {
  if (d0 == d1 || width <= MINVAL) return 0.5 * (d0 + d1);
  x = fabs(x / width);
  if (x >= 1.0) return d1;
  if (x <= 0.0) return d0;
  double y = x <= 0.5 ? 2.0 * x * x : 1.0 - 2.0 * (1.0 - x) * (1.0 - x);
  return d0 + y * (d1 - d0);
}

// height-field terrain (N4): terrain_sphere() -- the sphere against the closest feature of the terrain's sagittal section -- lives in
// cassie_terrain.h (shared with the other kernel families and the CPU instantiation of the two-lanes-per-environment core)

// ---------------------------------------------------------------- planar forward kinematics on the link lanes
// Reads sm.q/sm.v-like arrays (qsrc, vsrc), writes link arrays.  SEM selects the model semantics table.
template <int SEM, class SM>
__device__ __forceinline__ void planar_fk(SM& sm, const double* qsrc, const double* vsrc, const LaneConst& c, int lane) {
  double th = 0.0, w = 0.0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
    int dk = cp_link_dof[k];
    double sg = cp_link_sigma[k];
    double dq = sg * (qsrc[dk] - cp_qpos0[dk]), dv = sg * vsrc[dk];
    if ((c.ancmask >> k) & 1) { th += dq; w += dv; }
  }
  double sn, cs;
  sincos(th, &sn, &cs);
  if (lane < NL) { sm.lc[lane] = cs; sm.ls[lane] = sn; sm.lw[lane] = w; }
  lds_sync();
  // chain sums: origin (relative to the pelvis origin), origin velocity (relative part), velocity-product accel
  double ox = 0, oz = 0, vx = 0, vz = 0, ax = 0, az = CP_GRAVITY;
#pragma unroll
  for (int k = 1; k < NL; k++) {
    int p = cp_link_parent[k];
    double pc = sm.lc[p], ps = sm.ls[p], pw = sm.lw[p];
    double fx = cp_link_off[SEM][k][0], fz = cp_link_off[SEM][k][1];
    double tx = pc * fx + ps * fz, tz = -ps * fx + pc * fz;
    if ((c.ancmask >> k) & 1) {
      ox += tx; oz += tz;
      vx += pw * tz; vz -= pw * tx;          // w y^ x t
      ax -= pw * pw * tx; az -= pw * pw * tz;  // centripetal
    }
  }
  if (lane < NL) {
    const int lo_ = opaque(lane);
    double cx0 = cp_link_com[SEM][lo_][0], cz0 = cp_link_com[SEM][lo_][1];
    const double lmass = cp_link_mass[lo_];
    double rx = cs * cx0 + sn * cz0, rz = -sn * cx0 + cs * cz0;
    sm.lox[lane] = ox; sm.loz[lane] = oz; sm.lvx[lane] = vx; sm.lvz[lane] = vz; sm.lax[lane] = ax; sm.laz[lane] = az;
    sm.lcx[lane] = ox + rx; sm.lcz[lane] = oz + rz;
    sm.lfx[lane] = lmass * (ax - w * w * rx); sm.lfz[lane] = lmass * (az - w * w * rz);
  }
  lds_sync();
}

// point on a link (relative to the pelvis origin)
template <class SM>
__device__ __forceinline__ void link_point(const SM& sm, int link, double dx, double dz, double& px, double& pz) {
  double cs = sm.lc[link], sn = sm.ls[link];
  px = sm.lox[link] + cs * dx + sn * dz;
  pz = sm.loz[link] - sn * dx + cs * dz;
}

// compact Jacobian row (component comp: 0 = x, 1 = z) of a point on `link`; legbase = 1 (left) or 6 (right)
template <class SM>
__device__ __forceinline__ void jac_compact(const SM& sm, int pm, int legbase, int comp, double px, double pz, double sgn, double* J) {
  // base slides
  J[0] += sgn * (comp == 0 ? 1.0 : 0.0);
  J[1] += sgn * (comp == 1 ? 1.0 : 0.0);
  // pitch hinge: sigma = +1, anchor = pelvis origin (0,0)
  J[2] += sgn * (comp == 0 ? pz : -px);
#pragma unroll
  for (int k = 0; k < 5; k++) {
    int l = legbase + k;
    double rx = px - sm.lox[l], rz = pz - sm.loz[l];
    double val = cp_link_sigma[1 + k] * (comp == 0 ? rz : -rx);
    if ((pm >> (3 + k)) & 1) J[3 + k] += sgn * val;
  }
}

// ---------------------------------------------------------------- coalesced HBM <-> LDS state movement
// s1 lane map: 0 kv[12] | 1..13 qstate | 14..19 ctrl | 20 time | 21 PGS iterations
__device__ __forceinline__ double load_state(const double* st, Smem& sm, int lane) {
  double s0 = st[lane];
  double s1 = lane < ENV_STRIDE - 64 ? st[64 + lane] : 0.0;
  if (lane < 13) sm.q[lane] = s0;
  else if (lane < 26) sm.v[lane - 13] = s0;
  else if (lane < 39) sm.ws[lane - 26] = s0;
  else if (lane < 52) sm.kq[lane - 39] = s0;
  else sm.kv[lane - 52] = s0;
  if (lane == 0) sm.kv[12] = s1;
  if (lane >= 14 && lane < 20) sm.ctrl[lane - 14] = s1;
  return s1;
}
__device__ __forceinline__ void store_state(double* st, const Smem& sm, int lane, double qstate_l, double time, int niter) {
  double w0;
  if (lane < 13) w0 = sm.q[lane];
  else if (lane < 26) w0 = sm.v[lane - 13];
  else if (lane < 39) w0 = sm.ws[lane - 26];
  else if (lane < 52) w0 = sm.kq[lane - 39];
  else w0 = sm.kv[lane - 52];
  st[lane] = w0;
  if (lane < ENV_STRIDE - 64) {
    double w1 = 0.0;
    if (lane == 0) w1 = sm.kv[12];
    else if (lane < 14) w1 = qstate_l;
    else if (lane < 20) w1 = sm.ctrl[lane - 14];
    else if (lane == 20) w1 = time;
    else if (lane == 21) w1 = (double)niter;
    st[64 + lane] = w1;
  }
}

// ---------------------------------------------------------------- mass matrix rows and bias on the dof lanes
// Needs planar_fk<SEM> results in LDS.  Lane (d = lane&15) gets row d of M (+ armature, + h*damping when add_hb) and
// bias_d = C(q,v) + g(q) (RNE with qacc = 0).  Uses sm.s1x/s1z/s2 as exchange buffers.
template <int SEM, class SM>
__device__ __forceinline__ void mass_rows(SM& sm, const LaneConst& c, const DofConst& dc, int lane, double (&Mr)[NV], double& bias, bool add_hb) {
  double msub = 0, s1x = 0, s1z = 0, s2 = 0;
  bias = 0;
  const double odx = sm.lox[c.dlink], odz = sm.loz[c.dlink];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    double rx = sm.lcx[l] - odx, rz = sm.lcz[l] - odz;
    double m = cp_link_mass[l], fx = sm.lfx[l], fz = sm.lfz[l];
    if ((c.submask >> l) & 1) {
      msub += m; s1x += m * rx; s1z += m * rz; s2 += m * (rx * rx + rz * rz) + cp_link_inertia[SEM][l];
      bias += c.d == 0 ? fx : (c.d == 1 ? fz : dc.sigma * (fx * rz - fz * rx));
    }
  }
  lds_sync();  // previous readers of s1x/s1z/s2 are done
  if (c.dvalid && c.grp == 0) { sm.s1x[c.d] = s1x; sm.s1z[c.d] = s1z; sm.s2[c.d] = s2; }
  lds_sync();
  static_for<0, NV>([&](auto cc) {
    constexpr int C = decltype(cc)::value;
    double val;
    if constexpr (C < 2) {
      val = c.d < 2 ? (c.d == C ? msub : 0.0) : dc.sigma * (C == 0 ? s1z : -s1x);
    } else {
      constexpr int LC = C == 2 ? 0 : C - 2;
      constexpr double SC = C == 2 ? 1.0 : -1.0;
      double ocx = sm.lox[LC], ocz = sm.loz[LC];
      double c1x = sm.s1x[C], c1z = sm.s1z[C], c2 = sm.s2[C];
      int code = (c.rel >> (2 * C)) & 3;
      double deep_row = s2 + (odx - ocx) * s1x + (odz - ocz) * s1z;
      double deep_col = c2 + (ocx - odx) * c1x + (ocz - odz) * c1z;
      double hh = code == 1 ? deep_row : (code == 2 ? deep_col : 0.0);
      double slide = SC * (c.d == 0 ? c1z : -c1x);
      val = c.d < 2 ? slide : dc.sigma * SC * hh;
    }
    if (C == c.d) val += dc.armature + (add_hb ? H * dc.damping : 0.0);
    Mr[C] = val;
  });
  if (!c.dvalid) { static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; Mr[C] = (C == (lane & 15)) ? 1.0 : 0.0; }); }
}

// 1/d to ~1 ulp: hardware seed (v_rcp_f64) + two Newton steps: 5 dependent instructions instead of the ~14 of an
// IEEE-correct division.  Used on latency-critical paths of the 4-envs-per-wave kernel.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// In-register Gauss-Jordan inverse of an N x N SPD matrix (N <= 16) held one row per lane inside every 16-lane row of the
// wave; the four rows of the wave are independent problems executed by the same instruction stream (DPP row_newbcast).
// Lanes with (lane&15) >= N must hold a zero row.
template <int N, bool FAST = false>
__device__ __forceinline__ void gauss_jordan_rows(double (&Mr)[N], int lane) {
  static_for<0, N>([&](auto kk) {
    constexpr int K = decltype(kk)::value;
    double piv = row_bcast<K>(Mr[K]);
    double inv = FAST ? fast_rcp(piv) : 1.0 / piv;
    bool isk = (lane & 15) == K;
    double t = isk ? 1.0 - inv : Mr[K] * inv;  // pivot row: pk - (1 - inv) pk = pk * inv
    static_for<0, N>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      if constexpr (C != K) {
        double pk = row_bcast<K>(Mr[C]);
        Mr[C] = __builtin_fma(-t, pk, Mr[C]);
      }
    });
    Mr[K] = isk ? inv : -t;
  });
}

// The same inverse for the 13 x 13 mass matrix, using its structure: the two legs (dofs 3..7 and 8..12) couple only through the
// base dofs 0..2.  Pivoting on the leg dofs FIRST keeps the cross-leg blocks exactly zero (no fill-in before a base pivot), so
// a leg pivot has nothing to do in the five columns of the other leg: 10 pivots x 5 columns x (2 DPP moves + 1 FMA) less per
// inversion.  (Any pivot order is stable for an SPD matrix; the order only changes the rounding, at 1e-16.)
// The column updates Mr[C] -= t * (lane K of Mr[C]) as v_fmac_f64_dpp: the broadcast rides on the multiply-add's DPP source
// operand (the one FP64 arithmetic instruction with a DPP encoding on gfx950, row_newbcast the one control it takes) -- one
// instruction per column instead of a DPP move plus an FMA.  The compiler's DPP combiner does not form it, hence the assembly;
// it cannot see the "VALU write -> DPP read: 2 wait states" hazard across an assembly statement either, so each block opens
// and closes with its own s_nop 1 (the columns inside a block are independent of each other).
#define GJ_FMAC(i) "v_fmac_f64_dpp %" #i ", %" #i ", -%[t] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
template <int K> __device__ __forceinline__ void gj_elim7(double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double t) {
  asm("s_nop 1\n\t" GJ_FMAC(0) GJ_FMAC(1) GJ_FMAC(2) GJ_FMAC(3) GJ_FMAC(4) GJ_FMAC(5) GJ_FMAC(6) "s_nop 1"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : [t] "v"(t), [k] "n"(K));
}
template <int K> __device__ __forceinline__ void gj_elim12(double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double& a7,
                                                           double& a8, double& a9, double& a10, double& a11, double t) {
  asm("s_nop 1\n\t" GJ_FMAC(0) GJ_FMAC(1) GJ_FMAC(2) GJ_FMAC(3) GJ_FMAC(4) GJ_FMAC(5) GJ_FMAC(6) GJ_FMAC(7) GJ_FMAC(8) GJ_FMAC(9) GJ_FMAC(10) GJ_FMAC(11) "s_nop 1"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11) : [t] "v"(t), [k] "n"(K));
}
#undef GJ_FMAC
// j-th column a pivot on dof K touches: every column but its own and, for a leg pivot, the five of the other leg
constexpr int gj_col(int K, int j) {
  int n = 0;
  for (int C = 0; C < NV; C++) {
    const bool other_leg = (K >= 3 && K <= 7 && C >= 8) || (K >= 8 && C >= 3 && C <= 7);
    if (C != K && !other_leg) { if (n == j) return C; n++; }
  }
  return -1;
}
template <bool FAST = false>
__device__ __forceinline__ void gauss_jordan_rows_legs(double (&Mr)[NV], int lane) {
  constexpr int ORDER[NV] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0, 1, 2};
  static_for<0, NV>([&](auto ii) {
    constexpr int K = ORDER[decltype(ii)::value];
    double piv = row_bcast<K>(Mr[K]);
    double inv = FAST ? fast_rcp(piv) : 1.0 / piv;
    bool isk = (lane & 15) == K;
    double t = isk ? 1.0 - inv : Mr[K] * inv;  // pivot row: pk - (1 - inv) pk = pk * inv
    if constexpr (K >= 3)
      gj_elim7<K>(Mr[gj_col(K, 0)], Mr[gj_col(K, 1)], Mr[gj_col(K, 2)], Mr[gj_col(K, 3)], Mr[gj_col(K, 4)], Mr[gj_col(K, 5)], Mr[gj_col(K, 6)], t);
    else
      gj_elim12<K>(Mr[gj_col(K, 0)], Mr[gj_col(K, 1)], Mr[gj_col(K, 2)], Mr[gj_col(K, 3)], Mr[gj_col(K, 4)], Mr[gj_col(K, 5)], Mr[gj_col(K, 6)],
                   Mr[gj_col(K, 7)], Mr[gj_col(K, 8)], Mr[gj_col(K, 9)], Mr[gj_col(K, 10)], Mr[gj_col(K, 11)], t);
    Mr[K] = isk ? inv : -t;
  });
}

struct StepOut { int niter; unsigned long long active; };

// A[i][S] = X_i . J_S for a wave-uniform slot S (compact Jacobian of S read from LDS at a uniform address)
__device__ __forceinline__ double arow_entry(const Smem& sm, const double (&X)[NV], int S) {
  const double* js = sm.rowJ[S];
  const int legS = S < SLOT_LIM ? (S >> 1) : (S < SLOT_CON ? ((S - SLOT_LIM) >> 2) : (((S - SLOT_CON) >> 1) <= 8 ? 0 : 1));
  double a = X[0] * js[0] + X[1] * js[1] + X[2] * js[2];
  if (legS == 0) a += X[3] * js[3] + X[4] * js[4] + X[5] * js[5] + X[6] * js[6] + X[7] * js[7];
  else a += X[8] * js[3] + X[9] * js[4] + X[10] * js[5] + X[11] * js[6] + X[12] * js[7];
  return a;
}

// ---------------------------------------------------------------- one mj_forward (+ optional Euler integration)
// On entry sm.q/v/ws hold the state; ctrl is this dof lane's actuator command (pre-clamp).
template <bool INTEGRATE, int MAXACT, bool HF = false>
__device__ __forceinline__ void substep(Smem& sm, const LaneConst& c, int lane, double ctrl, StepOut& out, double* dbg, double* ovf,
                                        const Terrain* terrain = nullptr) {
  // ---- kinematics
  planar_fk<0>(sm, sm.q, sm.v, c, lane);
  // ---- mass-matrix row on every dof lane (group 1 adds h*damping on the diagonal), then both inverses at once
  double tau, qs;
  {
  DofConst dc;
  load_dof_const(dc, c);
  double Mr[NV];
  double bias;
  mass_rows<0>(sm, c, dc, lane, Mr, bias, c.grp == 1);
  if (dbg && c.dvalid && c.grp == 0) {
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; dbg[DBG_M + c.d * NV + C] = Mr[C]; });
    dbg[DBG_BIAS + c.d] = bias;
  }
  gauss_jordan_rows_legs<false>(Mr, lane);
  if (c.dvalid) {
    double* dst = c.grp == 0 ? sm.minv : sm.mhinv;
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; dst[c.d * NV + C] = Mr[C]; });
  }
  // ---- smooth forces and unconstrained acceleration
  double v_d = sm.v[c.d < NV ? c.d : 0];
  double u = ctrl < dc.clo ? dc.clo : (ctrl > dc.chi ? dc.chi : ctrl);
  tau = -dc.damping * v_d - bias + dc.gear * u;
  qs = 0.0;
  static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; qs += Mr[C] * row_bcast<C>(tau); });
  }  // Mr and the dof constants die here; the inverse rows are re-read from LDS after the solve
  if (c.dvalid && c.grp == 0) { sm.tau[c.d] = tau; sm.qs[c.d] = qs; }
  lds_sync();
  if (dbg && c.dvalid && c.grp == 0) dbg[DBG_QS + c.d] = qs;
  // ---- constraint rows on the row lanes (fixed slots)
  double b, jar, R;
  bool active = false;
  unsigned long long amask;
  double X[NV];
  {
  RowConst rc;
  load_row_const(rc, c, lane);
  double J[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double pos = 0.0;
  const int legbase = c.leg == 0 ? 1 : 6;
  const int vbase = c.leg == 0 ? 3 : 8;
  const double basez = sm.q[1] - cp_qpos0[1] + cp_link_off[0][0][1];
  if (c.kind == 0) {
    double p1x, p1z, p2x, p2z;
    link_point(sm, c.link1, rc.d1x, rc.d1z, p1x, p1z);
    link_point(sm, c.link2, rc.d2x, rc.d2z, p2x, p2z);
    jac_compact(sm, c.pm1, legbase, c.comp, p1x, p1z, 1.0, J);
    jac_compact(sm, c.pm2, legbase, c.comp, p2x, p2z, -1.0, J);
    pos = c.comp == 0 ? p1x - p2x : p1z - p2z;
    active = true;
  } else if (c.kind == 1) {
    double qd = sm.q[c.rdof];
    double dlo = qd - rc.lim_lo, dhi = rc.lim_hi - qd;
    int k = 3 + c.rdof - vbase;
    double sgn = 0.0;
    if (dlo < 0) { active = true; pos = dlo; sgn = 1.0; }
    else if (dhi < 0) { active = true; pos = dhi; sgn = -1.0; }
    static_for<3, 8>([&](auto kk) { constexpr int K = decltype(kk)::value; J[K] = (k == K) ? sgn : 0.0; });
  } else if (c.kind >= 2) {
    double cx, cz;
    link_point(sm, c.link1, rc.d1x, rc.d1z, cx, cz);
    if constexpr (HF) {
      // terrain: contact frame from the cell under the sphere; normal row along (nx, nz), tangent row along (nz, -nx)
      const double basex = sm.q[0] - cp_qpos0[0] + cp_link_off[0][0][0];
      const int sph = opaque(lane >= SLOT_CON && lane < NSLOT ? (lane - SLOT_CON) >> 1 : 0);
      double dist, nx, nz;
      terrain_sphere(*terrain, basex + cx, cp_sph_y[sph], basez + cz, rc.radius, dist, nx, nz);
      if (dist < 0) {
        active = true;
        const double back = rc.radius + 0.5 * dist;
        const double px = cx - nx * back, pz = cz - nz * back;
        const double dirx = c.kind == 2 ? nx : nz, dirz = c.kind == 2 ? nz : -nx;
        jac_compact(sm, c.pm1, legbase, 0, px, pz, dirx, J);
        jac_compact(sm, c.pm1, legbase, 1, px, pz, dirz, J);
        pos = dist;
      }
    } else {
      double dist = basez + cz - rc.radius;
      if (dist < 0) {
        active = true;
        double pz = 0.5 * dist - basez;
        jac_compact(sm, c.pm1, legbase, c.comp, cx, pz, 1.0, J);
        pos = dist;  // both rows of the pair keep the normal distance (shared regulariser); the tangent row's own pos is 0
      }
    }
  }
  amask = __ballot(active);
  // row velocity, impedance, regulariser, reference acceleration
  double vel = J[0] * sm.v[0] + J[1] * sm.v[1] + J[2] * sm.v[2];
  double bq = J[0] * sm.qs[0] + J[1] * sm.qs[1] + J[2] * sm.qs[2];
  double jw = J[0] * sm.ws[0] + J[1] * sm.ws[1] + J[2] * sm.ws[2];
  static_for<0, 5>([&](auto kk) {
    constexpr int K = decltype(kk)::value;
    vel += J[3 + K] * sm.v[vbase + K]; bq += J[3 + K] * sm.qs[vbase + K]; jw += J[3 + K] * sm.ws[vbase + K];
  });
  double tc = rc.solref0 < 2.0 * H ? 2.0 * H : rc.solref0;
  double kk_ = 1.0 / (rc.simp1 * rc.simp1 * tc * tc * rc.solref1 * rc.solref1), bb_ = 2.0 / (rc.simp1 * tc);
  double imp = impedance(rc.simp0, rc.simp1, rc.simp2, pos);
  R = (1.0 - imp) / imp * rc.invw;
  R = R > MINVAL ? R : MINVAL;
  double own_pos = c.kind == 3 ? 0.0 : pos;
  double imp_own = c.kind == 3 ? impedance(rc.simp0, rc.simp1, rc.simp2, 0.0) : imp;
  double aref = -bb_ * vel - kk_ * imp_own * own_pos;
  b = active ? bq - aref : 0.0;
  jar = jw - aref;
  if (dbg && lane < NSLOT) { dbg[DBG_R + lane] = active ? R : 0.0; dbg[DBG_AREF + lane] = active ? aref : 0.0; }
  if (lane < NSLOT) { static_for<0, 8>([&](auto kk) { constexpr int K = decltype(kk)::value; sm.rowJ[lane][K] = J[K]; }); }
  // ---- X = M^-1 J' (13 values per row lane)
  {
    const double* mi = sm.minv;
    static_for<0, NV>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      double sx = mi[C * NV + 0] * J[0] + mi[C * NV + 1] * J[1] + mi[C * NV + 2] * J[2];
      static_for<0, 5>([&](auto kk) { constexpr int K = decltype(kk)::value; sx += mi[C * NV + vbase + K] * J[3 + K]; });
      X[C] = sx;
    });
  }
  }  // row constants and J die here
  out.active = amask;
  lds_sync();
  // ---- A row in COMPACT column order: Ac[k] = A[this row][k-th active slot].  Only active columns occupy registers
  // (MAXACT of them); the slot of column k is recovered by a scalar bit scan of the activity mask, so the unrolled
  // loops keep static register indices while slots, legs and kinds are wave-uniform run-time values.
  const int nact = __popcll(amask);
  double Ac[MAXACT];
  double Adiag = 1.0, Ant = 0.0;
  {
    unsigned long long m = amask;
    static_for<0, MAXACT>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      double a = 0.0;
      if (K < nact) {
        const int S = __ffsll((long long)m) - 1;
        m &= m - 1;
        a = arow_entry(sm, X, S);
        if (lane == S) { a += R; Adiag = a; }
        if (S >= SLOT_CON && (S & 1) && lane == S - 1) Ant = a;  // A[n][t] on the normal lane of a contact pair
      }
      Ac[K] = a;
    });
    // columns beyond MAXACT (only when more than MAXACT rows are active): global workspace, slow path
    for (int k = MAXACT; k < nact; k++) {
      const int S = __ffsll((long long)m) - 1;
      m &= m - 1;
      double a = arow_entry(sm, X, S);
      if (lane == S) { a += R; Adiag = a; }
      if (S >= SLOT_CON && (S & 1) && lane == S - 1) Ant = a;
      ovf[(size_t)(k - MAXACT) * 64 + lane] = a;
    }
  }
  double Ainv = 1.0 / Adiag;
  double Apart = swap1(Adiag);  // for a normal lane: A_tt of its tangent partner
  // ---- warm start: forces from qacc_warmstart (mj_constraintUpdate), kept only if the dual cost beats zero
  const double mu = CP_CONTACT_MU;
  double f = 0.0;
  {
    double D = 1.0 / R;
    double pj = swap1(jar);
    if (active) {
      if (c.kind == 0) f = -D * jar;
      else if (c.kind == 1) f = jar < 0 ? -D * jar : 0.0;
      else {
        double jn = c.kind == 2 ? jar : pj, jt = c.kind == 2 ? pj : jar;
        double N = jn * mu, U1 = jt * mu, T = fabs(U1);
        double fn, ft;
        if (N >= mu * T || (T <= 0 && N >= 0)) { fn = 0; ft = 0; }
        else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { fn = -D * jn; ft = -D * jt; }
        else {
          double Dm = D / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
          fn = -Dm * NmT * mu;
          ft = -fn / T * U1 * mu;
        }
        f = c.kind == 2 ? fn : ft;
      }
    }
  }
  double res = 0.0;  // (A f)_i
  {
    unsigned long long m = amask;
    static_for<0, MAXACT>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      if (K < nact) { const int S = __ffsll((long long)m) - 1; m &= m - 1; res += Ac[K] * rdlane(f, S); }
    });
    for (int k = MAXACT; k < nact; k++) { const int S = __ffsll((long long)m) - 1; m &= m - 1; res += ovf[(size_t)(k - MAXACT) * 64 + lane] * rdlane(f, S); }
  }
  {
    double cost = wave_sum(active ? f * (0.5 * res + b) : 0.0);
    if (cost > 0) { f = 0.0; res = 0.0; }
  }
  res += b;
  if (dbg && lane < NSLOT) { dbg[DBG_F0 + lane] = f; dbg[DBG_B + lane] = b; dbg[DBG_ADIAG + lane] = active ? Adiag : 0.0; }
  // ---- PGS (mj_solPGS, elliptic cones): residual per lane, delta broadcast by v_readlane
  // Divisions by loop-invariant quantities are replaced by multiplications with reciprocals computed once per
  // substep (1/A_nn, 1/A_tt); the only per-iteration division left is the ray step's 1/denom.
  const double scale = 1.0 / (CP_MEANINERTIA * NV);
  const double AttInv = 1.0 / Apart;
  double improvement;
  // one single-row update (connect or joint limit) of slot S whose A column is aS
  auto update_single = [&](int S, double aS) {
    double cand = f - res * Ainv;
    if (S >= SLOT_LIM) cand = cand < 0 ? 0.0 : cand;
    double d = cand - f;
    double chg = d * (0.5 * d * Adiag + res);
    if (chg > 1e-10) { d = 0.0; chg = 0.0; }
    double Dd = rdlane(d, S);
    improvement -= rdlane(chg, S);
    if (lane == S) f += d;
    res += aS * Dd;
  };
  // one elliptic contact pair: normal row on the even lane S (A column aN), tangent on S+1 (A column aT).
  // This kernel runs as the last hand-over tier, a few wavefronts on an idle machine: an environment costs the LENGTH OF THE
  // DEPENDENCY CHAIN of its 50 sweeps, nothing else (r03 trace: ~900 cycles per pair, 180 us per substep).  So the step is
  // written for a short chain: no branches (selects on every lane; only the owner lane's values are used), and the ray
  // update's 1/denom -- a function of the pair's own forces alone -- is computed at the end of the pair's PREVIOUS update,
  // where it overlaps the cost evaluation and the broadcast, instead of sitting between the residual and the new force.
  auto ray_rden = [&](double fn, double ft) {
    const double denom = fn * (Adiag * fn + Ant * ft) + ft * (Ant * fn + Apart * ft);
    return denom >= MINVAL ? fast_rcp(denom) : 0.0;   // 0: the ray update leaves the point where it is (mj_solPGS: denom < mjMINVAL)
  };
  double rden = ray_rden(f, swap1(f));
  auto update_pair = [&](int S, double aN, double aT) {
    // lane S gathers the pair locally: its own (res,f) are the normal's; the partner's via DPP
    const double rt = swap1(res), ot = swap1(f);
    const double rn = res, on = f;
    const double Ann = Adiag, Att = Apart;
    // normal: ray update through the current point when the normal force is positive, else the plain 1-D update
    double x = -(on * rn + ot * rt) * rden;
    x = x < -1.0 ? -1.0 : x;  // keep the normal force non-negative: fn + x fn >= 0
    double fn1 = on - rn * Ainv;
    fn1 = fn1 < 0 ? 0.0 : fn1;
    const bool ray = on >= MINVAL;
    const double fn = ray ? on + x * on : fn1;
    double ft = ray ? ot + x * ot : 0.0;
    // friction: QCQP on one dimension (mu = CP_CONTACT_MU): unconstrained minimiser unless it leaves the cone
    {
      const double bc = (rt - Att * ot) + Ant * (fn - on);
      const double x0 = -bc * AttInv;
      const double v1 = x0 * (1.0 / mu);
      const double val = v1 * v1 - fn * fn;
      const bool out_of_cone = val >= 1e-10 && val * Att * (mu * mu) >= 2e-10 * (v1 * v1);
      const double ftq = out_of_cone ? (x0 > 0 ? mu : -mu) * fn : x0;
      ft = fn >= MINVAL ? ftq : ft;
    }
    double dn = fn - on, dt = ft - ot;
    double chg = 0.5 * (Ann * dn * dn + 2.0 * Ant * dn * dt + Att * dt * dt) + dn * rn + dt * rt;
    const double rden_new = ray_rden(fn, ft);   // off the chain: overlaps the cost test and the broadcast below (kept on every
                                                 // lane by the select further down: inside an EXEC-masked block it would not overlap)
    const bool reject = chg > 1e-10;
    if (reject) { dn = 0.0; dt = 0.0; chg = 0.0; }
    double Dn = rdlane(dn, S), Dt = rdlane(dt, S);
    improvement -= rdlane(chg, S);
    rden = (lane == S && !reject) ? rden_new : rden;
    f += lane == S ? dn : (lane == S + 1 ? Dt : 0.0);
    res += aN * Dn + aT * Dt;
  };
  int niter = 0;
#ifndef K1_ITERS   // timing builds only (tools/k1_latency.py): fewer sweeps, to separate the solver from the rest of a substep
#define K1_ITERS CP_ITERATIONS
#endif
  for (int iter = 0; iter < K1_ITERS; iter++) {
    improvement = 0.0;  // wave-uniform: every row's cost change is read back from its owner lane
    unsigned long long m = amask;
    static_for<0, MAXACT>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      if (K < nact) {
        const int S = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (S < SLOT_CON) update_single(S, Ac[K]);
        else if (!(S & 1)) {
          if constexpr (K + 1 < MAXACT) update_pair(S, Ac[K], Ac[K + 1]);
          else update_pair(S, Ac[K], ovf[lane]);  // pair straddles the register/workspace boundary
        }
      }
    });
    for (int k = MAXACT; k < nact; k++) {
      const int S = __ffsll((long long)m) - 1;
      m &= m - 1;
      const double* col = ovf + (size_t)(k - MAXACT) * 64 + lane;
      if (S < SLOT_CON) update_single(S, col[0]);
      else if (!(S & 1)) update_pair(S, col[0], col[64]);
    }
    niter = iter + 1;
    if (improvement * scale < CP_TOLERANCE) break;
  }
  out.niter = niter;
  if (lane < NSLOT) sm.rowf[lane] = active ? f : 0.0;
  lds_sync();
  if (dbg && lane < NSLOT) dbg[DBG_F + lane] = active ? f : 0.0;
  // ---- total generalised force g = tau + J' f on the dof lanes, then both accelerations
  double g = tau;
  static_for<0, NSLOT>([&](auto ss) {
    constexpr int S = decltype(ss)::value;
    constexpr int LEG = S < 4 ? S / 2 : (S < 12 ? (S - 4) / 4 : ((S - 12) / 2 <= 8 ? 0 : 1));
    if ((amask >> S) & 1) {
      int k = LEG == 0 ? c.kL : c.kR;
      double js = sm.rowJ[S][k < 0 ? 0 : k];
      g += (k < 0 ? 0.0 : js) * sm.rowf[S];
    }
  });
  double acc_d = 0.0;  // group 0: qacc = M^-1 g ; group 1: (M + hB)^-1 g
  {
    const double* mrow = (c.grp == 1 ? sm.mhinv : sm.minv) + (c.d < NV ? c.d : 0) * NV;
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; acc_d += mrow[C] * row_bcast<C>(g); });
  }
  if (dbg && c.dvalid) dbg[(c.grp == 0 ? DBG_QACC : DBG_QACCH) + c.d] = acc_d;
  lds_sync();
  if (c.dvalid && c.grp == 0) sm.ws[c.d] = acc_d;  // qacc_warmstart <- qacc
  if (INTEGRATE && c.dvalid && c.grp == 1) {
    double vn = sm.v[c.d] + H * acc_d;  // mj_Euler with implicit joint damping
    sm.v[c.d] = vn;
    sm.q[c.d] = sm.q[c.d] + H * vn;
  }
  lds_sync();
}

// ---------------------------------------------------------------- operational-space state (Cassie2d.cpp:218-237)
// kinematics of the LAST setState (kq,kv; quirk Q1/Q2) with the RBDL-semantics tables; pitch from the current state.
template <class SM>
__device__ __forceinline__ void opstate18(SM& sm, const LaneConst& c, int lane, bool fix_stale, double* s18 /*LDS*/) {
  const double* qk = fix_stale ? sm.q : sm.kq;
  const double* vk = fix_stale ? sm.v : sm.kv;
  planar_fk<1>(sm, qk, vk, c, lane);
  if (lane < 5) {
    int sid = lane + 1;
    int l = cp_site_link[sid];
    double dx = cp_site_d[1][sid][0], dz = cp_site_d[1][sid][1];
    double cs = sm.lc[l], sn = sm.ls[l], w = sm.lw[l];
    double rx = cs * dx + sn * dz, rz = -sn * dx + cs * dz;
    double bx = qk[0] - cp_qpos0[0] + cp_link_off[1][0][0], bz = qk[1] - cp_qpos0[1] + cp_link_off[1][0][1];
    sm.site[0][lane][0] = bx + sm.lox[l] + rx;
    sm.site[0][lane][1] = bz + sm.loz[l] + rz;
    sm.site[0][lane][2] = vk[0] + sm.lvx[l] + w * rz;
    sm.site[0][lane][3] = vk[1] + sm.lvz[l] - w * rx;
  }
  lds_sync();
  if (lane < 2) {
    int i = lane;
    s18[i] = sm.site[0][0][i];
    s18[3 + i] = sm.site[0][0][2 + i];
    s18[6 + i] = (sm.site[0][1][i] + sm.site[0][2][i]) / 2.0;
    s18[9 + i] = (sm.site[0][1][2 + i] + sm.site[0][2][2 + i]) / 2.0;
    s18[12 + i] = (sm.site[0][3][i] + sm.site[0][4][i]) / 2.0;
    s18[15 + i] = (sm.site[0][3][2 + i] + sm.site[0][4][2 + i]) / 2.0;
  }
  if (lane == 2) {
    s18[2] = sm.q[2]; s18[5] = sm.v[2];
    s18[8] = 0; s18[11] = 0; s18[14] = 0; s18[17] = 0;  // never written by the reference (Q4)
  }
  lds_sync();
}

// ---------------------------------------------------------------- Env.step outputs (reward, termination, gait lookup)
__device__ __forceinline__ bool in_range(double x) { return fabs(x) <= FINITE_BOUND; }  // false for NaN and +-inf

// Wave-per-environment layout.  sp = obs[lane] (lane < 17: pos-invariant op-space state) on entry; the walk env fills lanes
// 17..25 with the reference-gait joints.  qstate_l: lane 1 + j holds self.qstate[j].  act: this env's action vector or null.
//   walk  rllab/envs/cassie2d.py:158-225         stand  rllab/envs/cassie_stand2d.py:114-137
template <class SM>
__device__ __forceinline__ void env_outputs_wave(const VecParams& p, const SM& sm, const double* s18, int lane, const double* act, int adim,
                                                 double qstate_l, double time, double& sp, double& reward, int& done) {
  if (p.env_kind == 0) {
    // reference-gait lookup (cassie2d_trajectory.py:16-19)
    double tmax = p.traj_tmax;
    int idx = (int)(fmod(time, tmax) / tmax * p.traj_n);
    const double* rq = p.traj_qpos + (size_t)idx * NV;
    if (lane >= 17 && lane < 26) {
      int k = lane - 17;  // columns 0,1,2,3,4,6,8,9,11
      int col = k < 5 ? k : (k == 5 ? 6 : (k == 6 ? 8 : (k == 7 ? 9 : 11)));
      sp = rq[col];
    }
    // reward (cassie2d.py:197-218); qstate is the reset pose unless FLAG_FIX_STALE_QSTATE (quirk Q3)
    const bool fixq = (p.flags & FLAG_FIX_STALE_QSTATE) != 0;
    auto qst = [&](int j) { return fixq ? sm.q[j] : rdlane(qstate_l, 1 + j); };
    double j = qst(3) + qst(4) + qst(6);
    j += qst(8) + qst(9) + qst(11);
    double sum = 0.0;
    for (int i = 20; i < 26; i++) sum += rdlane(sp, i);
    j -= sum; j = exp(-(j * j));
    double pp = s18[0] + s18[1];
    pp -= rdlane(sp, 17) + rdlane(sp, 18); pp = exp(-(pp * pp));
    double oo = s18[2];
    oo -= rdlane(sp, 19); oo = exp(-(oo * oo));
    reward = 0.5 * j + 0.3 * pp + 0.1 * oo;
    done = (s18[1] < 0.6) || (s18[1] > 1.2) || (reward < 0.6);
  } else {
    double a2 = 0.0;
    if (act) for (int i = 0; i < adim; i++) { double a = act[i]; a2 += a * a; }
    double z = s18[1];
    double m = (rdlane(sp, 5) + rdlane(sp, 11)) / 2.0;
    reward = 0.0;
    reward -= 2 * (0.9 - z) * (0.9 - z);
    reward -= 2 * m * m;
    reward += 1;
    reward -= 0.001 * a2;
    done = z < 0.5;
  }
}

// Row layout (16 lanes per environment): lane l holds obs[l] in obs_a and obs[16 + l] in obs_b (lanes 0..9); qv on lane d is
// the joint position the reward's joint term reads (self.qstate[d], or qpos[d] with FLAG_FIX_STALE_QSTATE).
__device__ __forceinline__ void env_outputs_row(const VecParams& p, int l, const double* act, int adim, double time, double bodyx, double qv,
                                                double obs_a, double& obs_b, double& reward, int& done) {
  const double z = row_bcast<0>(obs_a), pitch = row_bcast<1>(obs_a);
  if (p.env_kind == 0) {
    double tmax = p.traj_tmax;
    int idx = (int)(fmod(time, tmax) / tmax * opaque(p.traj_n));   // opaque: keeps the conversion (and `col` below) out of the
    const double* rq = p.traj_qpos + (size_t)idx * NV;              // registers that live across the whole kernel
    l = opaque(l);
    if (l >= 1 && l < 10) {
      int k = l - 1;
      int col = k < 5 ? k : (k == 5 ? 6 : (k == 6 ? 8 : (k == 7 ? 9 : 11)));
      obs_b = rq[col];
    }
    double j = row_bcast<3>(qv) + row_bcast<4>(qv) + row_bcast<6>(qv);
    j += row_bcast<8>(qv) + row_bcast<9>(qv) + row_bcast<11>(qv);
    double sum = 0.0;
    sum += row_bcast<4>(obs_b); sum += row_bcast<5>(obs_b); sum += row_bcast<6>(obs_b);
    sum += row_bcast<7>(obs_b); sum += row_bcast<8>(obs_b); sum += row_bcast<9>(obs_b);
    j -= sum; j = exp(-(j * j));
    double pp = bodyx + z;
    pp -= row_bcast<1>(obs_b) + row_bcast<2>(obs_b); pp = exp(-(pp * pp));
    double oo = pitch;
    oo -= row_bcast<3>(obs_b); oo = exp(-(oo * oo));
    reward = 0.5 * j + 0.3 * pp + 0.1 * oo;
    done = (z < 0.6) || (z > 1.2) || (reward < 0.6);
  } else {
    double a2 = 0.0;
    if (act) for (int i = 0; i < adim; i++) { double a = act[i]; a2 += a * a; }
    double m = (row_bcast<5>(obs_a) + row_bcast<11>(obs_a)) / 2.0;
    reward = 0.0;
    reward -= 2 * (0.9 - z) * (0.9 - z);
    reward -= 2 * m * m;
    reward += 1;
    reward -= 0.001 * a2;
    done = z < 0.5;
  }
}

// ---------------------------------------------------------------- the fused Env.step kernel
// MODE: 0 PD (Cassie2d::StepPd), 1 torque (Cassie2d::Step), 2 motor commands from the state record (written by the controller
// kernel of cassie_ctrl_g16.hip; hand-over pass of the split StepOsc / StepJacobian path)
// WPS: waves per SIMD the register allocation is sized for.  4 (128 VGPRs, some spills) wins when the grid is only
// ~4 waves per SIMD deep (4096 envs); 3 (168 VGPRs, fewer spills) wins on deep grids (measured, profiles/r01_b_*).
template <int MODE, int WPS, int MAXACT, bool HF = false>
__global__ void __launch_bounds__(64, WPS) env_step_kernel(VecParams p) {
  __shared__ Smem sm;
  __shared__ double s18[18];
  const int lane = threadIdx.x;
  // Two launch shapes, one body.  Direct: one workgroup per environment (grid = n_envs).  Hand-over pass of the packed kernels
  // (p.pending != null): one workgroup per 64 environments (grid = n_envs / 64) that reads their `pending` counts with one
  // coalesced load and walks the non-zero ones -- nearly always none, so the pass costs ~4 us instead of the ~22 us of 65 536
  // workgroups that exit at once (r02 kernel trace), and touches neither state nor scratch.
  const int env0 = p.pending ? blockIdx.x * 64 : blockIdx.x;
  int mine = p.n_sub;
  if (p.pending) mine = (env0 + lane < p.n_envs) ? pending_count(p.pending[env0 + lane], p.pending_pick) : 0;
  unsigned long long todo = p.pending ? __ballot(mine != 0) : (env0 < p.n_envs ? 1ull : 0ull);
  if (p.pending && gridDim.y > 1) {
    // A pending environment costs its remaining substeps end to end (~0.1 ms each), so the environments of one 64-block are
    // dealt out over gridDim.y workgroups (the k-th pending one goes to workgroup k mod gridDim.y) instead of queueing behind
    // each other in one wavefront (r03 trace of fallen robots: two pending environments in a block doubled the pass).
    unsigned long long keep = 0, m = todo;
    for (int k = 0; m; k++, m &= m - 1)
      if (k % (int)gridDim.y == (int)blockIdx.y) keep |= m & (~m + 1);
    todo = keep;
  }
  while (todo) {
  const int bit = __ffsll((long long)todo) - 1;
  todo &= todo - 1;
  const int env = env0 + bit;
  const int n_sub = p.pending ? __builtin_amdgcn_readlane(mine, bit) : p.n_sub;
  if (p.pending && p.stats && lane == 0) atomicAdd(p.stats + STAT_K1_SUBSTEPS, (unsigned long long)n_sub);
  if (p.pending && p.deep_hint && lane == 0) *p.deep_hint = p.serial;   // robots are down: the host runs the lower tiers side by side for a while
  double* st = p.state + (size_t)env * ENV_STRIDE;
  LaneConst c;
  load_lane_const(c, lane);
  double s1 = load_state(st, sm, lane);
  double qstate_l = s1;  // lane 1+j holds qstate[j]
  double time = rdlane(s1, 20);
  double wset_keep = MODE == 2 ? st[ES_QPWSET] : 0.0;
  lds_sync();
  // action for this dof lane
  double act_l = 0.0;
  if (MODE != 2 && p.actions && c.act >= 0 && c.dvalid) act_l = p.actions[(size_t)env * p.adim + c.act];
  double* dbg = p.debug ? p.debug + (size_t)env * DBG_STRIDE : nullptr;
  double* ovf = p.ovf + (size_t)env * p.ovf_stride;
  StepOut so; so.niter = 0; so.active = 0;
  int niter_sum = 0;
  for (int sub = 0; sub < n_sub; sub++) {
    // DynamicModel::setState: remember the pre-step state (kinematics used by GetOperationalSpaceState)
    if (lane < 13) { sm.kq[lane] = sm.q[lane]; sm.kv[lane] = sm.v[lane]; }
    double ctrl;
    if (MODE == 0) {
      int dd = c.d < NV ? c.d : 0;
      ctrl = 10.0 * (act_l - sm.q[dd]) + 5.0 * (0.0 - sm.v[dd]);
    } else if (MODE == 2) {
      ctrl = c.act >= 0 ? sm.ctrl[c.act] : 0.0;
    } else {
      ctrl = act_l;
    }
    lds_sync();
    substep<true, MAXACT, HF>(sm, c, lane, ctrl, so, dbg, ovf, &p.hf);
    niter_sum += so.niter;
    time += 0.0005;
    if (sub == n_sub - 1 && c.dvalid && c.grp == 0 && c.act >= 0) sm.ctrl[c.act] = ctrl;  // mj_data->ctrl
  }
  lds_sync();
  // ---- observation, reward, termination (Cassie2dEnv.step) -- optional
  if (p.obs) {
    const bool fix_kin = (p.flags & FLAG_FIX_STALE_KIN) != 0;
    opstate18(sm, c, lane, fix_kin, s18);
    double sp = 0.0;  // obs[lane], lane < 26
    if (lane < 17) sp = s18[lane + 1];
    if (lane == 5 || lane == 11) sp -= s18[0];
    double reward = 0.0;
    int done = 0;
    env_outputs_wave(p, sm, s18, lane, p.actions ? p.actions + (size_t)env * p.adim : nullptr, p.adim, qstate_l, time, sp, reward, done);
    // failure guard (MuJoCo's mj_checkPos/mj_checkVel): a state outside the finite range terminates the
    // episode; the reset below then also clears everything a NaN could have reached (warm start, ctrl, setState copies)
    const bool bad = __ballot(lane < 26 && !in_range(lane < 13 ? sm.q[lane] : sm.v[lane - 13])) != 0 || !in_range(reward);
    if (bad) {
      sp = 0.0; reward = 0.0; done = 1;
      if (lane == 0 && p.stats) atomicAdd(p.stats + STAT_NONFINITE, 1ull);
      if (p.auto_reset) {
        if (lane < 13) { sm.ws[lane] = 0.0; sm.kq[lane] = cp_env_qinit[lane]; sm.kv[lane] = 0.0; }
        if (lane < NU) sm.ctrl[lane] = 0.0;
      }
    }
    if (p.terminal_obs && lane < 26) p.terminal_obs[(size_t)env * 26 + lane] = sp;
    if (done && p.auto_reset) {
      // Cassie2dEnv.reset: qinit -> Reset (mj_forward with the stale ctrl, no setState) -> op-space state from the
      // STALE kinematics (quirk Q2)
      if (lane < 13) { sm.q[lane] = cp_env_qinit[lane]; sm.v[lane] = 0.0; }
      if (lane >= 1 && lane < 14) qstate_l = cp_env_qinit[lane - 1];
      time = 0.0;
      wset_keep = 0.0;  // new episode: cold start of the OSC QP too
      lds_sync();
      substep<false, MAXACT, HF>(sm, c, lane, c.act >= 0 ? sm.ctrl[c.act] : 0.0, so, nullptr, ovf, &p.hf);
      opstate18(sm, c, lane, fix_kin, s18);
      sp = 0.0;
      if (lane < 17) sp = s18[lane + 1];
      if (lane == 5 || lane == 11) sp -= s18[0];
    }
    if (lane < 26) p.obs[(size_t)env * 26 + lane] = sp;
    if (lane == 0) { p.reward[env] = reward; p.done[env] = (uint8_t)done; }
  }
  // ---- coalesced state write-back
  store_state(st, sm, lane, qstate_l, time, niter_sum);
  if (MODE == 2 && lane == 0) st[ES_QPWSET] = wset_keep;  // store_state clears the slot; the OSC hot start survives a hand-over
  lds_sync();
  }  // while (todo)
}

// ---------------------------------------------------------------- masked reset (Cassie2dEnv.reset / Cassie2d::Reset)
template <bool HF>
__global__ void __launch_bounds__(64) env_reset_kernel(VecParams p, const uint8_t* mask, const double* qpos_in, const double* qvel_in) {
  __shared__ Smem sm;
  __shared__ double s18[18];
  const int env = blockIdx.x;
  const int lane = threadIdx.x;
  if (env >= p.n_envs) return;
  if (mask && !mask[env]) return;
  double* st = p.state + (size_t)env * ENV_STRIDE;
  LaneConst c;
  load_lane_const(c, lane);
  load_state(st, sm, lane);
  lds_sync();
  if (lane < 13) {
    sm.q[lane] = qpos_in ? qpos_in[(size_t)env * NV + lane] : cp_env_qinit[lane];
    sm.v[lane] = qvel_in ? qvel_in[(size_t)env * NV + lane] : 0.0;
  }
  lds_sync();
  double qstate_l = (lane >= 1 && lane < 14) ? sm.q[lane - 1] : 0.0;
  StepOut so;
  substep<false, 32, HF>(sm, c, lane, c.act >= 0 ? sm.ctrl[c.act] : 0.0, so, nullptr, p.ovf + (size_t)env * p.ovf_stride, &p.hf);
  if (p.obs) {
    opstate18(sm, c, lane, (p.flags & FLAG_FIX_STALE_KIN) != 0, s18);
    double sp = 0.0;
    if (lane < 17) sp = s18[lane + 1];
    if (lane == 5 || lane == 11) sp -= s18[0];
    if (lane < 26) p.obs[(size_t)env * 26 + lane] = sp;
  }
  store_state(st, sm, lane, qstate_l, 0.0, so.niter);
}

// The non-template kernels below are compiled by one translation unit only (tu_base.hip).
#ifdef CASSIE_TU_BASE
// op-space state only (GetOperationalSpaceState for every env)
__global__ void __launch_bounds__(64) env_opstate_kernel(VecParams p, double* out18) {
  __shared__ Smem sm;
  __shared__ double s18[18];
  const int env = blockIdx.x;
  const int lane = threadIdx.x;
  if (env >= p.n_envs) return;
  const double* st = p.state + (size_t)env * ENV_STRIDE;
  LaneConst c;
  load_lane_const(c, lane);
  load_state(st, sm, lane);
  lds_sync();
  opstate18(sm, c, lane, (p.flags & FLAG_FIX_STALE_KIN) != 0, s18);
  if (lane < 18) out18[(size_t)env * 18 + lane] = s18[lane];
}

// constructor state: qpos_init of Cassie2d.cpp:56-58, zero velocity, cold warm start, kin state = same
__global__ void env_init_kernel(double* state, int n_envs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_envs * ENV_STRIDE) return;
  int k = i % ENV_STRIDE;
  double v = 0.0;
  if (k < 13) v = cp_ctor_qinit[k];
  else if (k >= ES_KQ && k < ES_KQ + 13) v = cp_ctor_qinit[k - ES_KQ];
  else if (k >= ES_QSTATE && k < ES_QSTATE + 13) v = cp_env_qinit[k - ES_QSTATE];
  state[i] = v;
}

__global__ void get_state_kernel(const double* state, int n_envs, double* qpos, double* qvel) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_envs * NV) return;
  int e = i / NV, k = i % NV;
  if (qpos) qpos[i] = state[(size_t)e * ENV_STRIDE + ES_Q + k];
  if (qvel) qvel[i] = state[(size_t)e * ENV_STRIDE + ES_V + k];
}

// Rollout bookkeeping of one Env.step for the whole batch in ONE launch: returns[e] += reward[e] (the undiscounted return the
// north star gathers once per rollout batch; what rllab's sampler -- external, `rollout` in rllab/sampler/utils.py -- sums per path on
// the host) and episodes += number of done flags.  As three torch expressions (`returns += rew; dones += dn.sum()`) this was four
// launches and ~30 us per Env.step of the bench's timed loop: 3 % of a 65 536-env step (r06 kernel trace).
__global__ void __launch_bounds__(1024) accumulate_returns_kernel(const double* reward, const uint8_t* done, double* returns, unsigned long long* episodes, int n_envs) {
  __shared__ unsigned cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  bool d = false;
  if (e < n_envs) {
    if (returns) returns[e] += reward[e];
    d = episodes != nullptr && done[e] != 0;
  }
  const unsigned w = (unsigned)__popcll(__ballot(d));
  if ((threadIdx.x & 63) == 0 && w) atomicAdd(&cnt, w);
  __syncthreads();
  if (threadIdx.x == 0 && cnt) atomicAdd(episodes, (unsigned long long)cnt);
}

#endif  // CASSIE_TU_BASE

}  // namespace cassie
#endif  // CASSIE_KERNELS_HIP_
