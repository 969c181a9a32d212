// cassie3d_kernels.hip -- batched Cassie3d physics (model/cassie3d_stiff.xml) on MI355X: BASELINE.json configs[4],
// SURVEY.md section 8 row N3.  Included by tu_3d.hip after cassie_kernels.hip (shares its small helpers).
//
// The reference has no Cassie3d class (only the MJCF and vestigial hooks: xml_parser.h:321-323, RobotInterface.h:54,
// DynamicModel.cpp:250-265); what runs here is the same MuJoCo step the 2-D path restates -- mj_forward (kinematics, CRB mass
// matrix, RNE bias, plane-sphere/capsule collision, connect / joint-limit / elliptic-contact rows, PGS with warm start) and the
// implicit-damping Euler step (options of cassie3d_stiff.xml:5) -- for a floating base (3 world translations + unit
// quaternion, body-frame angular velocity) and 14 hinges: nq 21, nv 20, nu 10.  Checked in tests/test_gpu_cassie3d.py.
//
// One wavefront per environment, one wavefront per workgroup, everything between the state load and the state store in LDS
// and VGPRs.  Lane roles change by phase:
//   link lanes 0..14   kinematics level by level down the tree, then per-link inertial force/torque
//   dof  lanes 0..19   row of M (composite-inertia form), bias, Gauss-Jordan inverses of M and M + h B (v_readlane broadcasts)
//   row  lanes 0..63   ACTIVE constraint rows compacted in MuJoCo order (6 connect rows, active limits, 3 rows per active
//                      contact); lane i owns force / residual i and fills column i of A = J M^-1 J' + R in LDS
// PGS walks the rows with a wave-uniform loop: the owner's (f, residual) come by v_readlane, the update is computed
// redundantly on every lane, and each lane applies A[i][K] * delta to its own residual.  The model can produce at most
// 6 + 12 + 3 * 17 = 69 rows; beyond the 64 this kernel holds (>= 15 of the 17 collision spheres down at once, plus limits) the
// LAST contacts in MuJoCo order are left out for that substep (row cap).  That is a stated deviation for a robot lying flat
// with 14+ other contacts holding it -- never a frozen environment: E3_OVF counts such substeps per environment and the
// handle's counters report them (Cassie3dVecGetCounters); the oracle has the same cap as a switch, so parity covers it.
#ifndef CASSIE3D_KERNELS_HIP_
#define CASSIE3D_KERNELS_HIP_

#include "cassie3d_tables.h"
#include "cassie3d_layout.h"

namespace cassie3d {

using cassie::fast_rcp;
using cassie::lds_sync;
using cassie::rdlane;
using cassie::static_for;
using cassie::wave_sum;

// LDS of one environment, laid out by lifetime (r03: 17.2 -> 13.2 KB for MR = 32, so that 12 wavefronts share a CU instead of 9;
// this kernel issues one VALU instruction per 7.5 cycles per SIMD and LDS was what capped its occupancy):
//   * the by-products of the kinematics (velocities, accelerations, inertial forces, composite inertias: 605 doubles) are dead
//     once M and the bias exist -- the constraint Jacobian rows are written over them;
//   * frames, joint axes and collision spheres (402 doubles) are read while the rows are built and dead afterwards -- A is written
//     over them (and over the pad behind the by-products);
//   * A = J M^-1 J' + R is symmetric and kept as a packed upper triangle: entry (i, j) = A[tri(min) + max].  A lane reads column
//     entries (K, lane) for a wave-uniform K: `a_at` picks tri(K) + lane or tri(lane) + K (four integer instructions per read).
template <int MR>
struct Smem3 {
  static constexpr int NBY = 4 * NL * 3 + 3 * NL * 3 + NL * 6 + NV * 10;       // doubles of the by-products
  static constexpr int KPAD = MR * NV > NBY ? MR * NV - NBY : 0;               // frames start behind rowJ
  __host__ __device__ static constexpr int tri(int lo) { return lo * MR - lo * (lo + 1) / 2; }   // (lo, hi >= lo) -> tri(lo) + hi
  double q[22], v[NV], ws[NV], qs[NV];
  double minv[NV][NV];  // M^-1 (row d is lane d's scratch for M before the inversion)
  union {
    struct {
      // velocity / inertia by-products, dead once M and bias exist
      double w[NL][3], vo[NL][3], al[NL][3], ao[NL][3];
      double com[NL][3], F[NL][3], N[NL][3], Iw[NL][6];
      double comp[NV][10];
      double kpad_[KPAD ? KPAD : 1];
      // read while the constraint rows are built
      double xpos[NL][3], xmat[NL][9];
      double anchor[NV][3], axis[NV][3];
      double sphc[NSPH][3], sphdist[NSPH], spht1[NSPH][2];
    };
    struct {
      double rowJ[MR][NV];                 // constraint Jacobian rows (from the row build to the end of the substep)
      double A[MR * (MR + 1) / 2];         // packed upper triangle (from the A build to the end of the solve)
    };
  };
  // A(K, l) for a wave-uniform row K and this lane's column l (trl = tri(l))
  __device__ __forceinline__ double a_at(int K, int l, int trl) const { return A[l >= K ? tri(K) + l : trl + K]; }
};

__device__ __forceinline__ void cross3(const double* a, const double* b, double* r) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void matvec3(const double* m, const double* v, double* r) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void matmul3(const double* a, const double* b, double* r) {
  double t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] = t[i];
}
__device__ __forceinline__ double impedance3(const double* si, double pos) {
  if (si[0] == si[1] || si[2] <= MINVAL) return 0.5 * (si[0] + si[1]);
  double x = fabs(pos / si[2]);
  if (x >= 1) return si[1];
  if (x <= 0) return si[0];
  double y = x <= 0.5 ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x);
  return si[0] + y * (si[1] - si[0]);
}
__device__ __forceinline__ int nth_set(unsigned mask, int n) {
  int found = -1, cnt = 0;
  for (int i = 0; i < 20; i++)
    if ((mask >> i) & 1u) { if (cnt == n) found = i; cnt++; }
  return found;
}
__device__ __forceinline__ double rdlane_dyn(double x, int l) {  // l wave-uniform, not a compile-time constant
  int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

constexpr int IMPLICIT_DAMPING_SWEEPS3 = 12;

// In-register Gauss-Jordan inverse of the SPD matrix whose row d lives on lane d < NV (other lanes: zero rows).
__device__ __forceinline__ void gauss_jordan20(double (&Mr)[NV], int lane) {
  static_for<0, NV>([&](auto kk) {
    constexpr int K = decltype(kk)::value;
    double piv = rdlane(Mr[K], K);
    double inv = cassie::fast_rcp(piv);
    bool isk = lane == K;
    double t = isk ? 1.0 - inv : Mr[K] * inv;
    static_for<0, NV>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      if constexpr (C != K) {
        double pk = rdlane(Mr[C], K);
        Mr[C] = __builtin_fma(-t, pk, Mr[C]);
      }
    });
    Mr[K] = isk ? inv : -t;
  });
}

// ---------------------------------------------------------------- kinematics, inertial forces, bias and the rows of M
// Links on lanes 0..14 (one tree level at a time), dofs on lanes 0..19; leaves row d of M in lane d's row of sm.minv and returns the
// lane's bias force.  `lane` is the ROLE index: the lane of the wavefront in the one-environment-per-wavefront kernels, the lane
// inside its 32-lane half in the two-environments-per-wavefront kernel (cassie3d_pair.hip), where `sm` is the half's own block.
template <int MR>
__device__ __forceinline__ void kin_mass3(Smem3<MR>& sm, int lane, double& bias_out, double* dbg) {
  // ================= kinematics: links 0..14 on lanes 0..14, one tree level at a time
  const int lk = lane < NL ? lane : 0;
  const int depth = lane < NL ? c3_link_depth[lk] : 99, par = c3_link_parent[lk];
  // the hinge of link lk >= 1 is dof lk + 5 (links and hinges are numbered in the same depth-first order)
  for (int lvl = 0; lvl < 7; lvl++) {
    if (depth == lvl) {
      double pos[3], mat[9], w[3], vo[3], al[3], ao[3];
      if (lk == 0) {
        double qw = sm.q[3], qx = sm.q[4], qy = sm.q[5], qz = sm.q[6];
        double n = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
        qw /= n; qx /= n; qy /= n; qz /= n;
        mat[0] = 1 - 2 * (qy * qy + qz * qz); mat[1] = 2 * (qx * qy - qw * qz); mat[2] = 2 * (qx * qz + qw * qy);
        mat[3] = 2 * (qx * qy + qw * qz); mat[4] = 1 - 2 * (qx * qx + qz * qz); mat[5] = 2 * (qy * qz - qw * qx);
        mat[6] = 2 * (qx * qz - qw * qy); mat[7] = 2 * (qy * qz + qw * qx); mat[8] = 1 - 2 * (qx * qx + qy * qy);
        pos[0] = sm.q[0]; pos[1] = sm.q[1]; pos[2] = sm.q[2];
        double wl[3] = {sm.v[3], sm.v[4], sm.v[5]};
        matvec3(mat, wl, w);
        vo[0] = sm.v[0]; vo[1] = sm.v[1]; vo[2] = sm.v[2];
        al[0] = al[1] = al[2] = 0.0; ao[0] = ao[1] = ao[2] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          sm.anchor[k][0] = pos[0]; sm.anchor[k][1] = pos[1]; sm.anchor[k][2] = pos[2];  // (unused for slides)
          sm.axis[k][0] = k == 0; sm.axis[k][1] = k == 1; sm.axis[k][2] = k == 2;
          sm.anchor[3 + k][0] = pos[0]; sm.anchor[3 + k][1] = pos[1]; sm.anchor[3 + k][2] = pos[2];
          sm.axis[3 + k][0] = mat[k]; sm.axis[3 + k][1] = mat[3 + k]; sm.axis[3 + k][2] = mat[6 + k];
        }
      } else {
        const int dof = lk + 5;
        const double* pm = sm.xmat[par];
        double r[3], m0[9];
        matvec3(pm, c3_link_pos[lk], r);
        pos[0] = sm.xpos[par][0] + r[0]; pos[1] = sm.xpos[par][1] + r[1]; pos[2] = sm.xpos[par][2] + r[2];
        matmul3(pm, &c3_link_rot[lk][0][0], m0);
        double ax[3];
        matvec3(m0, c3_dof_axis[dof], ax);
        // Rodrigues rotation about the world axis, applied on the left (mj_kinematics)
        double ang = sm.q[c3_dof_qadr[dof]] - c3_dof_ref[dof], s = sin(ang), c = cos(ang), t1 = 1 - c;
        double R[9] = {c + ax[0] * ax[0] * t1, ax[0] * ax[1] * t1 - ax[2] * s, ax[0] * ax[2] * t1 + ax[1] * s,
                       ax[1] * ax[0] * t1 + ax[2] * s, c + ax[1] * ax[1] * t1, ax[1] * ax[2] * t1 - ax[0] * s,
                       ax[2] * ax[0] * t1 - ax[1] * s, ax[2] * ax[1] * t1 + ax[0] * s, c + ax[2] * ax[2] * t1};
        matmul3(R, m0, mat);
        const double qd = sm.v[dof];
        double wp[3] = {sm.w[par][0], sm.w[par][1], sm.w[par][2]}, alp[3] = {sm.al[par][0], sm.al[par][1], sm.al[par][2]};
        double t[3], t2[3], axqd[3] = {ax[0] * qd, ax[1] * qd, ax[2] * qd};
        cross3(wp, r, t);
        vo[0] = sm.vo[par][0] + t[0]; vo[1] = sm.vo[par][1] + t[1]; vo[2] = sm.vo[par][2] + t[2];
        cross3(wp, t, t2); cross3(alp, r, t);
        ao[0] = sm.ao[par][0] + t[0] + t2[0]; ao[1] = sm.ao[par][1] + t[1] + t2[1]; ao[2] = sm.ao[par][2] + t[2] + t2[2];
        cross3(wp, axqd, t);
        al[0] = alp[0] + t[0]; al[1] = alp[1] + t[1]; al[2] = alp[2] + t[2];
        w[0] = wp[0] + axqd[0]; w[1] = wp[1] + axqd[1]; w[2] = wp[2] + axqd[2];
        sm.anchor[dof][0] = pos[0]; sm.anchor[dof][1] = pos[1]; sm.anchor[dof][2] = pos[2];
        sm.axis[dof][0] = ax[0]; sm.axis[dof][1] = ax[1]; sm.axis[dof][2] = ax[2];
      }
#pragma unroll
      for (int i = 0; i < 3; i++) { sm.xpos[lk][i] = pos[i]; sm.w[lk][i] = w[i]; sm.vo[lk][i] = vo[i]; sm.al[lk][i] = al[i]; sm.ao[lk][i] = ao[i]; }
#pragma unroll
      for (int i = 0; i < 9; i++) sm.xmat[lk][i] = mat[i];
      // inertial quantities of this link in world axes, positions relative to the pelvis origin
      double rc[3], c0[3], Iw[9], tmp[9], Rt[9];
      matvec3(mat, c3_link_ipos[lk], rc);
      const double m = c3_link_mass[lk];
#pragma unroll
      for (int i = 0; i < 3; i++) c0[i] = pos[i] + rc[i];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rt[3 * i + j] = mat[3 * j + i];
      matmul3(mat, &c3_link_inertia[lk][0][0], tmp);
      matmul3(tmp, Rt, Iw);
      double t[3], t2[3], ac[3];
      cross3(w, rc, t); cross3(w, t, t2); cross3(al, rc, t);
      ac[0] = ao[0] + t[0] + t2[0]; ac[1] = ao[1] + t[1] + t2[1]; ac[2] = ao[2] + t[2] + t2[2] - GRAVITY_Z;
      double Iwv[3], Nn[3];
      matvec3(Iw, al, Nn); matvec3(Iw, w, Iwv); cross3(w, Iwv, t);
#pragma unroll
      for (int i = 0; i < 3; i++) { sm.com[lk][i] = c0[i]; sm.F[lk][i] = m * ac[i]; sm.N[lk][i] = Nn[i] + t[i]; }
      sm.Iw[lk][0] = Iw[0]; sm.Iw[lk][1] = Iw[4]; sm.Iw[lk][2] = Iw[8]; sm.Iw[lk][3] = Iw[1]; sm.Iw[lk][4] = Iw[2]; sm.Iw[lk][5] = Iw[5];
    }
    lds_sync();
  }
  // ================= dof lanes: composite of the moved subtree, bias, row of M
  const int d = lane < NV ? lane : 0;
  const bool dvalid = lane < NV;
  const int dtype = c3_dof_type[d], dlink = c3_dof_link[d], dsub = c3_dof_submask[d];
  const double O[3] = {sm.xpos[0][0], sm.xpos[0][1], sm.xpos[0][2]};
  double om[3], vv[3];  // spatial velocity of this dof about the pelvis origin
  {
    const double ax[3] = {sm.axis[d][0], sm.axis[d][1], sm.axis[d][2]};
    if (dtype == 0) { om[0] = om[1] = om[2] = 0.0; vv[0] = ax[0]; vv[1] = ax[1]; vv[2] = ax[2]; }
    else {
      double pr[3] = {sm.anchor[d][0] - O[0], sm.anchor[d][1] - O[1], sm.anchor[d][2] - O[2]};
      om[0] = ax[0]; om[1] = ax[1]; om[2] = ax[2];
      cross3(pr, ax, vv);
    }
  }
  double cm = 0, ch[3] = {0, 0, 0}, cJ[6] = {0, 0, 0, 0, 0, 0};  // composite: mass, first moment, inertia (xx yy zz xy xz yz) about O
  double bias = 0.0;
  {
    double Fs[3] = {0, 0, 0}, Ts[3] = {0, 0, 0};  // resultant force and moment about this dof's anchor
    const double p[3] = {sm.anchor[d][0], sm.anchor[d][1], sm.anchor[d][2]};
    for (int l = 0; l < NL; l++) {
      if (!((dsub >> l) & 1)) continue;
      const double m = c3_link_mass[l];
      const double c0[3] = {sm.com[l][0], sm.com[l][1], sm.com[l][2]};
      const double r[3] = {c0[0] - O[0], c0[1] - O[1], c0[2] - O[2]};
      const double r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      cm += m; ch[0] += m * r[0]; ch[1] += m * r[1]; ch[2] += m * r[2];
      cJ[0] += sm.Iw[l][0] + m * (r2 - r[0] * r[0]); cJ[1] += sm.Iw[l][1] + m * (r2 - r[1] * r[1]); cJ[2] += sm.Iw[l][2] + m * (r2 - r[2] * r[2]);
      cJ[3] += sm.Iw[l][3] - m * r[0] * r[1]; cJ[4] += sm.Iw[l][4] - m * r[0] * r[2]; cJ[5] += sm.Iw[l][5] - m * r[1] * r[2];
      const double F[3] = {sm.F[l][0], sm.F[l][1], sm.F[l][2]};
      const double rp[3] = {c0[0] - p[0], c0[1] - p[1], c0[2] - p[2]};
      double t[3];
      cross3(rp, F, t);
      Fs[0] += F[0]; Fs[1] += F[1]; Fs[2] += F[2];
      Ts[0] += t[0] + sm.N[l][0]; Ts[1] += t[1] + sm.N[l][1]; Ts[2] += t[2] + sm.N[l][2];
    }
    const double ax[3] = {sm.axis[d][0], sm.axis[d][1], sm.axis[d][2]};
    bias = dtype == 0 ? ax[0] * Fs[0] + ax[1] * Fs[1] + ax[2] * Fs[2] : ax[0] * Ts[0] + ax[1] * Ts[1] + ax[2] * Ts[2];
  }
  if (dvalid) {
    sm.comp[d][0] = cm; sm.comp[d][1] = ch[0]; sm.comp[d][2] = ch[1]; sm.comp[d][3] = ch[2];
#pragma unroll
    for (int i = 0; i < 6; i++) sm.comp[d][4 + i] = cJ[i];
  }
  lds_sync();
  // Row d of M goes to this lane's (still private) row of sm.minv, one column per trip of a ROLLED loop: unrolled, the
  // twenty column bodies were scheduled on top of each other and spilled ~600 VGPRs to scratch.
#pragma unroll 1
  for (int J = 0; J < NV; J++) {
    const int jl = c3_dof_link[J], jsub = c3_dof_submask[J], jt = c3_dof_type[J];
    const bool j_below = (dsub >> jl) & 1, i_below = (jsub >> dlink) & 1;  // J's link moves with me / my link moves with J
    double val = 0.0;
    if (j_below || i_below) {
      // composite of the deeper dof's subtree (same subtree when the two dofs sit on the same link)
      double m, h[3], Jc[6];
      m = j_below ? sm.comp[J][0] : cm;
#pragma unroll
      for (int i = 0; i < 3; i++) h[i] = j_below ? sm.comp[J][1 + i] : ch[i];
#pragma unroll
      for (int i = 0; i < 6; i++) Jc[i] = j_below ? sm.comp[J][4 + i] : cJ[i];
      double oj[3], vj[3];
      const double ax[3] = {sm.axis[J][0], sm.axis[J][1], sm.axis[J][2]};
      if (jt == 0) { oj[0] = oj[1] = oj[2] = 0.0; vj[0] = ax[0]; vj[1] = ax[1]; vj[2] = ax[2]; }
      else {
        double pr[3] = {sm.anchor[J][0] - O[0], sm.anchor[J][1] - O[1], sm.anchor[J][2] - O[2]};
        oj[0] = ax[0]; oj[1] = ax[1]; oj[2] = ax[2];
        cross3(pr, ax, vj);
      }
      double Jo[3] = {Jc[0] * oj[0] + Jc[3] * oj[1] + Jc[4] * oj[2], Jc[3] * oj[0] + Jc[1] * oj[1] + Jc[5] * oj[2], Jc[4] * oj[0] + Jc[5] * oj[1] + Jc[2] * oj[2]};
      double t1[3], t2[3];
      cross3(oj, h, t1); cross3(om, h, t2);
      val = om[0] * Jo[0] + om[1] * Jo[1] + om[2] * Jo[2] + m * (vv[0] * vj[0] + vv[1] * vj[1] + vv[2] * vj[2]) +
            vv[0] * t1[0] + vv[1] * t1[1] + vv[2] * t1[2] + vj[0] * t2[0] + vj[1] * t2[1] + vj[2] * t2[2];
    }
    if (J == d) val += c3_dof_armature[d];
    if (dvalid) sm.minv[d][J] = val;
  }
  bias_out = bias;
  if (dbg && dvalid) { for (int J = 0; J < NV; J++) dbg[D3_M + d * NV + J] = sm.minv[d][J]; dbg[D3_BIAS + d] = bias; }
}

struct Out3 { int niter, nefc; bool overflow, capped; };

// ---------------------------------------------------------------- one mj_forward (+ Euler step) of one environment
template <int MR, bool CAP>
__device__ void substep3(Smem3<MR>& sm, int lane, double ctrl_l /* dof lane: command of its motor, pre-clamp */, bool integrate, Out3& out, double* dbg) {
  double bias;
  kin_mass3(sm, lane, bias, dbg);
  const int d = lane < NV ? lane : 0;
  const bool dvalid = lane < NV;
  const double damping = c3_dof_damping[d];
  double Mr[NV];
  static_for<0, NV>([&](auto jj) { constexpr int J = decltype(jj)::value; Mr[J] = dvalid ? sm.minv[d][J] : 0.0; });
  gauss_jordan20(Mr, lane);
  // ================= smooth acceleration
  double tau;
  {
    const int a = c3_dof_act[d];
    double u = 0.0;
    if (a >= 0) { u = ctrl_l; const double lo = c3_act_ctrlrange[a][0], hi = c3_act_ctrlrange[a][1]; u = u < lo ? lo : (u > hi ? hi : u); u *= c3_act_gear[a]; }
    tau = dvalid ? -damping * sm.v[d] - bias + u : 0.0;
  }
  double qs = 0.0;
  static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; qs += Mr[C] * rdlane(tau, C); });
  lds_sync();
  if (dvalid) {
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; sm.minv[d][C] = Mr[C]; });
    sm.qs[d] = qs;
  }
  if (dbg && dvalid) dbg[D3_QS + d] = qs;
  // ================= collision: sphere s on lane s (capsule ends are spheres for a plane, mjc_PlaneCapsule)
  bool con_act = false;
  if (lane < NSPH) {
    const int l = c3_sph_link[lane];
    double r[3], hw[3];
    matvec3(sm.xmat[l], c3_sph_pos[lane], r);
    const double cx = sm.xpos[l][0] + r[0], cy = sm.xpos[l][1] + r[1], cz = sm.xpos[l][2] + r[2];
    const double dist = cz - c3_sph_radius[lane];
    con_act = dist < 0;
    sm.sphc[lane][0] = cx; sm.sphc[lane][1] = cy; sm.sphc[lane][2] = cz - c3_sph_radius[lane] - 0.5 * dist;  // contact point
    sm.sphdist[lane] = dist;
    // mju_makeFrame with normal +z: first tangent = hint minus its normal part (spheres: world y), second = n x t1
    matvec3(sm.xmat[l], c3_sph_hint[lane], hw);
    const bool has_hint = c3_sph_hint[lane][0] != 0.0 || c3_sph_hint[lane][1] != 0.0 || c3_sph_hint[lane][2] != 0.0;
    double tx = has_hint ? hw[0] : 0.0, ty = has_hint ? hw[1] : 1.0;
    const double n = sqrt(tx * tx + ty * ty);
    if (n < MINVAL) { tx = 1.0; ty = 0.0; } else { tx /= n; ty /= n; }
    sm.spht1[lane][0] = tx; sm.spht1[lane][1] = ty;
  }
  bool lim_act = false;
  double lim_dist = 0.0, lim_sgn = 0.0;
  if (lane < NLIM) {
    const int dof = c3_lim_dof[lane];
    const double qd = sm.q[c3_dof_qadr[dof]];
    const double dlo = qd - c3_lim_range[lane][0], dhi = c3_lim_range[lane][1] - qd;
    if (dlo < 0) { lim_act = true; lim_dist = dlo; lim_sgn = 1.0; }
    else if (dhi < 0) { lim_act = true; lim_dist = dhi; lim_sgn = -1.0; }
  }
  unsigned con_mask = (unsigned)__ballot(con_act);
  const unsigned lim_mask = (unsigned)__ballot(lim_act);
  int ncon = __popc(con_mask);
  const int nlim = __popc(lim_mask);
  out.capped = false;
  if (CAP) {  // last-resort kernel: keep the first contacts (MuJoCo order) that fit, say so
    const int room = (MR - 3 * NEQ - nlim) / 3;
    while (ncon > room) { con_mask &= ~(1u << (31 - __clz(con_mask))); ncon--; out.capped = true; }
  }
  const int nrows = 3 * NEQ + nlim + 3 * ncon;
  out.nefc = nrows;
  out.overflow = nrows > MR;
  if (out.overflow) { out.niter = 0; return; }  // wave-uniform
  lds_sync();
  // ================= the row owned by this lane: up to two (link, point, sign) point-Jacobian terms along `dir`
  int kind = K_NONE, cbase = lane;
  double pos = 0.0, invw = 0.0;
  const double* solref = c3_contact_solref;
  const double* solimp = c3_contact_solimp;
  int mask1 = 0, mask2 = 0, limdof = -1;
  double p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0}, dir[3] = {0, 0, 0};
  if (lane < 3 * NEQ) {
    kind = K_EQ;
    const int e = lane / 3, comp = lane % 3;
    const int l1 = c3_eq_link1[e], l2 = c3_eq_link2[e];
    double r[3];
    matvec3(sm.xmat[l1], c3_eq_p1[e], r);
    p1[0] = sm.xpos[l1][0] + r[0]; p1[1] = sm.xpos[l1][1] + r[1]; p1[2] = sm.xpos[l1][2] + r[2];
    matvec3(sm.xmat[l2], c3_eq_p2[e], r);
    p2[0] = sm.xpos[l2][0] + r[0]; p2[1] = sm.xpos[l2][1] + r[1]; p2[2] = sm.xpos[l2][2] + r[2];
    dir[comp] = 1.0;
    mask1 = c3_link_dofmask[l1]; mask2 = c3_link_dofmask[l2];
    pos = p1[comp] - p2[comp];
    invw = c3_eq_invweight[e];
    solref = c3_eq_solref[e]; solimp = c3_eq_solimp[e];
  } else if (lane < 3 * NEQ + nlim) {
    kind = K_LIM;
    cbase = nth_set(lim_mask, lane - 3 * NEQ);  // the lane that tested this limit (shuffle source below)
    limdof = c3_lim_dof[cbase];
    invw = c3_dof_invweight[limdof];
    solref = c3_limit_solref; solimp = c3_limit_solimp;
  } else if (lane < nrows) {
    const int k = (lane - 3 * NEQ - nlim) / 3, comp = (lane - 3 * NEQ - nlim) % 3;
    kind = comp == 0 ? K_CN : K_CT;
    cbase = lane - comp;
    const int s = nth_set(con_mask, k);
    const double tx = sm.spht1[s][0], ty = sm.spht1[s][1];
    dir[0] = comp == 0 ? 0.0 : (comp == 1 ? tx : -ty); dir[1] = comp == 0 ? 0.0 : (comp == 1 ? ty : tx); dir[2] = comp == 0 ? 1.0 : 0.0;
    p1[0] = sm.sphc[s][0]; p1[1] = sm.sphc[s][1]; p1[2] = sm.sphc[s][2];
    mask1 = c3_link_dofmask[c3_sph_link[s]];
    pos = comp == 0 ? sm.sphdist[s] : 0.0;
    invw = c3_sph_invweight[s];
  }
  double lim_s = 0.0;
  {
    // joint-limit rows take (distance, side) from the lane that tested the limit
    const int src = kind == K_LIM ? cbase : 0;
    const double ld = __shfl(lim_dist, src), ls = __shfl(lim_sgn, src);
    if (kind == K_LIM) { pos = ld; lim_s = ls; cbase = lane; }
  }
  const bool active = kind != K_NONE;
  const bool inrow = lane < MR;        // lanes beyond the row capacity of this instantiation own no LDS row / column
  const int rl = inrow ? lane : 0;
  double vel = 0.0, bq = 0.0, jw = 0.0;
#pragma unroll 1
  for (int j = 0; j < NV; j++) {  // rolled on purpose (register pressure); writes this lane's own LDS row
    const double ax[3] = {sm.axis[j][0], sm.axis[j][1], sm.axis[j][2]};
    const double an[3] = {sm.anchor[j][0], sm.anchor[j][1], sm.anchor[j][2]};
    const bool slide = c3_dof_type[j] == 0;
    double val = 0.0;
    if ((mask1 >> j) & 1) {
      double r[3] = {p1[0] - an[0], p1[1] - an[1], p1[2] - an[2]}, c[3];
      cross3(ax, r, c);
      val += slide ? dir[0] * ax[0] + dir[1] * ax[1] + dir[2] * ax[2] : dir[0] * c[0] + dir[1] * c[1] + dir[2] * c[2];
    }
    if ((mask2 >> j) & 1) {
      double r[3] = {p2[0] - an[0], p2[1] - an[1], p2[2] - an[2]}, c[3];
      cross3(ax, r, c);
      val -= slide ? dir[0] * ax[0] + dir[1] * ax[1] + dir[2] * ax[2] : dir[0] * c[0] + dir[1] * c[1] + dir[2] * c[2];
    }
    if (j == limdof) val = lim_s;
    if (inrow) sm.rowJ[lane][j] = val;
    vel += val * sm.v[j]; bq += val * sm.qs[j]; jw += val * sm.ws[j];
  }
  double R, aref;
  {
    double tc = solref[0] < 2.0 * H ? 2.0 * H : solref[0];
    const double dr = solref[1], dmax = solimp[1];
    const double kk = 1.0 / (dmax * dmax * tc * tc * dr * dr), bb = 2.0 / (dmax * tc);
    const double imp = impedance3(solimp, pos);
    R = (1.0 - imp) / imp * invw;
    R = R > MINVAL ? R : MINVAL;
    aref = -bb * vel - kk * imp * pos;
  }
  R = __shfl(R, cbase);  // friction rows share the normal row's regulariser (impratio 1, isotropic friction)
  const double b = active ? bq - aref : 0.0;
  const double jar = jw - aref;
  // X = J M^-1 (M^-1 is symmetric: its row j is read as a contiguous broadcast)
  double X[NV];
  static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; X[C] = 0.0; });
#pragma unroll 1
  for (int j = 0; j < NV; j++) {
    const double Jj = sm.rowJ[rl][j];
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; X[C] += sm.minv[j][C] * Jj; });
  }
  if (dbg && active) { for (int j = 0; j < NV; j++) dbg[D3_J + lane * NV + j] = sm.rowJ[rl][j]; dbg[D3_AREF + lane] = aref; }
  lds_sync();
  double Adiag = 1.0;
  for (int c = 0; c < nrows; c++) {
    double a = 0.0;
    static_for<0, NV>([&](auto jj) { constexpr int Jx = decltype(jj)::value; a += X[Jx] * sm.rowJ[c][Jx]; });
    if (c == lane) { a += R; Adiag = a; }
    if (inrow && c <= lane) sm.A[Smem3<MR>::tri(c) + lane] = active ? a : 0.0;   // upper triangle: (c, lane), c <= lane
  }
  lds_sync();
  const double Ainv = 1.0 / Adiag;
  // ================= warm start (mj_constraintUpdate on qacc_warmstart), kept only if its dual cost beats zero force
  const double mu = MU;
  double f = 0.0;
  {
    const double D = 1.0 / R;
    const double jn = __shfl(jar, cbase), j1 = __shfl(jar, cbase + 1 < 64 ? cbase + 1 : 63), j2 = __shfl(jar, cbase + 2 < 64 ? cbase + 2 : 63);
    if (kind == K_EQ) f = -D * jar;
    else if (kind == K_LIM) f = jar < 0 ? -D * jar : 0.0;
    else if (kind == K_CN || kind == K_CT) {
      const int comp = lane - cbase;
      const double N = jn * mu, U1 = j1 * mu, U2 = j2 * mu, T = sqrt(U1 * U1 + U2 * U2);
      double fn, ft;
      const double jown = comp == 0 ? jn : (comp == 1 ? j1 : j2), Uown = comp == 1 ? U1 : U2;
      if (N >= mu * T || (T <= 0 && N >= 0)) { fn = 0; ft = 0; }
      else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { fn = -D * jn; ft = -D * jown; }
      else {
        const double Dm = D / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
        fn = -Dm * NmT * mu;
        ft = -fn / T * Uown * mu;
      }
      f = comp == 0 ? fn : ft;
    }
  }
  double res = 0.0;
  const int trl = Smem3<MR>::tri(rl);
  for (int c = 0; c < nrows; c++) res += sm.a_at(c, rl, trl) * rdlane_dyn(f, c);
  {
    const double cost = wave_sum(active ? f * (0.5 * res + b) : 0.0);
    if (cost > 0) { f = 0.0; res = 0.0; }
  }
  res += b;
  // ================= PGS (mj_solPGS, elliptic cones)
  const double scale = 1.0 / (MEANINERTIA * NV);
  int niter = 0;
  const bool isLim = kind == K_LIM;
  const double hAdiag = 0.5 * Adiag;
  for (int iter = 0; iter < ITERATIONS; iter++) {
    double improvement = 0.0;  // contact rows: wave-uniform; single rows: accumulated on the owner lane, reduced once per sweep
    double acc = 0.0;
    for (int K = 0; K < nrows; K++) {
      const int kindK = __builtin_amdgcn_readlane(kind, K);
      if (kindK == K_EQ || kindK == K_LIM) {
        // every lane evaluates the single-row update of ITS OWN row from its own registers; only lane K's result is used
        double nf = f - res * Ainv;
        nf = isLim ? fmax(nf, 0.0) : nf;
        double dOwn = nf - f;
        const double chg = dOwn * (hAdiag * dOwn + res);
        const bool keep = chg <= 1e-10;
        dOwn = keep ? dOwn : 0.0;
        acc += (keep && lane == K) ? chg : 0.0;
        const double dK = rdlane_dyn(dOwn, K);
        res += sm.a_at(K, rl, trl) * dK;
        if (lane == K) f += dK;
      } else {  // contact: rows K (normal), K+1, K+2 (tangents)
        const double o0 = rdlane_dyn(f, K), o1 = rdlane_dyn(f, K + 1), o2 = rdlane_dyn(f, K + 2);
        const double r0 = rdlane_dyn(res, K), r1 = rdlane_dyn(res, K + 1), r2 = rdlane_dyn(res, K + 2);
        // symmetric 3x3 diagonal block of A (the mirrored entries agree to rounding; one of each pair is read)
        const int t0 = Smem3<MR>::tri(K), t1 = Smem3<MR>::tri(K + 1), t2 = Smem3<MR>::tri(K + 2);
        const double A00 = sm.A[t0 + K], A01 = sm.A[t0 + K + 1], A02 = sm.A[t0 + K + 2];
        const double A11 = sm.A[t1 + K + 1], A12 = sm.A[t1 + K + 2], A22 = sm.A[t2 + K + 2];
        // Straight-line selects instead of branches: every value here is wave-uniform, but the compiler cannot know that and
        // would emit exec-mask branches (VALU compare -> SALU -> taken branch) on the critical path of the solver.
        // normal-only update (taken when the normal force is ~0)
        const double fn_n = fmax(o0 - r0 * rdlane_dyn(Ainv, K), 0.0);
        // ray update: scale the force vector by (1 + x), x clamped so that the normal force stays >= 0
        const double v1_0 = A00 * o0 + A01 * o1 + A02 * o2, v1_1 = A01 * o0 + A11 * o1 + A12 * o2, v1_2 = A02 * o0 + A12 * o1 + A22 * o2;
        const double denom = o0 * v1_0 + o1 * v1_1 + o2 * v1_2;
        double x = -(o0 * r0 + o1 * r1 + o2 * r2) * fast_rcp(denom);
        x = fmax(x, -1.0);
        x = denom >= MINVAL ? x : 0.0;
        const bool use_n = o0 < MINVAL;
        const double f0 = use_n ? fn_n : o0 + x * o0;
        double f1 = use_n ? 0.0 : o1 + x * o1, f2 = use_n ? 0.0 : o2 + x * o2;
        {  // friction: QCQP on the cone given the normal force (result used only if f0 >= MINVAL)
          const double bc1 = r1 - (A11 * o1 + A12 * o2) + A01 * (f0 - o0);
          const double bc2 = r2 - (A12 * o1 + A22 * o2) + A02 * (f0 - o0);
          // mju_QCQP2, first Newton iterate (lambda = 0) inline; further iterates only for a sliding contact
          const double b1 = bc1 * mu, b2 = bc2 * mu, Q11 = A11 * (mu * mu), Q22 = A22 * (mu * mu), Q12 = A12 * (mu * mu);
          const double det0 = Q11 * Q22 - Q12 * Q12;
          const double di0 = fast_rcp(det0);
          double v1 = -(Q22 * di0) * b1 + (Q12 * di0) * b2, v2 = (Q12 * di0) * b1 - (Q11 * di0) * b2;
          const double val0 = v1 * v1 + v2 * v2 - f0 * f0;
          double la = 0.0;
          bool degenerate = det0 < 1e-10;
          if (__ballot(!degenerate && val0 >= 1e-10 && f0 >= MINVAL) != 0) {  // wave-uniform: sliding contact
            double val = val0, P11 = Q22 * di0, P22 = Q11 * di0, P12 = -Q12 * di0;
            for (int it = 0; it < 20; it++) {
              const double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
              const double delta = -val * fast_rcp(deriv);
              if (delta < 1e-10) break;
              la += delta;
              if (it == 19) break;  // iteration budget of mju_QCQP2: the last multiplier is kept, v is not recomputed
              const double det = (Q11 + la) * (Q22 + la) - Q12 * Q12;
              if (det < 1e-10) { degenerate = true; break; }
              const double di = fast_rcp(det);
              P11 = (Q22 + la) * di; P22 = (Q11 + la) * di; P12 = -Q12 * di;
              v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
              val = v1 * v1 + v2 * v2 - f0 * f0;
              if (val < 1e-10) break;
            }
          }
          double q1 = degenerate ? 0.0 : v1 * mu, q2 = degenerate ? 0.0 : v2 * mu;
          if (la != 0.0 && !degenerate) {  // active constraint: put the friction exactly on the cone
            double s = (q1 * q1 + q2 * q2) * (1.0 / (MU * MU));
            s = sqrt(f0 * f0 * fast_rcp(s > MINVAL ? s : MINVAL));
            q1 *= s; q2 *= s;
          }
          const bool fr = f0 >= MINVAL;
          f1 = fr ? q1 : f1; f2 = fr ? q2 : f2;
        }
        double d0 = f0 - o0, d1 = f1 - o1, d2 = f2 - o2;
        const double chg = 0.5 * (d0 * (A00 * d0 + A01 * d1 + A02 * d2) + d1 * (A01 * d0 + A11 * d1 + A12 * d2) + d2 * (A02 * d0 + A12 * d1 + A22 * d2)) +
                           d0 * r0 + d1 * r1 + d2 * r2;
        const bool keep = chg <= 1e-10;
        d0 = keep ? d0 : 0.0; d1 = keep ? d1 : 0.0; d2 = keep ? d2 : 0.0;
        improvement -= keep ? chg : 0.0;
        res += sm.a_at(K, rl, trl) * d0 + sm.a_at(K + 1, rl, trl) * d1 + sm.a_at(K + 2, rl, trl) * d2;
        if (lane == K) f += d0;
        if (lane == K + 1) f += d1;
        if (lane == K + 2) f += d2;
        K += 2;
      }
    }
    niter = iter + 1;
    improvement -= wave_sum(acc);
    if (improvement * scale < TOLERANCE) break;
  }
  out.niter = nrows > 0 ? niter : 0;
  if (dbg && active) dbg[D3_F + lane] = f;
  // ================= total force g = tau + J' f;  qacc = M^-1 g;  Euler with implicit joint damping: (M + h B)^-1 g
  double g = dvalid ? tau : 0.0;
  for (int r = 0; r < nrows; r++) {
    const double fr = rdlane_dyn(f, r);
    g += dvalid ? sm.rowJ[r][d] * fr : 0.0;
  }
  double qacc = 0.0, qacch = 0.0;
  {
    double a0 = 0.0, a1 = 0.0;
    static_for<0, NV>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      const double t = sm.minv[d][C] * rdlane(g, C);
      if constexpr (C & 1) a1 += t; else a0 += t;
    });
    qacc = a0 + a1;
  }
  // Implicit joint damping of mj_Euler, qacch = (M + h B)^-1 g = (I + E)^-1 qacc with E = M^-1 h B, by the fixed-point iteration
  // x <- qacc - E x (see the planar kernel, cassie_kernels_g16.hip): the eigenvalues of E are those of h B^1/2 M^-1 B^1/2, at most
  // 0.0393 for this model too (the knee-spring dofs; tests/test_implicit_damping_bound.py samples poses through the oracle), so
  // IMPLICIT_DAMPING_SWEEPS3 = 12 sweeps leave 1e-17.  Replaces a second 20 x 20 Gauss-Jordan per substep and its 3.2 KB of LDS.
  // The six base dofs are undamped: their columns of E are zero.
  {
    double me[NV - 6];
    static_for<6, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; me[C - 6] = sm.minv[d][C] * (H * c3_dof_damping[C]); });
    double x = qacc;
#pragma unroll
    for (int it = 0; it < IMPLICIT_DAMPING_SWEEPS3; it++) {
      double s0 = qacc, s1 = 0.0;
      static_for<6, NV>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        const double xc = rdlane(x, C);
        if constexpr (C & 1) s1 = __builtin_fma(-me[C - 6], xc, s1); else s0 = __builtin_fma(-me[C - 6], xc, s0);
      });
      x = s0 + s1;
    }
    qacch = x;
  }
  if (dbg && dvalid) { dbg[D3_QACC + d] = qacc; if (d == 0) dbg[D3_NEFC] = nrows; }
  lds_sync();
  if (dvalid) {
    sm.ws[d] = qacc;
    if (integrate) sm.v[d] = sm.v[d] + H * qacch;
  }
  lds_sync();
  if (integrate) {
    if (lane < 3) sm.q[lane] += H * sm.v[lane];
    if (lane >= 6 && lane < NV) sm.q[lane + 1] += H * sm.v[lane];
    if (lane == 3) {
      // mju_quatIntegrate: quat <- normalize(quat) * axisangle(omega_body, h |omega|)
      const double wx = sm.v[3], wy = sm.v[4], wz = sm.v[5], wn = sqrt(wx * wx + wy * wy + wz * wz);
      double ax = 1.0, ay = 0.0, az = 0.0, ang = 0.0;
      if (wn >= MINVAL) { ax = wx / wn; ay = wy / wn; az = wz / wn; ang = H * wn; }
      const double sh = sin(0.5 * ang), r0 = cos(0.5 * ang), r1 = ax * sh, r2 = ay * sh, r3 = az * sh;
      const double n = sqrt(sm.q[3] * sm.q[3] + sm.q[4] * sm.q[4] + sm.q[5] * sm.q[5] + sm.q[6] * sm.q[6]);
      const double a0 = sm.q[3] / n, a1 = sm.q[4] / n, a2 = sm.q[5] / n, a3 = sm.q[6] / n;
      sm.q[3] = a0 * r0 - a1 * r1 - a2 * r2 - a3 * r3;
      sm.q[4] = a0 * r1 + a1 * r0 + a2 * r3 - a3 * r2;
      sm.q[5] = a0 * r2 - a1 * r3 + a2 * r0 + a3 * r1;
      sm.q[6] = a0 * r3 + a1 * r2 - a2 * r1 + a3 * r0;
    }
    lds_sync();
  }
}

// ---------------------------------------------------------------- n_sub torque-mode substeps of every environment
// Two instantiations are launched back to back (cassie_cabi.hip): <MAXR_FAST, 2> does every environment that needs at most 32
// constraint rows at twice the occupancy and hands the others over through `pending` (substeps left, state saved at that
// point); <MAXR, 1> finishes those and returns at once for everyone else.
template <int MR, int WPS>
__global__ void __launch_bounds__(64, WPS) env_step3d_kernel(Params3 p) {
  __shared__ Smem3<MR> sm;
  constexpr bool CAP = MR == MAXR;  // the general kernel is the last resort: it caps instead of handing over
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= p.n_envs) return;
  const int n_sub = p.pending_in ? p.pending_in[env] : p.n_sub;
  if (n_sub == 0) {   // nothing left for this environment (a middle tier still has to say so to the tier behind it)
    if (p.pending_out && lane == 0) p.pending_out[env] = 0;
    return;
  }
  if (CAP && p.pending_in && p.stats && lane == 0) atomicAdd(p.stats + S3_GENERAL_SUBSTEPS, (unsigned long long)n_sub);
  double* st = p.state + (size_t)env * ENV3_STRIDE;
  if (lane < NQ) sm.q[lane] = st[E3_Q + lane];
  if (lane < NV) { sm.v[lane] = st[E3_V + lane]; sm.ws[lane] = st[E3_WS + lane]; }
  double time = st[E3_TIME];
  const int a = lane < NV ? c3_dof_act[lane] : -1;
  double ctrl_l = 0.0;
  if (a >= 0) ctrl_l = p.actions ? p.actions[(size_t)env * NU + a] : st[E3_CTRL + a];
  lds_sync();
  Out3 out; out.niter = 0; out.nefc = 0; out.overflow = false; out.capped = false;
  int niter_sum = p.pending_in ? (int)st[E3_NITER] : 0;
  double* dbg = p.debug ? p.debug + (size_t)env * D3_STRIDE : nullptr;
  int left = 0, ncapped = 0;
  for (int sub = 0; sub < n_sub; sub++) {
    substep3<MR, CAP>(sm, lane, ctrl_l, p.integrate != 0, out, dbg);
    if (out.overflow) { left = n_sub - sub; break; }  // (never with CAP) detected before anything of this substep was written
    niter_sum += out.niter;
    ncapped += out.capped ? 1 : 0;
    if (p.integrate) time += H;
  }
  if (lane < NQ) st[E3_Q + lane] = sm.q[lane];
  if (lane < NV) { st[E3_V + lane] = sm.v[lane]; st[E3_WS + lane] = sm.ws[lane]; }
  if (a >= 0) st[E3_CTRL + a] = ctrl_l;
  if (lane == 0) {
    st[E3_TIME] = time; st[E3_NITER] = (double)niter_sum;
    if (left == 0) st[E3_NEFC] = (double)out.nefc;
    if (ncapped) { st[E3_OVF] += (double)ncapped; if (p.stats) atomicAdd(p.stats + S3_CAPPED_SUBSTEPS, (unsigned long long)ncapped); }
    if (p.pending_out) p.pending_out[env] = left;
  }
}

__global__ void env_init3d_kernel(double* state, int n, const double* qpos, const double* qvel) {
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= n) return;
  double* st = state + (size_t)env * ENV3_STRIDE;
  for (int i = lane; i < ENV3_STRIDE; i += blockDim.x) {
    double v = 0.0;
    if (i < NQ) v = qpos ? qpos[(size_t)env * NQ + i] : c3_qpos_init[i];
    else if (i >= E3_V && i < E3_V + NV) v = qvel ? qvel[(size_t)env * NV + (i - E3_V)] : 0.0;
    st[i] = v;
  }
}

}  // namespace cassie3d
#endif
