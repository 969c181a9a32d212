// cassie_ctrl.hip -- in-loop controllers on the MI355X path (included by tu_ctrl.hip / tu_ctrl_g16.hip after cassie_kernels.hip).
//
//   Cassie2d::StepJacobian        src/Cassie2d/Cassie2d.cpp:119-177   (Jacobian-transpose force controller)
//   Cassie2d::StepOsc             src/Cassie2d/Cassie2d.cpp:179-209   -> OSC_RBDL::RunPTSC / SolveQP (src/OSC_RBDL.cpp:114-291)
//   DynamicState::UpdateDynamicState  src/DynamicState.cpp:45-91     (M, bias, Bt, Jc, Jeq, JeqdotQdot; RBDL semantics)
//   pseudoinverse                 src/HelperFunctions.h:8-29
//   standing_controller_osc / _jacobian   rllab/envs/cassie2d.py:263-331
//
// Planar form: the world-y rows/columns of the reference's matrices are identically zero for this mechanism and are
// dropped.  The OSC QP (39 variables, 13 equalities, 32 pyramid rows, bounds) is solved in the equivalent reduced form
//   z = (u[6], lambda[8]),  qdd = Hinv (Nc Bt u + Nc Jc' F(lambda) + ce)   (equality block eliminated, M invertible)
//   contact force of site c:  fx = mu (l1 - l2), fz = l1 + l2, l >= 0        (generators of the 2-D friction cone)
//   min (T z + t0)' W (T z + t0) + 1/2 1e-4 sum (fx^2 + fz^2),  u in ctrlrange, lambda >= 0
// i.e. a 14-variable box-constrained strictly convex QP, solved by a primal active-set method whose linear systems are
// 14x14 masked Gauss-Jordan solves with rows on lanes (tools/planar_proto.py::box_qp is the executable spec; the oracle
// solves the reference's literal 39-variable formulation, so parity also checks this reduction).
#ifndef CASSIE_CTRL_HIP_
#define CASSIE_CTRL_HIP_

namespace cassie {

constexpr int NCR = 15;  // controller rows: 0..3 Jeq (Lx,Lz,Rx,Rz), 4..13 target sites 1..5 (x,z), 14 pitch
constexpr int NZ = 14;   // QP variables
constexpr double OSC_W_COM = 5.0, OSC_W_STANCE = 10.0, OSC_W_REST = 0.1, OSC_W_F = 1e-4, OSC_MU = 0.5;  // OSC_RBDL.h:92-98, RobotInterface.h:64

// The controller code below is ROW-GENERIC: `l` is the lane index inside a 16-lane DPP row and `rowok` says whether the row
// hosts a problem.  The wave-per-environment kernel runs it with one live row (rowok = lane < 16); the 4-envs-per-wave
// kernel runs it with four.  Every cross-lane step is a DPP row operation and every loop is controlled by wave-wide
// "any row still busy" ballots, so rows never diverge around a DPP instruction.
// Scheduling fence between controller phases: keeps the loads and temporaries of one phase from being hoisted into the
// previous one (where they only add register pressure).
__device__ __forceinline__ void ctrl_fence() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

struct CtrlSmem {
  double Jc[NCR][8];   // controller rows, compact: 3 base columns + the 5 joint columns of the row's own leg (the other leg's are 0)
  double acc[16];      // JdotQdot of each row (velocity-product acceleration, no gravity)
  double JH[4][NV];    // Jeq Hinv
  double S4[16];       // Jeq Hinv Jeq' (4x4)
  union {
    struct { double T[NZ][12]; double t0[12]; };  // OSC: T[var][target row], t0
    struct { double U[6][NV]; };                  // Jacobian controller: Nc Bt columns
  };
  double bias[16];     // NonlinearEffects + damping*qvel (DynamicState.cpp:47-52)
  double y[16];
  double act[8];       // controller input (7 accelerations or 6 forces)
  double u[8];         // controller output
  double s18[18];
};

// Leg whose joint columns a controller row touches: rows 2,3 (right loop closure) and 10..13 (right foot sites) the right
// leg's, every other row the left leg's (rows 4,5 = pelvis site and 14 = pitch have base columns only; their joint part is 0).
__host__ __device__ constexpr int ctrl_row_leg(int r) { return (r == 2 || r == 3 || (r >= 10 && r < 14)) ? 1 : 0; }
// Entry (r, C) of the dense 15 x 13 controller Jacobian: C is static, the row (and its leg) may be per-lane values.
template <int C, class CS>
__device__ __forceinline__ double jd(const CS& cs, int r, int rleg) {
  if constexpr (C < 3) return cs.Jc[r][C];
  else {
    constexpr int LEGC = (C - 3) / 5, K = 3 + (C - 3) % 5;
    return rleg == LEGC ? cs.Jc[r][K] : 0.0;
  }
}
// ... with a static row: structural zeros are literal zeros (the multiply-adds on them disappear).
template <int R, int C, class CS>
__device__ __forceinline__ double jd(const CS& cs) {
  if constexpr (C < 3) return cs.Jc[R][C];
  else if constexpr (ctrl_row_leg(R) == (C - 3) / 5) return cs.Jc[R][3 + (C - 3) % 5];
  else return 0.0;
}
// Hinv is stored by rows on the dof lanes; both layouts (dense in Smem, packed upper triangle in the 4-envs-per-wave LDS)
// are read through SM::hidx(r, c), which always lands on the entry row min(r, c) wrote -- the two kernels therefore see
// bit-identical matrices.

// pseudoinverse of a symmetric 4x4 (singular values = |eigenvalues|, threshold tol) on per-environment registers.
// Fast path: S = Jeq Hinv Jeq' is positive definite with eigenvalues ~0.2 .. 11 in every pose the robot reaches, far above the
// 1e-3 threshold of pseudoinverse(.,1e-3) (Cassie2d.cpp:134, OSC_RBDL.cpp:171).  When that holds the pseudoinverse IS the
// inverse, which a 4-pivot Gauss-Jordan gives in ~150 instructions instead of ~5000 for the Jacobi eigen-decomposition.  It is
// certified without eigenvalues: lambda_min(S) = 1/lambda_max(S^-1) >= 1/||S^-1||_F, so ||S^-1||_F * tol < 1 with positive
// pivots proves that no singular value is at or below tol.  Otherwise: cyclic Jacobi, as before.
__device__ __forceinline__ void pinv_sym4(const double* Sin /*LDS 16*/, double tol, double (&P)[16], bool noshort) {
  double a[4][4], V[4][4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = 0.5 * (Sin[4 * i + j] + Sin[4 * j + i]); V[i][j] = i == j ? 1.0 : 0.0; }
  {
    double g[4][4];
    bool posdef = true;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) g[i][j] = a[i][j];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // in-place Gauss-Jordan inverse of an SPD matrix (no pivoting needed)
      const double piv = g[k][k];
      posdef = posdef && piv > 0.0;
      const double inv = 1.0 / piv;
#pragma unroll
      for (int j = 0; j < 4; j++) if (j != k) g[k][j] *= inv;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (i == k) continue;
        const double t = g[i][k];
#pragma unroll
        for (int j = 0; j < 4; j++) if (j != k) g[i][j] -= t * g[k][j];
        g[i][k] = -t * inv;
      }
      g[k][k] = inv;
    }
    double fro2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) fro2 += g[i][j] * g[i][j];
    if (!noshort && posdef && fro2 * tol * tol < 1.0) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) P[4 * i + j] = 0.5 * (g[i][j] + g[j][i]);
      return;
    }
  }
  for (int sweep = 0; sweep < 10; sweep++) {
    double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[0][3]) + fabs(a[1][2]) + fabs(a[1][3]) + fabs(a[2][3]);
    double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]) + fabs(a[3][3]);
    if (off <= 1e-17 * dia) break;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double apq = a[p][q];
        if (fabs(apq) > 1e-300) {
          double th = (a[q][q] - a[p][p]) / (2.0 * apq);
          double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(1.0 + th * th));
          double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
#pragma unroll
          for (int k = 0; k < 4; k++) { double kp = a[k][p], kq = a[k][q]; a[k][p] = cs * kp - sn * kq; a[k][q] = sn * kp + cs * kq; }
#pragma unroll
          for (int k = 0; k < 4; k++) { double pk = a[p][k], qk = a[q][k]; a[p][k] = cs * pk - sn * qk; a[q][k] = sn * pk + cs * qk; }
#pragma unroll
          for (int k = 0; k < 4; k++) { double vp = V[k][p], vq = V[k][q]; V[k][p] = cs * vp - sn * vq; V[k][q] = sn * vp + cs * vq; }
        }
      }
  }
  double winv[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { double w = a[k][k]; winv[k] = fabs(w) > tol ? 1.0 / w : 0.0; }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) s += V[i][k] * V[j][k] * winv[k];
      P[4 * i + j] = s;
    }
}

// ---------------------------------------------------------------- DynamicState + constraint projector (both controllers)
// Leaves in LDS: sm.minv = Hinv (RBDL semantics, read through SM::hidx), cs.Jc, cs.acc, cs.JH, cs.bias; returns the wave-uniform P4 = (Jeq Hinv Jeq')^+
// and g4 = P4 * JeqdotQdot.
template <class SM, class CS>
__device__ __forceinline__ void ctrl_dyn(SM& sm, CS& cs, const LaneConst& c, int l, bool rowok, double (&P4)[16], double (&g4)[4], bool noshort) {
  const int lane = rowok ? l : 63;  // role tests below are written against `lane`; dead rows take no role
  planar_fk<1>(sm, sm.q, sm.v, c, lane);
  {
    DofConst dc;
    load_dof_const(dc, c);
    double Mr[NV], bias;
    mass_rows<1>(sm, c, dc, l, Mr, bias, false);
    gauss_jordan_rows_legs<true>(Mr, l);
    if (c.dvalid && c.grp == 0 && rowok) {
      static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; if (C >= c.d) sm.minv[SM::hidx(c.d, C)] = Mr[C]; });
      cs.bias[c.d] = bias + dc.damping * sm.v[c.d];
    }
  }
  // controller rows
  double J[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double racc = 0.0;
  int leg = 0;
  if (lane < 4) {
    leg = lane >> 1;
    const int comp = lane & 1;
    const int l1 = cp_slot_link1[lane], l2 = cp_slot_link2[lane];
    double p1x, p1z, p2x, p2z;
    link_point(sm, l1, cp_slot_d1[1][lane][0], cp_slot_d1[1][lane][1], p1x, p1z);
    link_point(sm, l2, cp_slot_d2[1][lane][0], cp_slot_d2[1][lane][1], p2x, p2z);
    const int lb = leg == 0 ? 1 : 6;
    jac_compact(sm, cp_link_pathmask8[l1], lb, comp, p1x, p1z, 1.0, J);
    jac_compact(sm, cp_link_pathmask8[l2], lb, comp, p2x, p2z, -1.0, J);
    double w1 = sm.lw[l1], w2 = sm.lw[l2];
    double a1 = comp == 0 ? sm.lax[l1] - w1 * w1 * (p1x - sm.lox[l1]) : sm.laz[l1] - w1 * w1 * (p1z - sm.loz[l1]);
    double a2 = comp == 0 ? sm.lax[l2] - w2 * w2 * (p2x - sm.lox[l2]) : sm.laz[l2] - w2 * w2 * (p2z - sm.loz[l2]);
    racc = a1 - a2;  // the gravity offset of laz cancels in the difference
  } else if (lane < 14) {
    const int sid = 1 + ((lane - 4) >> 1), comp = (lane - 4) & 1;
    const int l = cp_site_link[sid];
    leg = l <= 5 ? 0 : 1;
    double px, pz;
    link_point(sm, l, cp_site_d[1][sid][0], cp_site_d[1][sid][1], px, pz);
    jac_compact(sm, cp_link_pathmask8[l], leg == 0 ? 1 : 6, comp, px, pz, 1.0, J);
    double w = sm.lw[l];
    racc = comp == 0 ? sm.lax[l] - w * w * (px - sm.lox[l]) : (sm.laz[l] - CP_GRAVITY) - w * w * (pz - sm.loz[l]);
  } else if (lane == 14) {
    J[2] = 1.0;  // AddQDDIdx(2): body pitch (Cassie2d.cpp:41)
  }
  const int vbase = leg == 0 ? 3 : 8;
  if (lane < NCR) {
#pragma unroll
    for (int k = 0; k < 8; k++) cs.Jc[lane][k] = J[k];
    cs.acc[lane] = racc;
  }
  lds_sync();
  ctrl_fence();
  // JH = Jeq Hinv (rows 0..3) and S4 = JH Jeq'
  if (lane < 4) {
    double X[NV];
    static_for<0, NV>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      double s = sm.minv[SM::hidx(C, 0)] * J[0] + sm.minv[SM::hidx(C, 1)] * J[1] + sm.minv[SM::hidx(C, 2)] * J[2];
      static_for<0, 5>([&](auto kk) { constexpr int K = decltype(kk)::value; s += sm.minv[SM::hidx(C, vbase + K)] * J[3 + K]; });
      X[C] = s;
      cs.JH[lane][C] = s;
    });
    static_for<0, 4>([&](auto ss) {
      constexpr int S = decltype(ss)::value;
      double d = 0;
      static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; d += X[C] * jd<S, C>(cs); });
      cs.S4[4 * lane + S] = d;
    });
  }
  lds_sync();
  ctrl_fence();
  pinv_sym4(cs.S4, 1e-3, P4, noshort);
  ctrl_fence();  // pseudoinverse(Jeq*Hinv*Jeq', 1e-3)  (Cassie2d.cpp:134, OSC_RBDL.cpp:171)
#pragma unroll
  for (int r = 0; r < 4; r++) g4[r] = P4[4 * r] * cs.acc[0] + P4[4 * r + 1] * cs.acc[1] + P4[4 * r + 2] * cs.acc[2] + P4[4 * r + 3] * cs.acc[3];
}

// y = Nc w (+ gamma when add_gamma):  Nc = I - Jeq' P4 Jeq Hinv,  gamma = Jeq' P4 JeqdotQdot
template <class CS>
__device__ __forceinline__ void apply_nc(const CS& cs, const double (&P4)[16], const double (&g4)[4], bool add_gamma, double (&w)[NV]) {
  double t4[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double s = 0;
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; s += cs.JH[r][C] * w[C]; });
    t4[r] = s;
  }
  double s4[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    s4[r] = P4[4 * r] * t4[0] + P4[4 * r + 1] * t4[1] + P4[4 * r + 2] * t4[2] + P4[4 * r + 3] * t4[3];
    if (add_gamma) s4[r] -= g4[r];
  }
  static_for<0, NV>([&](auto cc) {
    constexpr int C = decltype(cc)::value;
    w[C] -= jd<0, C>(cs) * s4[0] + jd<1, C>(cs) * s4[1] + jd<2, C>(cs) * s4[2] + jd<3, C>(cs) * s4[3];
  });
}

// ---------------------------------------------------------------- Cassie2d::StepOsc controller: cs.act[7] -> cs.u[6]
// `wset` (uniform inside a row) carries the QP working set from one call to the next, as qpOASES' hotstart does in the reference
// (OSC_RBDL.cpp:276-280): bits 0..13 = variable i sits on a bound, bits 14..27 = ... on its lower bound, bit 28 = valid; 0 = cold start (all
// friction-cone generators at 0, motors free).  The QP is strictly convex, so the warm start changes the number of active-set
// iterations (typically 8 -> 1 or 2), not the solution.
// BASELINE.md C3 "QP iterations / step": per environment the sum, the maximum and the number of StepOsc calls since the counters were cleared
// (CassieVecQpIterations; [n] sums, [n] maxima, [n] calls; each environment is written by the one row that owns it)
__device__ __forceinline__ void qp_stats_add(unsigned* s, int n, int env, int iters) {
  s[env] += (unsigned)iters;
  if ((unsigned)iters > s[n + env]) s[n + env] = (unsigned)iters;
  s[2 * n + env] += 1u;
}

constexpr unsigned QP_COLD_WSET = 0x3FC0u | (0x3FFFu << 14);
template <class SM, class CS>
__device__ __forceinline__ void ctrl_osc(SM& sm, CS& cs, const LaneConst& c, int l, bool rowok, int rowid, unsigned& wset, bool noshort = false,
                                         PhaseClock* pc = nullptr, int* qp_iterations = nullptr) {
  const int lane = rowok ? l : 63;
  double P4[16], g4[4];
  ctrl_dyn(sm, cs, c, l, rowok, P4, g4, noshort);
  PHASE_MARK(*pc, 1);
  // ---- column lanes: 0..5 motors, 6..13 friction-cone generators of contact sites 2..5, 14 the bias column
  {
    double w[NV];
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; w[C] = 0.0; });
    if (lane < 6) {
      const int dof = cp_act_dof[lane];
      const double gear = cp_act_gear[lane];
      static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; w[C] = dof == C ? gear : 0.0; });
    } else if (lane < 14) {
      const int cidx = (lane - 6) >> 1;
      const double sg = (lane & 1) ? -OSC_MU : OSC_MU;
      static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; w[C] = sg * jd<C>(cs, 6 + 2 * cidx, cidx >> 1) + jd<C>(cs, 7 + 2 * cidx, cidx >> 1); });
    } else if (lane == 14) {
      static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; w[C] = cs.bias[C]; });
    }
    apply_nc(cs, P4, g4, lane == 14, w);  // lane 14: Nc bias + gamma = -ce
    // x = Hinv y ;  T column = A x  (A = controller rows 4..14)
    double x[NV];
    static_for<0, NV>([&](auto rr) { constexpr int R = decltype(rr)::value; x[R] = 0.0; });
    static_for<0, NV>([&](auto rr) {  // one read per entry of the symmetric Hinv; every x[R] still sums its terms in column order
      constexpr int R = decltype(rr)::value;
      static_for<R, NV>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        const double h = sm.minv[SM::hidx(R, C)];
        x[R] += h * w[C];
        if constexpr (C != R) x[C] += h * w[R];
      });
    });
    // the whole column in registers FIRST, stored behind the last read of Hinv / the controller rows: in the packed kernel's LDS layout T lives where
    // those lived (EnvLdsC, cassie_ctrl_g16.hip)
    double tcol[11];
    static_for<0, 11>([&](auto rr) {
      constexpr int r = decltype(rr)::value;
      double s = 0;
      static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; s += jd<4 + r, C>(cs) * x[C]; });
      asm volatile("" : "+v"(s));   // the sum is finished HERE (without the pin the 143 row loads are hoisted above all the sums and spilled: the G loop's lesson)
      tcol[r] = s;
    });
    ctrl_fence();
    if (lane < 15) {
      static_for<0, 11>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        const double s = tcol[r];
        if (lane < 14) cs.T[lane][r] = s;
        else {
          // t0 = A q0 + AdotQdot - xdd, q0 = Hinv ce = -x ; xdd packing of Cassie2d.cpp:185-193
          constexpr int XI = r == 10 ? 6 : (r < 2 ? r : (r < 6 ? 2 + (r & 1) : 4 + (r & 1)));
          const double xdd = cs.act[XI];
          double rowacc = 0.0;
          if constexpr (r < 10) rowacc = cs.acc[4 + r];
          cs.t0[r] = -s + rowacc - xdd;
        }
      });
    }
  }
  lds_sync();
  ctrl_fence();
  PHASE_MARK(*pc, 2);
  // ---- QP data: row i of G and c_i on lane i (< 14)
  double G[NZ], cq = 0.0;
  {
    const int i = lane < NZ ? lane : 0;
    double Ti[11];
#pragma unroll
    for (int r = 0; r < 11; r++) {
      double wr = r < 2 ? OSC_W_COM : (r < 10 ? OSC_W_STANCE : OSC_W_REST);  // all four contacts desired (Cassie2d.cpp:199)
      Ti[r] = 2.0 * wr * cs.T[i][r];
      cq += Ti[r] * cs.t0[r];
    }
    static_for<0, NZ>([&](auto jj) {
      constexpr int Jv = decltype(jj)::value;
      double s = 0;
#pragma unroll
      for (int r = 0; r < 11; r++) s += Ti[r] * cs.T[Jv][r];
      if (Jv >= 6) {
        if (lane == Jv) s += OSC_W_F * (OSC_MU * OSC_MU + 1.0);
        if (lane == (Jv ^ 1) && lane >= 6) s += OSC_W_F * (1.0 - OSC_MU * OSC_MU);
      }
      asm volatile("" : "+v"(s));  // G[Jv] is finished HERE: without the pin the 154 T loads are hoisted above all the sums
      G[Jv] = s;                   // and held in registers (r03: 116 of them spilled, one scratch round trip per reload)
    });
  }
  const bool isvar = rowok && l < NZ;
  const double lo = lane < 6 ? cp_act_ctrlrange[lane < 6 ? lane : 0][0] : 0.0;
  const double hi = lane < 6 ? cp_act_ctrlrange[lane < 6 ? lane : 0][1] : 1.7976931348623157e308;
  // ---- primal active-set iterations; each row runs its own problem, the loop ends when no row is busy
  const unsigned w0 = wset ? wset : QP_COLD_WSET;
  bool bound = isvar && ((w0 >> (l & 15)) & 1u), atlo = (w0 >> (14 + (l & 15))) & 1u;
  if (!isvar) { bound = l >= 6; atlo = true; }
  double z = bound ? (atlo ? lo : hi) : 0.0;  // feasible start: bound variables on their bound, free ones at 0 (inside every box)
  if (l >= 6) z = 0.0;                          // generators have no upper bound
  bool busy = rowok;  // uniform inside a row
  // (active-set iterations of THIS row's QP -- the loop runs until the slowest row of the wavefront is done -- are noted in a spare LDS word when the row
  // converges: a counter register would be the 169th of a kernel sized for three wavefronts per SIMD)
  PHASE_MARK(*pc, 3);
  ctrl_fence();
  for (int it = 0; it < 60; it++) {
    if (__ballot(busy) == 0) break;
#ifdef CASSIE_PHASE_TIMING
    pc->acc[8] += 1;
#endif
    double g = cq;
    static_for<0, NZ>([&](auto jj) { constexpr int Jv = decltype(jj)::value; g += G[Jv] * row_bcast<Jv>(z); });
    const unsigned bm = (unsigned)(__ballot(bound && isvar) >> (16 * rowid)) & 0xFFFFu;
    double Mr[NZ];
    static_for<0, NZ>([&](auto jj) {
      constexpr int Jv = decltype(jj)::value;
      double val = G[Jv];
      if (bound || ((bm >> Jv) & 1)) val = 0.0;
      if (Jv == (l & 15) && (bound || !isvar)) val = 1.0;
      if (!isvar) val = (Jv == (l & 15)) ? 1.0 : 0.0;
      Mr[Jv] = val;
    });
    gauss_jordan_rows<NZ, true>(Mr, l);
    const double rhs = (bound || !isvar) ? 0.0 : -g;
    double d = 0.0;
    static_for<0, NZ>([&](auto jj) { constexpr int Jv = decltype(jj)::value; d += Mr[Jv] * row_bcast<Jv>(rhs); });
    // ratio test
    double t = 2.0;
    if (isvar && !bound) {
      if (d > 0 && l < 6) t = (hi - z) / d;
      else if (d < 0) t = (lo - z) / d;
      if (t < 0) t = 0;
    }
    const double tmin = row_min(t);
    const bool blocked = tmin < 1.0;
    const double alpha = blocked ? tmin : 1.0;
    if (busy) z += alpha * d;
    {
      const unsigned hit = (unsigned)(__ballot(busy && blocked && isvar && !bound && t == tmin) >> (16 * rowid)) & 0xFFFFu;
      const int blk = __ffs(hit) - 1;
      if (busy && blocked && l == blk) { bound = true; atlo = d < 0; z = atlo ? lo : hi; }
    }
    // full step: multipliers of the working set at the new point (evaluated for every row, used by the unblocked ones)
    double g2 = cq;
    static_for<0, NZ>([&](auto jj) { constexpr int Jv = decltype(jj)::value; g2 += G[Jv] * row_bcast<Jv>(z); });
    const double viol = (bound && isvar) ? (atlo ? -g2 : g2) : -1.0e300;
    const double vmax = row_max(viol);
    const bool full = busy && !blocked;
    const bool conv = full && vmax <= 1e-9;
    {
      const unsigned hit = (unsigned)(__ballot(full && !conv && isvar && bound && viol == vmax) >> (16 * rowid)) & 0xFFFFu;
      const int rel = __ffs(hit) - 1;
      if (full && !conv && l == rel) bound = false;
    }
    if (conv) { busy = false; if (l == 15) cs.u[7] = (double)(it + 1); }
#ifdef CASSIE_PHASE_TIMING
    if (it == 59) {  // environments that leave the loop unconverged, bucketed by the size of their last multiplier violation
      const bool un = busy && l == 0;
      pc->acc[9] += (unsigned)__popcll(__ballot(un));
      pc->acc[10] += (unsigned)__popcll(__ballot(un && vmax < 1e-8));
      pc->acc[11] += (unsigned)__popcll(__ballot(un && vmax >= 1e-8 && vmax < 1e-6));
      pc->acc[12] += (unsigned)__popcll(__ballot(un && vmax >= 1e-6 && vmax < 1e-4));
      pc->acc[13] += (unsigned)__popcll(__ballot(un && vmax >= 1e-4 && vmax < 1e-2));
      pc->acc[14] += (unsigned)__popcll(__ballot(un && vmax >= 1e-2));
      pc->acc[15] += (unsigned)__popcll(__ballot(un && blocked));
    }
#endif
  }
  {
    const unsigned bm = (unsigned)(__ballot(bound && isvar) >> (16 * rowid)) & 0x3FFFu;
    const unsigned am = (unsigned)(__ballot(bound && isvar && atlo) >> (16 * rowid)) & 0x3FFFu;
    if (rowok) wset = bm | (am << 14) | (1u << 28);  // bit 28: "a working set is stored" -- an EMPTY set (nothing on a bound) is a
                                                      // perfectly good hot start and must not read as 0 = cold (r02: it did, and
                                                      // every call with all feet loaded re-freed the 8 generators one by one)
  }
  if (lane < 6) cs.u[lane] = z;
  if (busy && l == 15) cs.u[7] = 60.0;   // left the loop unconverged
  lds_sync();
  ctrl_fence();
  if (qp_iterations) *qp_iterations = (int)cs.u[7];
  PHASE_MARK(*pc, 4);
}

// ---------------------------------------------------------------- Cassie2d::StepJacobian controller: cs.act[6] -> cs.u[6]
template <class SM, class CS>
__device__ __forceinline__ void ctrl_jacobian(SM& sm, CS& cs, const LaneConst& c, int l, bool rowok, int rowid, double* dbg = nullptr, bool noshort = false) {
  const int lane = rowok ? l : 63;
  (void)rowid;
  double P4[16], g4[4];
  ctrl_dyn(sm, cs, c, l, rowok, P4, g4, noshort);
  if (dbg && lane == 0) {
    for (int i = 0; i < 13; i++) dbg[97 + i] = cs.bias[i];
    for (int i = 0; i < 4; i++) dbg[110 + i] = g4[i];
    for (int i = 0; i < 16; i++) dbg[114 + i] = P4[i];
    for (int i = 0; i < NCR * NV; i++) {
      const int r = i / NV, C = i % NV, lg = C < 3 ? -1 : (C - 3) / 5;
      dbg[130 + i] = C < 3 ? cs.Jc[r][C] : (lg == ctrl_row_leg(r) ? cs.Jc[r][3 + (C - 3) % 5] : 0.0);
    }
    for (int i = 0; i < NV * NV; i++) dbg[325 + i] = sm.minv[SM::hidx(i / NV, i % NV)];  // Hinv (RBDL semantics): inverse of DynamicState's M
    for (int i = 0; i < NCR; i++) dbg[494 + i] = cs.acc[i];        // JdotQdot of every controller row (rows 0..3: JeqdotQdot)
  }
  // Jc6' f on the dof lanes: per foot the mean of the two 6-D site Jacobians; f = (My, Fx, Fz) (Cassie2d.cpp:139-163)
  if (c.dvalid && c.grp == 0 && rowok) {
    double jtf = 0.0;
#pragma unroll
    for (int foot = 0; foot < 2; foot++) {
      double Fx = cs.act[3 * foot + 0], Fz = cs.act[3 * foot + 1], My = cs.act[3 * foot + 2];
      // controller rows 6..9 (left foot sites 2,3) / 10..13 (right foot sites 4,5)
      int r0 = 6 + 4 * foot;
      const int kc = foot == 0 ? c.kL : c.kR;  // compact column of this dof in a row of that foot's leg (-1: none)
      const int kk = kc < 0 ? 0 : kc;
      double jx = 0.5 * (cs.Jc[r0][kk] + cs.Jc[r0 + 2][kk]), jz = 0.5 * (cs.Jc[r0 + 1][kk] + cs.Jc[r0 + 3][kk]);
      if (kc < 0) { jx = 0.0; jz = 0.0; }
      // angular Jacobian about +y of the toe link: sigma_d for every hinge on its path
      int toe = foot == 0 ? 4 : 9;
      int pm = cp_link_pathmask8[toe];
      int k = foot == 0 ? c.kL : c.kR;
      double jw = (k >= 2 && ((pm >> k) & 1)) ? (c.d == 2 ? 1.0 : -1.0) : 0.0;  // sigma of the hinge
      jtf += jx * Fx + jz * Fz + jw * My;
    }
    cs.y[c.d] = cs.bias[c.d] - jtf;
  }
  lds_sync();
  ctrl_fence();
  {
    double w[NV];
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; w[C] = 0.0; });
    if (lane < 6) {
      const int dof = cp_act_dof[lane];
      const double gear = cp_act_gear[lane];
      static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; w[C] = dof == C ? gear : 0.0; });
    } else if (lane == 6) {
      static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; w[C] = cs.y[C]; });
    }
    apply_nc(cs, P4, g4, lane == 6, w);
    lds_sync();
  ctrl_fence();
    if (lane < 6) { static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; cs.U[lane][C] = w[C]; }); }
    if (lane == 6) { static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; cs.y[C] = w[C]; }); }
  }
  lds_sync();
  ctrl_fence();
  // ---- u = pseudoinverse(Nc Bt, 1e-4) * rhs; row r of U = Nc Bt (13x6) on lane r
  // Fast path: U has full column rank in every reachable pose (singular values ~12 .. 100 against the 1e-4 threshold), and then
  // U^+ = (U'U)^-1 U'.  N = U'U is 6x6 and well conditioned (cond ~ 70), so the normal equations are safe in double precision;
  // lambda_min(N) >= 1/||N^-1||_F, hence ||N^-1||_F * tol^2 < 1 (with positive pivots) certifies sigma_min(U) > tol.  The
  // one-sided Jacobi SVD (~10 k instructions) only runs for rows whose certificate fails.
  const bool rowU = lane < NV, rowV = lane < 6;
  double Ur[6], Vr[6];
#pragma unroll
  for (int k = 0; k < 6; k++) { Ur[k] = rowU ? cs.U[k][lane < NV ? lane : 0] : 0.0; Vr[k] = (rowV && lane == k) ? 1.0 : 0.0; }
  const double rhs = rowU ? cs.y[lane < NV ? lane : 0] : 0.0;
  double u = 0.0;
  bool certified = false;
  {
    double N[6][6], yk[6];
    static_for<0, 6>([&](auto ii) {
      constexpr int I = decltype(ii)::value;
      yk[I] = row_sum(Ur[I] * rhs);
      static_for<I, 6>([&](auto jj) { constexpr int J = decltype(jj)::value; N[I][J] = row_sum(Ur[I] * Ur[J]); N[J][I] = N[I][J]; });
    });
    bool posdef = true;
#pragma unroll
    for (int k = 0; k < 6; k++) {  // in-place Gauss-Jordan inverse (values are uniform inside a row)
      const double piv = N[k][k];
      posdef = posdef && piv > 0.0;
      const double inv = 1.0 / piv;
#pragma unroll
      for (int j = 0; j < 6; j++) if (j != k) N[k][j] *= inv;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        if (i == k) continue;
        const double t = N[i][k];
#pragma unroll
        for (int j = 0; j < 6; j++) if (j != k) N[i][j] -= t * N[k][j];
        N[i][k] = -t * inv;
      }
      N[k][k] = inv;
    }
    double fro2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j < 6; j++) fro2 += N[i][j] * N[i][j];
    const double tol2 = 1e-4 * 1e-4;
    certified = !noshort && posdef && fro2 * tol2 * tol2 < 1.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      double uk = 0.0;
#pragma unroll
      for (int j = 0; j < 6; j++) uk += 0.5 * (N[k][j] + N[j][k]) * yk[j];
      if (lane == k) u = uk;
    }
  }
  bool busy = rowok && !certified;
  if (__ballot(busy) != 0) {  // fallback: one-sided Jacobi SVD with the singular-value threshold applied literally
    for (int sweep = 0; sweep < 30; sweep++) {
      if (__ballot(busy) == 0) break;
      double off = 0.0;  // uniform inside a row
      static_for<0, 5>([&](auto pp) {
        constexpr int Pp = decltype(pp)::value;
        static_for<Pp + 1, 6>([&](auto qq) {
          constexpr int Q = decltype(qq)::value;
          double a = row_sum(Ur[Pp] * Ur[Pp]), b = row_sum(Ur[Q] * Ur[Q]), cc = row_sum(Ur[Pp] * Ur[Q]);
          const bool rot = busy && fabs(cc) > 1e-300 && fabs(cc) > 1e-17 * sqrt(a * b);
          // rotation parameters (computed on every lane, applied where `rot`): no DPP below this point of the pair
          double ab = sqrt(a * b);
          double zeta = (b - a) / (2.0 * (rot ? cc : 1.0));
          double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          double cs_ = 1.0 / sqrt(1.0 + t * t), sn = cs_ * t;
          cs_ = rot ? cs_ : 1.0; sn = rot ? sn : 0.0;
          off += rot ? fabs(cc) / ab : 0.0;
          double up = Ur[Pp], uq = Ur[Q];
          Ur[Pp] = cs_ * up - sn * uq; Ur[Q] = sn * up + cs_ * uq;
          double vp = Vr[Pp], vq = Vr[Q];
          Vr[Pp] = cs_ * vp - sn * vq; Vr[Q] = sn * vp + cs_ * vq;
        });
      });
      if (off < 1e-15) busy = false;
    }
    double us = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      double s2 = row_sum(Ur[k] * Ur[k]);
      double pr = row_sum(Ur[k] * rhs);
      double coef = sqrt(s2) > 1e-4 ? pr / s2 : 0.0;
      us += Vr[k] * coef;
    }
    if (!certified) u = us;
  }
  if (lane < 6) cs.u[lane] = u;
  lds_sync();
  ctrl_fence();
  if (dbg && lane == 0) {
    for (int i = 0; i < 13; i++) dbg[i] = cs.y[i];
    for (int i = 0; i < 78; i++) dbg[13 + i] = cs.U[i / NV][i % NV];
    for (int i = 0; i < 6; i++) dbg[91 + i] = cs.u[i];
  }
}

// ---------------------------------------------------------------- scripted standing controllers (cassie2d.py:263-331)
// Reads the operational-space state exactly as the Python does (GetOperationalSpaceState before the step: stale kinematics).
template <int CTRL, class SM, class CS>
__device__ __forceinline__ void scripted_targets(SM& sm, CS& cs, const LaneConst& c, int l, bool rowok, bool fix_kin, double zpos, double zvel) {
  opstate18(sm, c, rowok ? l : 63, fix_kin, cs.s18);
  if (rowok && l == 0) {
    const double* s = cs.s18;  // body_x 0..2, body_xd 3..5, left_x 6..8, left_xd 9..11, right_x 12..14, right_xd 15..17
    if (CTRL == 2) {
      const double stance_kp = 100.0, com_kp = 100.0, com_kd = 20.0, pitch_kp = 20.0, pitch_kd = 10.0;
      double xpos_target = (s[6] + s[12]) / 2.0;
      cs.act[0] = com_kp * (xpos_target - s[0]) + com_kd * (0.0 - s[3]);
      cs.act[1] = com_kp * (zpos - s[1]) + com_kd * (zvel - s[4]);
      cs.act[2] = 0.0; cs.act[3] = stance_kp * (-5e-3 - s[7]);
      cs.act[4] = 0.0; cs.act[5] = stance_kp * (-5e-3 - s[13]);
      cs.act[6] = pitch_kp * (0.0 - s[2]) + pitch_kd * (0.0 - s[5]);
    } else {
      const double com_kp = 200.0, com_kd = 50.0, pitch_kp = 100.0, pitch_kd = 10.0;
      double xpos_target = (s[6] + s[12]) / 2.0;
      double fx = com_kp * (xpos_target - s[0]) + com_kd * (0.0 - s[3]);
      double fz = 0.5 * 9.806 * 31.0 + com_kp * (zpos - s[1]) + com_kd * (zvel - s[4]);
      double my = pitch_kp * (0.0 - s[2]) + pitch_kd * (0.0 - s[5]);
      if (fz < 0.0) fz = 0.0;
      cs.act[0] = fx; cs.act[1] = fz; cs.act[2] = my;
      cs.act[3] = fx; cs.act[4] = fz; cs.act[5] = my;
    }
  }
  lds_sync();
  ctrl_fence();
}

// ---------------------------------------------------------------- controller kernel, one wavefront per environment
// CTRL: 2 = OSC (StepOsc), 3 = Jacobian (StepJacobian).  SCRIPTED: targets come from standing_controller_* instead of actions.
// DynamicModel::setState + DynamicState + controller; the motor commands go into the state record and env_step_kernel<2, ..>
// (cassie_kernels.hip) does the mj_step -- the same split as the packed path (cassie_ctrl_g16.hip), kept as the independent
// cross-check of that path (CASSIE_WAVE_PER_ENV) and for the debug record of the Jacobian controller.
template <int CTRL, bool SCRIPTED>
__global__ void __launch_bounds__(64, 1) env_ctrl_kernel(VecParams p, const double* zpos, const double* zvel) {
  __shared__ Smem sm;
  __shared__ CtrlSmem cs;
  const int env = blockIdx.x;
  const int lane = threadIdx.x;
  if (env >= p.n_envs) return;
  double* st = p.state + (size_t)env * ENV_STRIDE;
  LaneConst c;
  load_lane_const(c, lane);
  load_state(st, sm, lane);
  constexpr int ADIM = CTRL == 2 ? 7 : 6;
  if (!SCRIPTED && lane < ADIM) cs.act[lane] = p.actions[(size_t)env * ADIM + lane];
  lds_sync();
  ctrl_fence();
  const bool fix_kin = (p.flags & FLAG_FIX_STALE_KIN) != 0, noshort = (p.flags & FLAG_NO_PINV_SHORTCUT) != 0;
  unsigned wset = (unsigned)st[ES_QPWSET];
  if (SCRIPTED) scripted_targets<CTRL>(sm, cs, c, lane, lane < 16, fix_kin, zpos[env], zvel[env]);  // kinematics of the LAST setState
  if (lane < 13) { st[ES_KQ + lane] = sm.q[lane]; st[ES_KV + lane] = sm.v[lane]; }  // DynamicModel::setState
  lds_sync();
  ctrl_fence();
  int qpit = 0;
  if (CTRL == 2) ctrl_osc(sm, cs, c, lane, lane < 16, lane >> 4, wset, noshort, nullptr, &qpit);
  else ctrl_jacobian(sm, cs, c, lane, lane < 16, lane >> 4, p.debug ? p.debug + (size_t)env * DBG_STRIDE : nullptr, noshort);
  if (lane < NU) st[ES_CTRL + lane] = cs.u[lane];  // mj_data->ctrl (pre-clamp), consumed by the physics kernel
  if (CTRL == 2 && lane == 0) st[ES_QPWSET] = (double)wset;
  if (CTRL == 2 && lane == 0 && p.qp_stats) qp_stats_add(p.qp_stats, p.n_envs, env, qpit);
}

}  // namespace cassie
#endif
