// cassie_leg_core.h -- Cassie2d Env.step with TWO LANES PER ENVIRONMENT: one lane per leg, 32 environments per wavefront.
//
// Why (r02 PMC + phase profile): the 4-environments-per-wave kernel (cassie_kernels_g16.hip) is
// VALU-issue bound, and in its PGS sweeps -- 60 % of its time -- a Gauss-Seidel step is a scalar computation replicated on the
// 16 lanes of an environment's DPP row: 4 useful lanes of 64.  Only fewer instructions per environment help.  This file is
// the structural answer: every phase of the substep is SCALAR-PER-LANE code over ONE LEG of the robot, so a wave instruction
// does useful work for 64 (environment, leg) pairs; the two lanes of an environment meet only where the mechanism couples
// its legs -- through the three base dofs (x, z, pitch).
//
//   * Mass matrix M = [[B, C_L', C_R'], [C_L, L_L, 0], [C_R, 0, L_R]]: each lane builds and inverts its own 5x5 leg block
//     (hip, knee, ankle, toe, achilles rod), forms Y = L^-1 C and its share C' L^-1 C of the 3x3 Schur complement
//     S = B - sum_k C_k' L_k^-1 C_k; the shares meet by one lane-pair exchange, both lanes factor S = F F' and keep G = F^-1.
//     M^-1 is never formed: x_b = G'G (g_b - sum_k Y_k' g_k), x_k = L_k^-1 g_k - Y_k x_b.
//   * Constraint rows belong to a leg (connect, joint limits, the leg's collision spheres; the pelvis sphere rides on the left
//     lane).  With z_i = L^-1 jl_i and u~_i = G (jb_i - C' z_i):  A_ij = jl_i . z_j [same leg] + u~_i . u~_j  (+ R_i on the
//     diagonal), so a lane keeps only its leg's symmetric block (36 doubles for 8 rows) and three numbers per row; the coupling
//     to the other leg lives in ONE shared 3-vector a~ = sum_j u~_j f_j, updated by a lane-pair exchange per Gauss-Seidel step.
//   * PGS runs in MuJoCo's row order (connect L, connect R, limits L, limits R, pelvis + left contacts, right contacts): a step
//     is executed by both lanes on their own row slot, the owner leg's result is kept.  ~16 wave instructions per
//     environment-sweep against ~46 in the 16-lanes-per-environment kernel; the setup phases are ~12x shorter.
//   * Capacity: 8 rows per leg (2 connect + contact pairs from slot 2 upward + joint limits from slot 7 downward).  An
//     environment that needs more is left untouched from that substep on and handed to the packed 16-row kernel and, behind
//     it, the wave-per-environment kernel through `pending` (same mechanism as before, one tier more).
//
// The arithmetic is the same restatement of mj_step as in cassie_kernels.hip (reference call sites: Cassie2d::Step/StepPd,
// src/Cassie2d/Cassie2d.cpp:86-117; mj_step of MuJoCo 1.50 configured by model/cassie2d_stiff.xml:5; Cassie2dEnv.step,
// rllab/envs/cassie2d.py:97-225, cassie_stand2d.py:86-137); what differs is the factorisation (block elimination instead of a
// 13x13 Gauss-Jordan) and the grouping of sums, i.e. roundings at the 1e-16 level.
// NOT bit-faithful to MuJoCo's / the oracle's operation order in three places, all at the last-bits level and all inside the tolerance the parity
// tests state (teacher-forced 1e-9, measured 3e-13 .. 7e-12): (1) the impedance uses x * (1 / width) where mj_makeImpedance divides (1 / width is formed
// once per row kind and substep); (2) the middle-zone warm start of a cone uses D * (1 / cone) where mj_constraintUpdate divides; (3) Jacobian terms
// that are STRUCTURALLY zero (a connect row has no entry on the toe dof, a contact / limit row none on the rod) are left out of the row construction
// instead of adding 0 x value -- so a non-finite velocity or warm start no longer reaches every row as 0 x Inf = NaN; it reaches the failure guard
// through the state it produces (tests/test_gpu_fullsize.py::test_failure_guard_in_the_lane_per_leg_tiers).
//
// The code is written against a small "backend" B (per-lane types D/I/M, lane-pair exchange, table gathers, per-lane LDS
// slots) so that the SAME source is compiled (a) by hipcc with B = the gfx950 backend of cassie_kernels_leg.hip -- the
// product -- and (b) by g++ with a lane emulation (oracle/leg_host/) that the CPU test-suite checks against the oracle.
// (b) is test infrastructure: the library has no CPU path.
#ifndef CASSIE_LEG_CORE_H_
#define CASSIE_LEG_CORE_H_

#include "cassie2d_planar.h"
#include "cassie2d_legk.h"
#include "cassie_vec_layout.h"
#include "cassie_terrain.h"

#ifndef LEG_FN
#define LEG_FN __device__ __forceinline__
#endif
#ifndef LEG_FP_CONTRACT_OFF
#define LEG_FP_CONTRACT_OFF _Pragma("clang fp contract(off)")
#endif

namespace cassie {
namespace leg {

constexpr int LNV = CP_NV;
constexpr double LH = CP_TIMESTEP;
constexpr double LMINVAL = 1e-15;
constexpr int CAP = 8;  // constraint rows per leg
#define LEG_NPAIR_SLOTS 3   // contact-pair descriptor slots per lane
#ifndef LEG_STAT_SMALL   // instrumented CPU builds only (tools/small_stats.py): how often a wavefront's group takes the eight-row sweep, and why
#define LEG_STAT_SMALL(small, go, nlim, ncon)
#endif
#ifndef LEG_ITERS
#define LEG_ITERS CP_ITERATIONS   // (timing experiments only: -DLEG_ITERS=n)
#endif
enum { K_NONE = 0, K_EQ = 1, K_LIM = 2, K_CN = 3, K_CT = 4 };

template <int I_> struct LI { static constexpr int value = I_; };
template <int B_, int E_, class F> LEG_FN void lfor(F&& f) {
  if constexpr (B_ < E_) { f(LI<B_>{}); lfor<B_ + 1, E_>(f); }
}
// index of (i, j), i <= j, in the packed upper triangle of a symmetric N x N matrix
constexpr int symidx(int n, int i, int j) { return i <= j ? i * n - i * (i - 1) / 2 + (j - i) : j * n - j * (j - 1) / 2 + (i - j); }

// What the core reads from the launch parameters (scalars / read-only tables; no per-environment pointers).
struct EnvCfg {
  int n_sub, flags, env_kind, auto_reset, adim;
  bool want_obs;
  // a launch that does only a SEGMENT of the Env.step's substeps (cassie_cabi.hip: launch_physics_tiers while robots are down):
  // cont = not the first segment (the iteration counter of the record carries on), pend_extra = substeps of the later segments
  // (a handed-over environment is finished to the end of the Env.step by the lower tiers)
  int pend_extra = 0;
  bool cont = false;
  const double* traj_qpos;
  double traj_tmax;
  int traj_n;
};

template <class B> struct Core {
  typedef typename B::D D;
  typedef typename B::I I;
  typedef typename B::M M;

  // ------------------------------------------------------------------------------------------------ per-lane state
  // Hot part (registers): what every phase of a substep touches.  qpos / qvel / qacc_warmstart: base (x, z, pitch) replicated on
  // both lanes of an environment, own leg (hip, knee, ankle, toe, rod).
  struct Lane {
    D qb[3], ql[5], vb[3], vl[5];
    D wb[3], wl[5];
  };
  // Cold part (per-lane LDS slots, B::Lds::cld / cst): values that live across all substeps of an Env.step but are touched
  // once per substep or once per step -- in registers they would be the allocator's first spills (cf. EnvLds of the g16 kernel).
  enum {
    C_KQ = 0,     // 8: qpos at the last DynamicModel::setState (base 3 + leg 5; quirks Q1/Q2)
    C_KV = 8,     // 8: qvel at the last setState
    C_QST = 16,   // 5: self.qstate of the own leg's joints (quirk Q3); the base entries are never read
    C_CTRL = 21,  // 3: last mj_data->ctrl of the own leg's actuators (hip, knee, toe)
    C_TIME = 24,  // env clock
    C_ACT = 25,   // 3: this step's action components of the own leg's actuators
    C_A2 = 28,    // this lane's share of sum(action^2) (cassie_stand2d.py reward)
    // per-substep values that are produced before the solve and consumed after it (parked here across the PGS sweeps)
    C_TAUB = 29,  // 3: smooth generalised force, base dofs
    C_TAUL = 32,  // 5: ... own leg
    C_OX = 37,    // 6: link origins (Kin numbering) relative to the pelvis origin, x
    C_OZ = 43,    // 6: ... z
    C_N = 49
  };
  struct Out {
    M do_reset, bad, set_state;
    I pend, niter;
  };

  // kinematics of one leg (+ pelvis): index 0 pelvis, 1 thigh, 2 shin, 3 tarsus, 4 toe, 5 achilles rod
  struct Kin {
    D c[6], s[6], w[6], ox[6], oz[6], cx[6], cz[6], fx[6], fz[6], vx[6], vz[6];
  };

  static LEG_FN D ldc(const double* t, I i) { return B::ldc(t, i); }
  // model constant IDX of the lane's leg from the packed per-leg row (cassie2d_legk.h): base(leg) + a compile-time offset
  typedef typename B::K KP_;
  static LEG_FN D kc(KP_ K, int idx) { return B::kld(K, idx); }

  // parent of leg link j (1..5) in the Kin numbering
  static constexpr int kparent(int j) { return j == 1 ? 0 : (j == 5 ? 1 : j - 1); }

  // ------------------------------------------------------------------------------------------------ planar FK of one leg
  template <int SEM>
  static LEG_FN void fk(const D (&qb)[3], const D (&ql)[5], const D (&vb)[3], const D (&vl)[5], I leg, KP_ K, Kin& k) {   // K is read for SEM 0 only
    const I lb = leg * 5 + 1;      // first link of the leg in the model tables
    const I db = leg * 5 + 3;      // first dof of the leg
    D th[6];
    th[0] = cp_link_sigma[0] * (qb[2] - cp_qpos0[2]);
    k.w[0] = cp_link_sigma[0] * vb[2];
    lfor<1, 6>([&](auto jj) {
      constexpr int J = decltype(jj)::value;
      constexpr int P = kparent(J);
      const D sg = SEM == 0 ? kc(K, LK_LINK_SIGMA + J - 1) : ldc(cp_link_sigma, lb + (J - 1));
      th[J] = th[P] + sg * (ql[J - 1] - (SEM == 0 ? kc(K, LK_QPOS0 + J - 1) : ldc(cp_qpos0, db + (J - 1))));
      k.w[J] = k.w[P] + sg * vl[J - 1];
    });
    lfor<0, 6>([&](auto jj) { constexpr int J = decltype(jj)::value; B::sincos(th[J], k.s[J], k.c[J]); });
    k.ox[0] = 0.0; k.oz[0] = 0.0; k.vx[0] = 0.0; k.vz[0] = 0.0;
    D ax[6], az[6];
    ax[0] = 0.0; az[0] = CP_GRAVITY;
    lfor<1, 6>([&](auto jj) {
      constexpr int J = decltype(jj)::value;
      constexpr int P = kparent(J);
      const I li = (lb + (J - 1)) * 2 + SEM * (CP_NLINK * 2);
      const D fx = SEM == 0 ? kc(K, LK_LINK_OFF + 2 * (J - 1)) : ldc(&cp_link_off[0][0][0], li), fz = SEM == 0 ? kc(K, LK_LINK_OFF + 2 * (J - 1) + 1) : ldc(&cp_link_off[0][0][0], li + 1);
      const D tx = k.c[P] * fx + k.s[P] * fz, tz = -k.s[P] * fx + k.c[P] * fz;
      const D pw = k.w[P];
      k.ox[J] = k.ox[P] + tx; k.oz[J] = k.oz[P] + tz;
      k.vx[J] = k.vx[P] + pw * tz; k.vz[J] = k.vz[P] - pw * tx;
      ax[J] = ax[P] - pw * pw * tx; az[J] = az[P] - pw * pw * tz;
    });
    lfor<0, 6>([&](auto jj) {
      constexpr int J = decltype(jj)::value;
      const I l = J == 0 ? I(0) : lb + (J - 1);
      const I ci = l * 2 + SEM * (CP_NLINK * 2);
      const D cx0 = SEM == 0 ? kc(K, LK_LINK_COM + 2 * J) : ldc(&cp_link_com[0][0][0], ci), cz0 = SEM == 0 ? kc(K, LK_LINK_COM + 2 * J + 1) : ldc(&cp_link_com[0][0][0], ci + 1);
      const D m = SEM == 0 ? kc(K, LK_LINK_MASS + J) : ldc(cp_link_mass, l);
      const D rx = k.c[J] * cx0 + k.s[J] * cz0, rz = -k.s[J] * cx0 + k.c[J] * cz0;
      k.cx[J] = k.ox[J] + rx; k.cz[J] = k.oz[J] + rz;
      const D w2 = k.w[J] * k.w[J];
      k.fx[J] = m * (ax[J] - w2 * rx); k.fz[J] = m * (az[J] - w2 * rz);
    });
  }

  // point on Kin link J (relative to the pelvis origin)
  template <int J> static LEG_FN void link_point(const Kin& k, D dx, D dz, D& px, D& pz) {
    px = k.ox[J] + k.c[J] * dx + k.s[J] * dz;
    pz = k.oz[J] - k.s[J] * dx + k.c[J] * dz;
  }

  // ------------------------------------------------------------------------------------------------ mass matrix blocks, bias
  // Ls: own 5x5 leg block (packed symmetric, armature included); C[d][b]: coupling of leg dof d with base dof b;
  // Bb: 3x3 base block (packed symmetric, identical on both lanes); bias: base (identical on both lanes) and leg parts.
  struct Mass { D Ls[15], C[5][3], Bb[6], biasb[3], biasl[5]; };

  // subtree of leg dof d (0 hip .. 4 rod) as a bit mask over the Kin links 1..5
  static constexpr int submask(int d) { return d == 0 ? 0b111110 : (d == 1 ? 0b011100 : (d == 2 ? 0b011000 : (d == 3 ? 0b010000 : 0b100000))); }
  // dof i is an ancestor-or-self of dof j (leg-local indices)
  static constexpr bool anc(int i, int j) { return i == j || (i == 0) || (i == 1 && (j == 2 || j == 3)) || (i == 2 && j == 3); }

  template <int SEM>
  static LEG_FN void mass_bias(const Kin& k, I leg, KP_ K, Mass& mm) {
    const I lb = leg * 5 + 1, db = leg * 5 + 3;
    D mass[6], inert[6];
    lfor<0, 6>([&](auto jj) {
      constexpr int J = decltype(jj)::value;
      const I l = J == 0 ? I(0) : lb + (J - 1);
      mass[J] = SEM == 0 ? kc(K, LK_LINK_MASS + J) : ldc(cp_link_mass, l);
      inert[J] = SEM == 0 ? kc(K, LK_LINK_INERTIA + J) : ldc(&cp_link_inertia[0][0], l + SEM * CP_NLINK);
    });
    D s1x[5], s1z[5], s2[5], sg[5];
    lfor<0, 5>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      const D odx = k.ox[Dd + 1], odz = k.oz[Dd + 1];
      sg[Dd] = SEM == 0 ? kc(K, LK_DOF_SIGMA + Dd) : ldc(cp_dof_sigma, db + Dd);
      D a1x = 0.0, a1z = 0.0, a2 = 0.0, bs = 0.0;
      lfor<1, 6>([&](auto ll) {
        constexpr int Lk = decltype(ll)::value;
        if constexpr ((submask(Dd) >> Lk) & 1) {
          const D rx = k.cx[Lk] - odx, rz = k.cz[Lk] - odz;
          a1x += mass[Lk] * rx; a1z += mass[Lk] * rz; a2 += mass[Lk] * (rx * rx + rz * rz) + inert[Lk];
          bs += k.fx[Lk] * rz - k.fz[Lk] * rx;
        }
      });
      s1x[Dd] = a1x; s1z[Dd] = a1z; s2[Dd] = a2;
      mm.biasl[Dd] = sg[Dd] * bs;
    });
    // leg block (sigma_i sigma_j = 1 inside a leg in this model, kept for generality)
    lfor<0, 5>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      lfor<Ii, 5>([&](auto jj) {
        constexpr int Jj = decltype(jj)::value;
        D val = 0.0;
        if constexpr (anc(Ii, Jj)) {  // deep = Jj
          val = sg[Ii] * sg[Jj] * (s2[Jj] + (k.ox[Jj + 1] - k.ox[Ii + 1]) * s1x[Jj] + (k.oz[Jj + 1] - k.oz[Ii + 1]) * s1z[Jj]);
        }
        if constexpr (Ii == Jj) val += SEM == 0 ? kc(K, LK_DOF_ARMATURE + Ii) : ldc(cp_dof_armature, db + Ii);
        mm.Ls[symidx(5, Ii, Jj)] = val;
      });
    });
    // coupling with the base: slides (x, z) and pitch (sigma = +1, anchor = pelvis origin)
    lfor<0, 5>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      mm.C[Dd][0] = sg[Dd] * s1z[Dd];
      mm.C[Dd][1] = -(sg[Dd] * s1x[Dd]);
      mm.C[Dd][2] = cp_dof_sigma[2] * sg[Dd] * (s2[Dd] + k.ox[Dd + 1] * s1x[Dd] + k.oz[Dd + 1] * s1z[Dd]);
    });
    // whole-body sums about the pelvis origin: own leg's share, partner's share (exchange), pelvis
    D lm = 0.0, l1x = 0.0, l1z = 0.0, l2 = 0.0, lfx = 0.0, lfz = 0.0, lt = 0.0;
    lfor<1, 6>([&](auto ll) {
      constexpr int Lk = decltype(ll)::value;
      lm += mass[Lk]; l1x += mass[Lk] * k.cx[Lk]; l1z += mass[Lk] * k.cz[Lk];
      l2 += mass[Lk] * (k.cx[Lk] * k.cx[Lk] + k.cz[Lk] * k.cz[Lk]) + inert[Lk];
      lfx += k.fx[Lk]; lfz += k.fz[Lk]; lt += k.fx[Lk] * k.cz[Lk] - k.fz[Lk] * k.cx[Lk];
    });
    const D tm = (lm + B::swap(lm)) + mass[0];
    const D t1x = (l1x + B::swap(l1x)) + mass[0] * k.cx[0], t1z = (l1z + B::swap(l1z)) + mass[0] * k.cz[0];
    const D t2 = (l2 + B::swap(l2)) + (mass[0] * (k.cx[0] * k.cx[0] + k.cz[0] * k.cz[0]) + inert[0]);
    mm.Bb[symidx(3, 0, 0)] = tm; mm.Bb[symidx(3, 0, 1)] = 0.0; mm.Bb[symidx(3, 1, 1)] = tm;
    mm.Bb[symidx(3, 0, 2)] = cp_dof_sigma[2] * t1z; mm.Bb[symidx(3, 1, 2)] = -(cp_dof_sigma[2] * t1x);
    mm.Bb[symidx(3, 2, 2)] = t2 + cp_dof_armature[2];
    mm.biasb[0] = (lfx + B::swap(lfx)) + k.fx[0];
    mm.biasb[1] = (lfz + B::swap(lfz)) + k.fz[0];
    mm.biasb[2] = cp_dof_sigma[2] * ((lt + B::swap(lt)) + (k.fx[0] * k.cz[0] - k.fz[0] * k.cx[0]));
  }

  // ------------------------------------------------------------------------------------------------ block factorisation
  // In-place inverse of a packed symmetric positive definite N x N matrix (symmetric Gauss-Jordan, no pivoting).
  template <int N> static LEG_FN void sym_inverse(D (&a)[N * (N + 1) / 2]) {
    lfor<0, N>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      const D p = B::rcp(a[symidx(N, K, K)]);
      D col[N];
      lfor<0, N>([&](auto ii) { constexpr int Ii = decltype(ii)::value; col[Ii] = a[symidx(N, Ii, K)]; });
      lfor<0, N>([&](auto ii) {
        constexpr int Ii = decltype(ii)::value;
        if constexpr (Ii != K) {
          const D t = col[Ii] * p;
          lfor<Ii, N>([&](auto jj) {
            constexpr int Jj = decltype(jj)::value;
            if constexpr (Jj != K) a[symidx(N, Ii, Jj)] = a[symidx(N, Ii, Jj)] - t * col[Jj];
          });
        }
      });
      lfor<0, N>([&](auto ii) {
        constexpr int Ii = decltype(ii)::value;
        if constexpr (Ii != K) a[symidx(N, Ii, K)] = -(col[Ii] * p);
      });
      a[symidx(N, K, K)] = -p;
    });
    // the sweeps leave -A^-1
    lfor<0, N * (N + 1) / 2>([&](auto ii) { constexpr int Ii = decltype(ii)::value; a[Ii] = -a[Ii]; });
  }

  struct Fact { D Li[15], Y[5][3], G[6]; };  // L^-1 (packed symmetric), Y = L^-1 C, G = F^-1 lower triangular (row-major packed: 00,10,11,20,21,22)

  static LEG_FN void factor(const Mass& mm, Fact& fc) {
    lfor<0, 15>([&](auto ii) { constexpr int Ii = decltype(ii)::value; fc.Li[Ii] = mm.Ls[Ii]; });
    sym_inverse<5>(fc.Li);
    lfor<0, 5>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      lfor<0, 3>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        D a = 0.0;
        lfor<0, 5>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += fc.Li[symidx(5, Ii, Jj)] * mm.C[Jj][Bc]; });
        fc.Y[Ii][Bc] = a;
      });
    });
    D S[6];
    lfor<0, 3>([&](auto aa) {
      constexpr int A_ = decltype(aa)::value;
      lfor<A_, 3>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        D a = 0.0;
        lfor<0, 5>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += mm.C[Jj][A_] * fc.Y[Jj][Bc]; });
        S[symidx(3, A_, Bc)] = mm.Bb[symidx(3, A_, Bc)] - (a + B::swap(a));   // own + partner: commutative, identical on both lanes
      });
    });
    // Cholesky S = F F' (F lower), G = F^-1
    const D f00 = B::sqrt(S[symidx(3, 0, 0)]);
    const D i00 = B::rcp(f00);
    const D f10 = S[symidx(3, 0, 1)] * i00, f20 = S[symidx(3, 0, 2)] * i00;
    const D f11 = B::sqrt(S[symidx(3, 1, 1)] - f10 * f10);
    const D i11 = B::rcp(f11);
    const D f21 = (S[symidx(3, 1, 2)] - f20 * f10) * i11;
    const D f22 = B::sqrt(S[symidx(3, 2, 2)] - f20 * f20 - f21 * f21);
    const D i22 = B::rcp(f22);
    fc.G[0] = i00;
    fc.G[1] = -(f10 * i00) * i11; fc.G[2] = i11;
    fc.G[4] = -(f21 * i11) * i22; fc.G[5] = i22;
    fc.G[3] = -(f20 * fc.G[0] + f21 * fc.G[1]) * i22;
  }
  static LEG_FN void Gmul(const Fact& fc, const D (&u)[3], D (&o)[3]) {
    o[0] = fc.G[0] * u[0];
    o[1] = fc.G[1] * u[0] + fc.G[2] * u[1];
    o[2] = fc.G[3] * u[0] + fc.G[4] * u[1] + fc.G[5] * u[2];
  }
  static LEG_FN void GTmul(const Fact& fc, const D (&t)[3], D (&o)[3]) {
    o[0] = fc.G[0] * t[0] + fc.G[1] * t[1] + fc.G[3] * t[2];
    o[1] = fc.G[2] * t[1] + fc.G[4] * t[2];
    o[2] = fc.G[5] * t[2];
  }
  // x = M^-1 g for g = (gb identical on both lanes, gl own leg)
  static LEG_FN void minv_apply(const Fact& fc, const D (&gb)[3], const D (&gl)[5], D (&xb)[3], D (&xl)[5]) {
    D t[3];
    lfor<0, 3>([&](auto bb) {
      constexpr int Bc = decltype(bb)::value;
      D a = 0.0;
      lfor<0, 5>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += fc.Y[Jj][Bc] * gl[Jj]; });
      t[Bc] = gb[Bc] - (a + B::swap(a));
    });
    D gt[3];
    Gmul(fc, t, gt);
    GTmul(fc, gt, xb);
    lfor<0, 5>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      D a = 0.0;
      lfor<0, 5>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += fc.Li[symidx(5, Ii, Jj)] * gl[Jj]; });
      xl[Ii] = a - (fc.Y[Ii][0] * xb[0] + fc.Y[Ii][1] * xb[1] + fc.Y[Ii][2] * xb[2]);
    });
  }

  // (iw = 1 / width, formed once per row kind and substep: the rows multiply where mj_makeImpedance divides)
  static LEG_FN D impedance(D d0, D d1, D width, D iw, D x) {
    const M flat = (d0 == d1) | (width <= LMINVAL);
    D xx = B::fabs(x * iw);
    D y = B::sel(xx <= 0.5, 2.0 * xx * xx, 1.0 - 2.0 * (1.0 - xx) * (1.0 - xx));
    D r = d0 + y * (d1 - d0);
    r = B::sel(xx >= 1.0, d1, r);
    r = B::sel(xx <= 0.0, d0, r);
    return B::sel(flat, 0.5 * (d0 + d1), r);
  }

  // ------------------------------------------------------------------------------------------------ one mj_forward (+ Euler)
  struct SubOut { I niter; M overflow, go; };
#ifndef LEG_DAMPING_SWEEPS
#define LEG_DAMPING_SWEEPS 12
#endif
  constexpr static int DAMPING_SWEEPS = LEG_DAMPING_SWEEPS;

  // `live` masks environments that must not be touched (identical on the two lanes of an environment).  `integrate` (wave-uniform)
  // = false gives mj_forward only (Cassie2d::Reset) and none of the per-substep bookkeeping.  `from_rec` (wave-uniform): the motor
  // commands are the record's ctrl (MODE 2, and Reset's mj_forward with the stale ctrl); otherwise MODE 0 = PD law on the step's
  // action, MODE 1 = the action itself.  Bookkeeping of a substep that is carried out (cold LDS slots): setState snapshot of the
  // pre-step state, mj_data->ctrl, env clock.
  // What one substep carries from phase to phase (r05: the substep is three functions -- set-up, sweeps, finish -- so that the
  // 64-environments-per-wavefront kernel, cassie_duo_core.h, can run the set-up of two groups of environments, ONE joint sweep with a lane
  // per environment, and the two finishes; the kernels of this file call them back to back: `substep`).
  struct Sub {
    I leg; KP_ K;
    Fact fc;
    D p1x, p1z, p2x, p2z;       // connect anchors
    I nlim, ncon;
    M go;
    bool small;
    // rows: slots 0,1 connect (x, z); contact pair p at slots (2 + 2p, 3 + 2p); limit j at slot 7 - j
    D r[CAP], f[CAP], ut[CAP][3], Al[CAP * (CAP + 1) / 2], Adiag[CAP], Ainv[CAP];
    D Ant[3];
    I kind[CAP];
    D a0, a1, a2;
    I niter;
  };

  // on_small() / on_general(): what the caller does with the rows, called at the END of the set-up inside the branch that built them (six slots /
  // eight): the two cases never meet again before their rows are consumed, so nothing of a row has to survive a merge of the two paths.
  template <int MODE, bool HF = false, class FS, class FG>
  static LEG_FN void sub_setup(typename B::Lds& lds, Lane& st, bool from_rec, M live, bool integrate, SubOut& out, Sub& s, const Terrain* hf, FS&& on_small, FG&& on_general) {
    // Model constants are re-read from the constant tables in every substep through indices the optimiser cannot see through
    // (B::opq / B::zs): otherwise it hoists ~200 loop-invariant table values out of the substep loop and then spills them
    // (the same trap as in cassie_kernels.hip, r01 PMC: scratch traffic at every kernel boundary).
    s.leg = B::opq(B::leg());
    s.K = B::kbase(s.leg);
    // The common configuration -- no joint limit active, at most two contact pairs per leg, in EVERY environment of the wavefront
    // (robots on their feet) -- leaves slots 6 and 7 empty: those two row slots are not built and the sweeps touch six residuals per
    // step instead of eight (no limit steps, no third pair).  Decided per wavefront, once per substep; the arithmetic of an
    // environment is the same either way (the skipped work adds exact zeros to rows that nobody reads).
    // ---- rows: slots 0,1 connect (x, z); contact pair p at slots (2 + 2p, 3 + 2p); limit j at slot 7 - j
    s.nlim = 0; s.ncon = 0;
    const I leg = s.leg;
    const KP_ K = s.K;
    Fact& fc = s.fc;
    D &p1x = s.p1x, &p1z = s.p1z, &p2x = s.p2x, &p2z = s.p2z;
    I &nlim = s.nlim, &ncon = s.ncon;
    M& go = s.go;
    bool& small = s.small;
    D (&r)[CAP] = s.r; D (&f)[CAP] = s.f; D (&ut)[CAP][3] = s.ut; D (&Al)[CAP * (CAP + 1) / 2] = s.Al; D (&Adiag)[CAP] = s.Adiag; D (&Ainv)[CAP] = s.Ainv;
    D (&Ant)[3] = s.Ant;
    I (&kind)[CAP] = s.kind;
    D &a0 = s.a0, &a1 = s.a1, &a2 = s.a2;
    {
      D qsb[3], qsl[5];
      Mass mm;
      D c_damp[5], c_actr[6], c_gear[3];        // table constants of the smooth force, read in the batch of the active-set tests
      D c_eqsol[2], c_eqimp[3], c_eqw, sgl[5];  // ... of the rows, read ahead of the factorisation (sgl: signs of the leg's dofs)
      lds.mark(0);
      // motor commands of the own leg's actuators: hip (dof 0), knee (1), toe (3).  Read FIRST: where the action row / the record's ctrl
      // live in global memory (the 64-environments kernel) the loads' latency passes behind the kinematics instead of being waited for,
      // one after the other, in front of the factorisation.
      D cu[3];
      if (from_rec) {
        lfor<0, 3>([&](auto aa) { constexpr int A_ = decltype(aa)::value; cu[A_] = lds.cld(C_CTRL + A_); });
      } else {
        D act_[3];
        lfor<0, 3>([&](auto aa) { constexpr int A_ = decltype(aa)::value; act_[A_] = lds.cld(C_ACT + A_); });
        lfor<0, 3>([&](auto aa) {
          constexpr int A_ = decltype(aa)::value;
          constexpr int Dd = A_ == 2 ? 3 : A_;
          if constexpr (MODE == 0) cu[A_] = 10.0 * (act_[A_] - st.ql[Dd]) + 5.0 * (0.0 - st.vl[Dd]);
          else cu[A_] = act_[A_];
        });
      }
      {
        Kin k;
        fk<0>(st.qb, st.ql, st.vb, st.vl, leg, K, k);
        lds.mark(1);
        mass_bias<0>(k, leg, K, mm);
        lds.mark(2);
        // The table constants of the active-set tests, in ONE batch ahead of the tests: every masked descriptor store below is a basic
        // block of its own, and a constant loaded inside it is waited for there -- thirteen to sixteen load latencies in a row for a lone
        // wavefront (r05 phase clocks: ~4800 cycles for these 490 instructions).
        D c_rng[8], c_dofw[4], c_sphd[18], c_sphr[9], c_sphw[9], c_sphy[HF ? 9 : 1], c_anch[4];
        lfor<0, 8>([&](auto ii) { constexpr int Ii = decltype(ii)::value; c_rng[Ii] = kc(K, LK_JNT_RANGE + Ii); });
        lfor<0, 4>([&](auto ii) { constexpr int Ii = decltype(ii)::value; c_dofw[Ii] = kc(K, LK_DOF_INVWEIGHT + Ii); });
        lfor<0, 18>([&](auto ii) { constexpr int Ii = decltype(ii)::value; c_sphd[Ii] = kc(K, LK_SPH_D + Ii); });
        lfor<0, 9>([&](auto ii) {
          constexpr int Ii = decltype(ii)::value;
          c_sphr[Ii] = kc(K, LK_SPH_R + Ii); c_sphw[Ii] = kc(K, LK_SPH_INVWEIGHT + Ii);
          if constexpr (HF) c_sphy[Ii] = kc(K, LK_SPH_Y + Ii);
        });
        c_anch[0] = kc(K, LK_EQ_D1); c_anch[1] = kc(K, LK_EQ_D1 + 1); c_anch[2] = kc(K, LK_EQ_D2); c_anch[3] = kc(K, LK_EQ_D2 + 1);
        lfor<0, 5>([&](auto ii) { constexpr int Ii = decltype(ii)::value; c_damp[Ii] = kc(K, LK_DOF_DAMPING + Ii); });   // (for the smooth force, below)
        lfor<0, 6>([&](auto ii) { constexpr int Ii = decltype(ii)::value; c_actr[Ii] = kc(K, LK_ACT_RANGE + Ii); });
        lfor<0, 3>([&](auto ii) { constexpr int Ii = decltype(ii)::value; c_gear[Ii] = kc(K, LK_ACT_GEAR + Ii); });
        B::fence();
        lfor<0, 6>([&](auto jj) { constexpr int J = decltype(jj)::value; lds.cst(C_OX + J, k.ox[J], live | !live); lds.cst(C_OZ + J, k.oz[J], live | !live); });
        // ---- active set.  Limits: leg dofs 0..3 (the rod is unlimited); contacts: pelvis sphere (left lane only) + 8 leg spheres.
        const D basez = st.qb[1] - cp_qpos0[1] + cp_link_off[0][0][1];
        lfor<0, 4>([&](auto jj) {
          constexpr int Jj = decltype(jj)::value;
          const D qd = st.ql[Jj];
          const D lo = c_rng[2 * Jj], hi = c_rng[2 * Jj + 1];
          const D dlo = qd - lo, dhi = hi - qd;
          const M act = (dlo < 0.0) | (dhi < 0.0);
          const D pos = B::sel(dlo < 0.0, dlo, dhi);
          const D sgn = B::sel(dlo < 0.0, D(1.0), D(-1.0));
          lds.st_lim(nlim, pos, sgn, c_dofw[Jj], I(Jj), act & (nlim < 4));
          nlim = nlim + B::toI(act);
        });
        lfor<0, 9>([&](auto cc) {
          constexpr int Cc = decltype(cc)::value;
          constexpr int Lk = Cc == 0 ? 0 : (Cc + 1) / 2;   // Kin link of candidate Cc: pelvis, thigh x2, shin x2, tarsus x2, toe x2
          D cx, cz;
          link_point<Lk>(k, c_sphd[2 * Cc], c_sphd[2 * Cc + 1], cx, cz);
          if constexpr (HF) {
            // height field (terrain_sphere, cassie_kernels.hip): the sphere against the local plane of the cell under its centre;
            // contact frame = (normal (nx, nz), tangent (nz, -nx)), contact point half-way into the penetration along the normal
            const D basex = st.qb[0] - cp_qpos0[0] + cp_link_off[0][0][0];
            const D rad = c_sphr[Cc];
            D dist, nx, nz;
            B::hf_sphere(*hf, basex + cx, c_sphy[Cc], basez + cz, rad, dist, nx, nz);
            M act = dist < 0.0;
            if constexpr (Cc == 0) act = act & (leg == 0);
            const D back = rad + 0.5 * dist;
            lds.st_pair(ncon, cx - nx * back, cz - nz * back, dist, c_sphw[Cc], I(Lk), act & (ncon < 3));
            lds.st_nrm(ncon, nx, act & (ncon < 3));
            ncon = ncon + B::toI(act);
          } else {
          const D dist = basez + cz - c_sphr[Cc];
          M act = dist < 0.0;
          if constexpr (Cc == 0) act = act & (leg == 0);
          // contact point: half-way into the penetration, on the vertical through the sphere centre
          lds.st_pair(ncon, cx, 0.5 * dist - basez, dist, c_sphw[Cc], I(Lk), act & (ncon < 3));
          ncon = ncon + B::toI(act);
          }
        });
        // connect anchors: rod end (Kin link 5) against the heel-spring anchor on the tarsus (Kin link 3)
        link_point<5>(k, c_anch[0], c_anch[1], p1x, p1z);
        link_point<3>(k, c_anch[2], c_anch[3], p2x, p2z);
      }
      B::fence();
      lds.mark(3);
      const I nrows = nlim + ncon * 2 + 2;
      M ovf = live & (nrows > CAP);
      ovf = ovf | B::swapm(ovf);
      out.overflow = ovf;
      go = live & !ovf;
      out.go = go;
      small = !B::any(go & ((nlim > 0) | (ncon > 2)));
      LEG_STAT_SMALL(small, go, nlim, ncon);
      c_eqsol[0] = kc(K, LK_EQ_SOLREF); c_eqsol[1] = kc(K, LK_EQ_SOLREF + 1);
      lfor<0, 3>([&](auto ii) { constexpr int Ii = decltype(ii)::value; c_eqimp[Ii] = kc(K, LK_EQ_SOLIMP + Ii); });
      c_eqw = kc(K, LK_EQ_INVWEIGHT);
      lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; sgl[Dd] = kc(K, LK_DOF_SIGMA + Dd); });
      {
        if (integrate) {
          // DynamicModel::setState of this substep (pre-step state), mj_data->ctrl, env clock -- for environments that carry it out
          lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; lds.cst(C_KQ + Bc, st.qb[Bc], go); lds.cst(C_KV + Bc, st.vb[Bc], go); });
          lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; lds.cst(C_KQ + 3 + Dd, st.ql[Dd], go); lds.cst(C_KV + 3 + Dd, st.vl[Dd], go); });
          if (!from_rec) lfor<0, 3>([&](auto aa) { constexpr int A_ = decltype(aa)::value; lds.cst(C_CTRL + A_, cu[A_], go); });
          lds.cst(C_TIME, lds.cld(C_TIME) + 0.0005, go);
        }
        // smooth force: passive damping, bias, actuation (ctrl clamped to ctrlrange, times gear)
        D taub[3], taul[5];
        lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; taub[Bc] = -mm.biasb[Bc]; lds.cst(C_TAUB + Bc, taub[Bc], live | !live); });
        lfor<0, 5>([&](auto dd) {
          constexpr int Dd = decltype(dd)::value;
          D t = -c_damp[Dd] * st.vl[Dd] - mm.biasl[Dd];
          if constexpr (Dd == 0 || Dd == 1 || Dd == 3) {
            constexpr int A_ = Dd == 3 ? 2 : Dd;
            const D lo = c_actr[2 * A_], hi = c_actr[2 * A_ + 1];
            const D u = B::sel(cu[A_] < lo, lo, B::sel(cu[A_] > hi, hi, cu[A_]));
            t = t + c_gear[A_] * u;
          }
          taul[Dd] = t;
          lds.cst(C_TAUL + Dd, t, live | !live);
        });
        B::fence();
        factor(mm, fc);
        B::fence();
        minv_apply(fc, taub, taul, qsb, qsl);
      }
      B::fence();
      lds.mark(4);
      // The rows are built one slot at a time: a row's z = L^-1 jl is kept (5 doubles per row), its Jacobian is not -- the entry
      // A_ij of the leg block is z_i . jl_j, formed when row j >= i is built -- and its warm-start force is formed as soon as the
      // row (for a contact pair: the tangent row) is complete, so nothing but z, u~, b and the packed A survives a slot.
      // soft-constraint constants of the three row kinds: k = 1 / (dmax^2 tc^2 dampratio^2), b = 2 / (dmax tc), tc >= 2 h; the
      // impedance at position 0 (what a friction row uses for its own reference acceleration)
      struct KindPar { D kk, bb, d0, d1, w, iw, imp0; };
      auto kind_par = [&](D solref0, D solref1, D d0, D d1, D w) {
        KindPar k_;
        const D tc = B::sel(solref0 < 2.0 * LH, D(2.0 * LH), solref0);
        k_.kk = 1.0 / (d1 * d1 * tc * tc * solref1 * solref1); k_.bb = 2.0 / (d1 * tc);
        k_.d0 = d0; k_.d1 = d1; k_.w = w; k_.iw = 1.0 / w; k_.imp0 = impedance(d0, d1, w, k_.iw, D(0.0));
        return k_;
      };
      const KindPar kp_eq = kind_par(c_eqsol[0], c_eqsol[1], c_eqimp[0], c_eqimp[1], c_eqimp[2]);
      const KindPar kp_lim = kind_par(D(cp_limit_solref[0]), D(cp_limit_solref[1]), D(cp_limit_solimp[0]), D(cp_limit_solimp[1]), D(cp_limit_solimp[2]));
      const KindPar kp_con = kind_par(D(cp_contact_solref[0]), D(cp_contact_solref[1]), D(cp_contact_solimp[0]), D(cp_contact_solimp[1]), D(cp_contact_solimp[2]));
      D bvec[CAP], z[CAP][5];
      D jar_prev = 0.0, Rr_prev = 1.0;   // the normal row's values while its tangent row is built
      const D mu = CP_CONTACT_MU;
      auto build_slot = [&](auto ss) {
        constexpr int S = decltype(ss)::value;
        D pos = 0.0, invw = 0.0;
        D jb[3], jl[5];
        I kd = K_NONE;
        lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; jl[Dd] = 0.0; });
        jb[0] = 0.0; jb[1] = 0.0; jb[2] = 0.0;
        if constexpr (S < 2) {
          kd = K_EQ;
          invw = c_eqw;
          pos = S == 0 ? p1x - p2x : p1z - p2z;
          // J = J(p1 on rod: pitch, hip, rod) - J(p2 on tarsus: pitch, hip, knee, ankle); the base slides cancel
          auto ent = [&](D px, D pz, auto jl_) { constexpr int Jl = decltype(jl_)::value; return S == 0 ? pz - lds.cld(C_OZ + Jl) : -(px - lds.cld(C_OX + Jl)); };  // y^ x (p - o), component S
          jb[2] = cp_dof_sigma[2] * (ent(p1x, p1z, LI<0>{}) - ent(p2x, p2z, LI<0>{}));
          const D sg0 = sgl[0], sg1 = sgl[1], sg2 = sgl[2], sg4 = sgl[4];
          jl[0] = sg0 * ent(p1x, p1z, LI<1>{}) - sg0 * ent(p2x, p2z, LI<1>{});
          jl[1] = -(sg1 * ent(p2x, p2z, LI<2>{}));
          jl[2] = -(sg2 * ent(p2x, p2z, LI<3>{}));
          jl[4] = sg4 * ent(p1x, p1z, LI<5>{});
        } else {
          constexpr int P = (S - 2) >> 1;     // contact pair that may live here
          constexpr int ODD = (S - 2) & 1;    // 0 normal (z row), 1 tangent (x row)
          constexpr int LJ = 7 - S;           // limit index that may live here
          const M isc = ncon > P;
          const M isl = nlim > LJ;
          kd = B::seli(isc, I(ODD ? K_CT : K_CN), B::seli(isl, I(K_LIM), I(K_NONE)));
          // contact
          D px, pz, dist, cinvw; I depth;
          lds.ld_pair(P, px, pz, dist, cinvw, depth);
          D cjl[4];
          D cj0 = ODD ? 1.0 : 0.0, cj1 = ODD ? 0.0 : 1.0, cj2;
          if constexpr (HF) {
            // row direction: the normal (nx, nz) or the tangent (nz, -nx) of the local terrain plane
            const D nx = lds.ld_nrm(P), nz = B::sqrt(1.0 - nx * nx);
            const D dx = ODD ? nz : nx, dz = ODD ? -nx : nz;
            lfor<0, 4>([&](auto dd) {
              constexpr int Dd = decltype(dd)::value;
              const D val = sgl[Dd] * (dx * (pz - lds.cld(C_OZ + Dd + 1)) - dz * (px - lds.cld(C_OX + Dd + 1)));
              cjl[Dd] = B::sel(depth > Dd, val, D(0.0));
            });
            cj0 = dx; cj1 = dz;
            cj2 = cp_dof_sigma[2] * (dx * pz - dz * px);
          } else {
          lfor<0, 4>([&](auto dd) {
            constexpr int Dd = decltype(dd)::value;
            const D val = sgl[Dd] * (ODD ? pz - lds.cld(C_OZ + Dd + 1) : -(px - lds.cld(C_OX + Dd + 1)));
            cjl[Dd] = B::sel(depth > Dd, val, D(0.0));
          });
          cj2 = cp_dof_sigma[2] * (ODD ? pz : -px);
          }
          // limit
          D lpos = 0.0, lsgn = 0.0, linvw = 0.0; I lj = 0;
          if constexpr (LJ < 4) lds.ld_lim(LJ, lpos, lsgn, linvw, lj);
          lfor<0, 4>([&](auto dd) {
            constexpr int Dd = decltype(dd)::value;
            jl[Dd] = B::sel(isc, cjl[Dd], B::sel(isl & (lj == Dd), lsgn, D(0.0)));
          });
          jb[0] = B::sel(isc, cj0, D(0.0));
          jb[1] = B::sel(isc, cj1, D(0.0));
          jb[2] = B::sel(isc, cj2, D(0.0));
          pos = B::sel(isc, dist, lpos);
          invw = B::sel(isc, cinvw, linvw);
        }
        kd = B::seli(go, kd, I(K_NONE));
        kind[S] = kd;
        const M active = kd != K_NONE;
        const M islim = kd == K_LIM;
        // solver parameters of the row's kind (formed once per substep, above): slots 0, 1 are connect rows, the others limit or contact
        const D kk_ = S < 2 ? kp_eq.kk : B::sel(islim, kp_lim.kk, kp_con.kk), bb_ = S < 2 ? kp_eq.bb : B::sel(islim, kp_lim.bb, kp_con.bb);
        const D simp0 = S < 2 ? kp_eq.d0 : B::sel(islim, kp_lim.d0, kp_con.d0), simp1 = S < 2 ? kp_eq.d1 : B::sel(islim, kp_lim.d1, kp_con.d1);
        const D simp2 = S < 2 ? kp_eq.w : B::sel(islim, kp_lim.w, kp_con.w), imp_at0 = S < 2 ? kp_eq.imp0 : B::sel(islim, kp_lim.imp0, kp_con.imp0);
        const D simp2i = S < 2 ? kp_eq.iw : B::sel(islim, kp_lim.iw, kp_con.iw);
        D vel = jb[0] * st.vb[0] + jb[1] * st.vb[1] + jb[2] * st.vb[2];
        D bq = jb[0] * qsb[0] + jb[1] * qsb[1] + jb[2] * qsb[2];
        D jw = jb[0] * st.wb[0] + jb[1] * st.wb[1] + jb[2] * st.wb[2];
        // (a connect row has no entry on the toe dof, slot 3 of the leg's dofs; a contact / limit row none on the rod, slot 4: the terms
        // that would add an exact zero are left out, here and in z, u, A below)
        lfor<0, 5>([&](auto dd) {
          constexpr int Dd = decltype(dd)::value;
          if constexpr (S < 2 ? Dd != 3 : Dd != 4) { vel += jl[Dd] * st.vl[Dd]; bq += jl[Dd] * qsl[Dd]; jw += jl[Dd] * st.wl[Dd]; }
        });
        const D imp = impedance(simp0, simp1, simp2, simp2i, pos);
        D R = (1.0 - imp) / imp * invw;
        R = B::sel(R > LMINVAL, R, D(LMINVAL));
        const M ist = kd == K_CT;
        const D own_pos = B::sel(ist, D(0.0), pos);
        const D imp_own = B::sel(ist, imp_at0, imp);   // the tangent row's own position is 0
        const D aref = -bb_ * vel - kk_ * imp_own * own_pos;
        bvec[S] = B::sel(active, bq - aref, D(0.0));
        const D jar = jw - aref;
        const D Rr = B::sel(active, R, D(1.0));
        lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; jl[Dd] = B::sel(active, jl[Dd], D(0.0)); });
        lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; jb[Bc] = B::sel(active, jb[Bc], D(0.0)); });
        // z = L^-1 jl, u = jb - C' z, u~ = G u
        lfor<0, 5>([&](auto ii) {
          constexpr int Ii = decltype(ii)::value;
          D a = 0.0;
          lfor<0, 5>([&](auto jj) { constexpr int Jj = decltype(jj)::value; if constexpr (S < 2 ? Jj != 3 : Jj != 4) a += fc.Li[symidx(5, Ii, Jj)] * jl[Jj]; });
          z[S][Ii] = a;
        });
        D u[3];
        lfor<0, 3>([&](auto bb) {
          constexpr int Bc = decltype(bb)::value;
          D a = 0.0;
          lfor<0, 5>([&](auto jj) { constexpr int Jj = decltype(jj)::value; if constexpr (S < 2 ? Jj != 3 : Jj != 4) a += fc.Y[Jj][Bc] * jl[Jj]; });   // C' z = C' L^-1 jl = Y' jl
          u[Bc] = jb[Bc] - a;
        });
        Gmul(fc, u, ut[S]);
        // column S of the own-leg block of A (R on the diagonal) and the full diagonal entry
        lfor<0, S + 1>([&](auto ii) {
          constexpr int Ii = decltype(ii)::value;
          D a = 0.0;
          lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; if constexpr (S < 2 ? Dd != 3 : Dd != 4) a += z[Ii][Dd] * jl[Dd]; });
          if constexpr (Ii == S) a += Rr;
          Al[symidx(CAP, Ii, S)] = a;
        });
        Adiag[S] = Al[symidx(CAP, S, S)] + (ut[S][0] * ut[S][0] + ut[S][1] * ut[S][1] + ut[S][2] * ut[S][2]);
        Ainv[S] = B::rcp(Adiag[S]);
        if constexpr (S < 2) Adiag[S] = 0.5 * Adiag[S];   // a connect row's step only ever uses A_ii / 2 (the change of cost): halved here, once, exactly
        // warm start (mj_constraintUpdate from qacc_warmstart) of a single row; of a contact pair when its tangent row is complete
        const D Dd_ = B::rcp(Rr);
        D fv = 0.0;
        fv = B::sel(kd == K_EQ, -Dd_ * jar, fv);
        fv = B::sel((kd == K_LIM) & (jar < 0.0), -Dd_ * jar, fv);
        f[S] = fv;
        if constexpr (S >= 3 && ((S - 2) & 1)) {
          constexpr int NS = S - 1, P = (S - 2) >> 1;
          const D jn = jar_prev, jt = jar;
          const D Dn_ = B::rcp(Rr_prev);
          const D Nn = jn * mu, U1 = jt * mu, Tt = B::fabs(U1);
          const M top = (Nn >= mu * Tt) | ((Tt <= 0.0) & (Nn >= 0.0));
          const M bot = (mu * Nn + Tt <= 0.0) | ((Tt <= 0.0) & (Nn < 0.0));
          // middle zone: each row with its own D, as mj_constraintUpdate does (the two are equal: the pair shares its regulariser)
          const D NmT = Nn - mu * Tt;
          const D inv_cone = 1.0 / (mu * mu * (1.0 + mu * mu));   // (a constant of the model: the two divisions of mj_constraintUpdate's middle zone as multiplications)
          const D fnm_n = -(Dn_ * inv_cone) * NmT * mu;
          const D fnm_t = -(Dd_ * inv_cone) * NmT * mu;
          const D ftm = -fnm_t / Tt * U1 * mu;
          const D fn = B::sel(top, D(0.0), B::sel(bot, -Dn_ * jn, fnm_n));
          const D ft = B::sel(top, D(0.0), B::sel(bot, -Dd_ * jt, ftm));
          f[NS] = B::sel(kind[NS] == K_CN, fn, f[NS]);
          f[S] = B::sel(kd == K_CT, ft, f[S]);
          Ant[P] = Al[symidx(CAP, NS, S)] + (ut[NS][0] * ut[S][0] + ut[NS][1] * ut[S][1] + ut[NS][2] * ut[S][2]);
        }
        jar_prev = jar; Rr_prev = Rr;
        B::fence();   // keep the scheduler from interleaving the slots (longer live ranges -> spills)
      };
      lfor<0, 6>([&](auto ss) { build_slot(ss); });
      // warm start of NS slots: a~, the own-leg part of A f, the cost of the warm start (kept only if negative)
      auto warm_start = [&](auto ns_) {
        constexpr int NS = decltype(ns_)::value;
        D at[3] = {D(0.0), D(0.0), D(0.0)};
        lfor<0, NS>([&](auto ss) { constexpr int S = decltype(ss)::value; lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; at[Bc] += ut[S][Bc] * f[S]; }); });
        lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; at[Bc] = at[Bc] + B::swap(at[Bc]); });
        D cost = 0.0;
        lfor<0, NS>([&](auto ii) {
          constexpr int Ii = decltype(ii)::value;
          D a = 0.0;
          lfor<0, NS>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += Al[symidx(CAP, Ii, Jj)] * f[Jj]; });
          r[Ii] = a;   // own-leg part of (A f)_i, R included
          const D full = a + (ut[Ii][0] * at[0] + ut[Ii][1] * at[1] + ut[Ii][2] * at[2]);
          cost += f[Ii] * (0.5 * full + bvec[Ii]);
        });
        cost = cost + B::swap(cost);
        const M drop = cost > 0.0;
        lfor<0, NS>([&](auto ss) {
          constexpr int S = decltype(ss)::value;
          f[S] = B::sel(drop, D(0.0), f[S]);
          r[S] = B::sel(drop, D(0.0), r[S]) + bvec[S];
        });
        a0 = B::sel(drop, D(0.0), at[0]); a1 = B::sel(drop, D(0.0), at[1]); a2 = B::sel(drop, D(0.0), at[2]);
      };
      // Slots 6 and 7 of a wavefront on its feet (`small`: no joint limit, at most two pairs per leg, in EVERY environment) are empty in
      // every lane: they are not built (r05: ~760 of the ~3000 instructions of this phase built exactly zeros and a unit diagonal), and the
      // warm start runs over six slots -- the two left-out slots would add exact zeros to every sum.  Nothing reads their rows in that
      // case (six-row sweeps); the finish reads their kind and force.
      if constexpr (!B::SPLIT_TAIL) {
        // (backends that always build the two empty slots: the CPU lane emulation; until r05 the 64-environments kernel, whose set-up then had the
        // eight-row solve inline behind it)
        build_slot(LI<6>{}); build_slot(LI<7>{});
        B::fence();
        lds.mark(5);
        warm_start(LI<CAP>{});
        B::fence();
        lds.mark(6);
        if (small) on_small(); else on_general();
      } else if (small) {
        B::fence();
        lds.mark(5);
        warm_start(LI<6>{});
        kind[6] = K_NONE; kind[7] = K_NONE; f[6] = 0.0; f[7] = 0.0; r[6] = 0.0; r[7] = 0.0;
        B::fence();
        lds.mark(6);
        on_small();
      } else {
        build_slot(LI<6>{}); build_slot(LI<7>{});
        B::fence();
        lds.mark(5);
        warm_start(LI<CAP>{});
        B::fence();
        lds.mark(6);
        on_general();
      }
    }
  }

  // ---- PGS sweeps (mj_solPGS, elliptic cones) in MuJoCo's row order; a~ = (a0, a1, a2) = sum_j u~_j f_j is shared by the two lanes
  static LEG_FN void sub_sweeps(Sub& s) {
    const I leg = s.leg;
    const KP_ K = s.K;
    Fact& fc = s.fc;
    D &p1x = s.p1x, &p1z = s.p1z, &p2x = s.p2x, &p2z = s.p2z;
    I &nlim = s.nlim, &ncon = s.ncon;
    M& go = s.go;
    bool& small = s.small;
    D (&r)[CAP] = s.r; D (&f)[CAP] = s.f; D (&ut)[CAP][3] = s.ut; D (&Al)[CAP * (CAP + 1) / 2] = s.Al; D (&Adiag)[CAP] = s.Adiag; D (&Ainv)[CAP] = s.Ainv;
    D (&Ant)[3] = s.Ant;
    I (&kind)[CAP] = s.kind;
    D &a0 = s.a0, &a1 = s.a1, &a2 = s.a2;
    {
      const D mu = CP_CONTACT_MU;
      const D scale = 1.0 / (CP_MEANINERTIA * LNV);
      M sweeping = go;
      const M isL = leg == 0;
      // wave-uniform "some environment has such a row" bits
      bool anyLim[2][4], anyPair[2][3];
      lfor<0, 4>([&](auto jj) { constexpr int Jj = decltype(jj)::value; anyLim[0][Jj] = B::any(go & isL & (nlim > Jj)); anyLim[1][Jj] = B::any(go & (!isL) & (nlim > Jj)); });
      lfor<0, 3>([&](auto pp) { constexpr int P = decltype(pp)::value; anyPair[0][P] = B::any(go & isL & (ncon > P)); anyPair[1][P] = B::any(go & (!isL) & (ncon > P)); });
      D acc = 0.0;
      // a~ is kept per lane.  MuJoCo's row order groups the rows of a leg into blocks (connect L | connect R | limits L | limits R
      // | contacts L | contacts R): inside a block only the owner lane's steps change a~ -- the other lane's deltas are zero, its
      // copy stays at the value both lanes shared when the block began -- so the lanes exchange a~ once per BLOCK (both lanes take the
      // owner lane's value by one DPP broadcast per word: `sync`), not once per step: 12 instructions per block instead of 15 per step, and no DPP move on
      // the chain that runs from one step to the next.
      auto sync = [&](auto ww) {   // both lanes of every pair take the value of the owner's lane: one DPP broadcast per word
        constexpr int W = decltype(ww)::value;
        a0 = B::template pair_bcast<W>(a0); a1 = B::template pair_bcast<W>(a1); a2 = B::template pair_bcast<W>(a2);
      };
      // a connect row (slots 0, 1).  Reduced form: no clamp and no cost-increase revert -- for an unclamped row d = -res / A exactly
      // minimises its own quadratic, the change is -res^2 / (2 A) <= 0, so mj_solPGS's revert can never fire (see cassie_kernels_g16.hip).
      // The step functions are instantiated twice (six-row and eight-row sweeps), and which of the two a wavefront runs depends on
      // ALL of its 32 environments -- so their roundings must be identical, or an environment's result would depend on its
      // neighbours (the compiler contracts a*b + c*d into an FMA one way or the other depending on the code around it; measured:
      // the first version failed the neighbour tests).  Contraction is therefore off inside them and every fused multiply-add is
      // written out (B::fma), as in the step functions of cassie_kernels_g16.hip.
      auto eq_step = [&](auto ss, M owner, auto nr_) {
        LEG_FP_CONTRACT_OFF
        constexpr int NR = decltype(nr_)::value;   // own-row slots in use in this wavefront (6 or CAP)
        typename B::OwnerScope scope_(owner);   // op-counting builds of the CPU emulation only; empty on the device
        constexpr int S = decltype(ss)::value;
        const M mine = owner & sweeping & (kind[S] == K_EQ);
        const D res = B::fma(ut[S][2], a2, B::fma(ut[S][1], a1, B::fma(ut[S][0], a0, r[S])));
        D d = -(res * Ainv[S]);
        D chg = d * B::fma(Adiag[S], d, res);   // (Adiag of a connect row holds A_ii / 2)
        d = B::sel(mine, d, D(0.0)); chg = B::sel(mine, chg, D(0.0));
        a0 = B::fma(ut[S][0], d, a0); a1 = B::fma(ut[S][1], d, a1); a2 = B::fma(ut[S][2], d, a2);
        acc = acc + chg;
        f[S] = f[S] + d;
        lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; r[Ii] = B::fma(Al[symidx(CAP, Ii, S)], d, r[Ii]); });
      };
      auto lim_step = [&](auto ss, M owner, auto nr_) {
        LEG_FP_CONTRACT_OFF
        constexpr int NR = decltype(nr_)::value;
        typename B::OwnerScope scope_(owner);   // op-counting builds of the CPU emulation only; empty on the device
        constexpr int S = decltype(ss)::value;
        const M mine = owner & sweeping & (kind[S] == K_LIM);
        const D res = B::fma(ut[S][2], a2, B::fma(ut[S][1], a1, B::fma(ut[S][0], a0, r[S])));
        const D cand = B::fmax(B::fma(-res, Ainv[S], f[S]), D(0.0));
        D d = cand - f[S];
        D chg = d * B::fma(0.5 * Adiag[S], d, res);
        const M keep = mine & (chg <= 1e-10);
        d = B::sel(keep, d, D(0.0)); chg = B::sel(keep, chg, D(0.0));
        a0 = B::fma(ut[S][0], d, a0); a1 = B::fma(ut[S][1], d, a1); a2 = B::fma(ut[S][2], d, a2);
        acc = acc + chg;
        f[S] = f[S] + d;
        lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; r[Ii] = B::fma(Al[symidx(CAP, Ii, S)], d, r[Ii]); });
      };
      // 1 / (f' A f) of the ray update of pair P: a function of the pair's own force only, which nothing but the pair's own step
      // changes -- so it is formed at the head of the sweep, off the chain that runs from step to step through a~.
      D rden[3];
      auto pair_step = [&](auto pp, M owner, auto nr_) {
        LEG_FP_CONTRACT_OFF
        constexpr int NR = decltype(nr_)::value;
        typename B::OwnerScope scope_(owner);   // op-counting builds of the CPU emulation only; empty on the device
        constexpr int P = decltype(pp)::value;
        constexpr int N = 2 + 2 * P, T = 3 + 2 * P;
        const M mine = owner & sweeping & (kind[N] == K_CN);
        const D rn = B::fma(ut[N][2], a2, B::fma(ut[N][1], a1, B::fma(ut[N][0], a0, r[N])));
        const D rt = B::fma(ut[T][2], a2, B::fma(ut[T][1], a1, B::fma(ut[T][0], a0, r[T])));
        const D on = f[N], ot = f[T];
        const D Ann = Adiag[N], Att = Adiag[T], Ant_ = Ant[P];
        // normal-only update (taken when the normal force is ~0)
        const D fn_n = B::fmax(B::fma(-rn, Ainv[N], on), D(0.0));
        // ray update; rden = 0 where f' A f < MINVAL
        D x = -B::fma(ot, rt, on * rn) * rden[P];
        x = B::fmax(x, D(-1.0));
        const M use_n = on < LMINVAL;
        D fn = B::sel(use_n, fn_n, B::fma(x, on, on));
        D ft = B::sel(use_n, D(0.0), B::fma(x, ot, ot));
        // friction on one dimension: unconstrained minimiser unless it leaves the cone
        const D bc = B::fma(Ant_, fn - on, B::fma(-Att, ot, rt));
        const D x0 = -bc * Ainv[T];
        const D v1 = x0 * (1.0 / mu);
        const D val = B::fma(v1, v1, -(fn * fn));
        const M on_cone = (val >= 1e-10) & (val * Att * (mu * mu) >= 2e-10 * (v1 * v1));
        const D ftc = B::sel(on_cone, B::copysign(mu * fn, x0), x0);
        ft = B::sel(fn >= LMINVAL, ftc, ft);
        D dn = fn - on, dt = ft - ot;
        // 1/2 d'A d + d'res, grouped so that few operations wait for the tangent step
        D chg = B::fma(dt, B::fma(0.5 * Att, dt, B::fma(Ant_, dn, rt)), dn * B::fma(0.5 * Ann, dn, rn));
        const M keep = mine & (chg <= 1e-10);
        dn = B::sel(keep, dn, D(0.0)); dt = B::sel(keep, dt, D(0.0)); chg = B::sel(keep, chg, D(0.0));
        // a~ first (the next step waits for it), the own rows' residuals behind it
        a0 = B::fma(ut[T][0], dt, B::fma(ut[N][0], dn, a0)); a1 = B::fma(ut[T][1], dt, B::fma(ut[N][1], dn, a1)); a2 = B::fma(ut[T][2], dt, B::fma(ut[N][2], dn, a2));
        acc = acc + chg;
        f[N] = f[N] + dn; f[T] = f[T] + dt;
        lfor<0, NR>([&](auto ii) {
          constexpr int Ii = decltype(ii)::value;
          r[Ii] = B::fma(Al[symidx(CAP, Ii, T)], dt, B::fma(Al[symidx(CAP, Ii, N)], dn, r[Ii]));
        });
      };
      auto ray_den = [&](auto pp) {
        LEG_FP_CONTRACT_OFF
        constexpr int P = decltype(pp)::value;
        constexpr int N = 2 + 2 * P, T = 3 + 2 * P;
        const D on = f[N], ot = f[T];
        const D denom = B::fma(ot, B::fma(Adiag[T], ot, Ant[P] * on), on * B::fma(Ant[P], ot, Adiag[N] * on));
        rden[P] = B::sel(denom >= LMINVAL, B::rcp(denom), D(0.0));
      };
      I niter = 0;
      auto sweeps = [&](auto nr_) {
        constexpr int NR = decltype(nr_)::value;
        constexpr int NPAIR = NR == CAP ? 3 : 2;
        for (int iter = 0; iter < LEG_ITERS; iter++) {
          if (!B::any(sweeping)) break;
          acc = 0.0;
          lfor<0, NPAIR>([&](auto pp) { ray_den(pp); });
          eq_step(LI<0>{}, isL, nr_); eq_step(LI<1>{}, isL, nr_);
          sync(LI<0>{});
          eq_step(LI<0>{}, !isL, nr_); eq_step(LI<1>{}, !isL, nr_);
          sync(LI<1>{});
          if constexpr (NR == CAP) {
            lfor<0, 2>([&](auto ww) {
              constexpr int W = decltype(ww)::value;
              if (anyLim[W][0]) {   // limit j exists only if limit j - 1 does
                lfor<0, 4>([&](auto jj) { constexpr int Jj = decltype(jj)::value; if (anyLim[W][Jj]) lim_step(LI<7 - Jj>{}, W == 0 ? isL : !isL, nr_); });
                sync(ww);
              }
            });
          }
          lfor<0, 2>([&](auto ww) {
            constexpr int W = decltype(ww)::value;
            if (anyPair[W][0]) {
              lfor<0, NPAIR>([&](auto pp) { constexpr int P = decltype(pp)::value; if (anyPair[W][P]) pair_step(LI<P>{}, W == 0 ? isL : !isL, nr_); });
              sync(ww);
            }
          });
          const D improvement = -(acc + B::swap(acc));
          niter = niter + B::toI(sweeping);
          sweeping = sweeping & !(improvement * scale < CP_TOLERANCE);
        }
      };
      if (small) sweeps(LI<6>{});
      else sweeps(LI<CAP>{});
      s.niter = niter;
    }
  }

  // ---- after the solve: total generalised force, qacc (next warm start), mj_Euler with implicit joint damping
  template <bool HF = false>
  static LEG_FN void sub_finish(typename B::Lds& lds, Lane& st, bool integrate, Sub& s) {
    const I leg = s.leg;
    const KP_ K = s.K;
    Fact& fc = s.fc;
    D &p1x = s.p1x, &p1z = s.p1z, &p2x = s.p2x, &p2z = s.p2z;
    I &nlim = s.nlim, &ncon = s.ncon;
    M& go = s.go;
    bool& small = s.small;
    D (&r)[CAP] = s.r; D (&f)[CAP] = s.f; D (&ut)[CAP][3] = s.ut; D (&Al)[CAP * (CAP + 1) / 2] = s.Al; D (&Adiag)[CAP] = s.Adiag; D (&Ainv)[CAP] = s.Ainv;
    D (&Ant)[3] = s.Ant;
    I (&kind)[CAP] = s.kind;
    D &a0 = s.a0, &a1 = s.a1, &a2 = s.a2;
    // ---- total generalised force g = tau + J' f, accumulated from the rows' geometry (no Jacobian rows kept across the solve):
    // a force (Fx, Fz) at point p moves dof d (origin o_d, sign sigma_d) by sigma_d (Fx (pz - oz_d) - Fz (px - ox_d))
    D gb[3], gl[5], sb[3] = {D(0.0), D(0.0), D(0.0)};
    lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; gl[Dd] = lds.cld(C_TAUL + Dd); });
    {
      D oxk[6], ozk[6];
      lfor<0, 6>([&](auto jj) { constexpr int J = decltype(jj)::value; oxk[J] = lds.cld(C_OX + J); ozk[J] = lds.cld(C_OZ + J); });
      D sg[5];
      lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; sg[Dd] = kc(K, LK_DOF_SIGMA + Dd); });
      auto push = [&](D Fx, D Fz, D px, D pz, auto jl_) {   // generalised force of (Fx, Fz) at p on the dof whose link is Kin link Jl
        constexpr int Jl = decltype(jl_)::value;
        return Fx * (pz - ozk[Jl]) - Fz * (px - oxk[Jl]);
      };
      // connect: +F at p1 on the rod (hip, rod), -F at p2 on the tarsus (hip, knee, ankle); F = (f[0], f[1])
      const D Fx = f[0], Fz = f[1];
      gl[0] += sg[0] * (push(Fx, Fz, p1x, p1z, LI<1>{}) - push(Fx, Fz, p2x, p2z, LI<1>{}));
      gl[1] += -(sg[1] * push(Fx, Fz, p2x, p2z, LI<2>{}));
      gl[2] += -(sg[2] * push(Fx, Fz, p2x, p2z, LI<3>{}));
      gl[4] += sg[4] * push(Fx, Fz, p1x, p1z, LI<5>{});
      sb[2] += cp_dof_sigma[2] * (push(Fx, Fz, p1x, p1z, LI<0>{}) - push(Fx, Fz, p2x, p2z, LI<0>{}));
      // contact pairs: (x, z) force = (tangent row, normal row) at the contact point
      lfor<0, 3>([&](auto pp) {
        constexpr int P = decltype(pp)::value;
        constexpr int N = 2 + 2 * P, T = 3 + 2 * P;
        D px, pz, dist, cinvw; I depth;
        lds.ld_pair(P, px, pz, dist, cinvw, depth);
        const M isc = kind[N] == K_CN;
        D cfx = B::sel(isc, f[T], D(0.0)), cfz = B::sel(isc, f[N], D(0.0));
        if constexpr (HF) {   // (tangent, normal) forces -> world (x, z)
          const D nx = B::sel(isc, lds.ld_nrm(P), D(0.0)), nz = B::sqrt(1.0 - nx * nx);   // (an unused slot holds whatever LDS held)
          const D ft_ = cfx, fn_ = cfz;
          cfx = ft_ * nz + fn_ * nx; cfz = fn_ * nz - ft_ * nx;
        }
        px = B::sel(isc, px, D(0.0)); pz = B::sel(isc, pz, D(0.0));
        lfor<0, 4>([&](auto dd) {
          constexpr int Dd = decltype(dd)::value;
          gl[Dd] += B::sel(isc & (depth > Dd), sg[Dd] * push(cfx, cfz, px, pz, LI<Dd + 1>{}), D(0.0));
        });
        sb[0] += cfx; sb[1] += cfz;
        sb[2] += cp_dof_sigma[2] * push(cfx, cfz, px, pz, LI<0>{});
      });
      // joint limits: +-f on the limited dof
      lfor<0, 4>([&](auto ljj) {
        constexpr int LJ = decltype(ljj)::value;
        constexpr int S = 7 - LJ;
        D lpos, lsgn, linvw; I lj;
        lds.ld_lim(LJ, lpos, lsgn, linvw, lj);
        const M isl = kind[S] == K_LIM;
        lfor<0, 4>([&](auto dd) { constexpr int Dd = decltype(dd)::value; gl[Dd] += B::sel(isl & (lj == Dd), lsgn * f[S], D(0.0)); });
      });
    }
    lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; gb[Bc] = lds.cld(C_TAUB + Bc) + (sb[Bc] + B::swap(sb[Bc])); });
    // qacc = M^-1 g (next warm start).  mj_Euler's implicit joint damping, (M + h B) qacc' = g, without a second factorisation:
    // qacc' = (I + E)^-1 qacc with E = M^-1 h B a contraction whatever the pose (eigenvalues <= h B_d / (armature_d + joint
    // inertia) = 0.0393 for this model; tests/test_implicit_damping_bound.py), so x <- qacc - E x from x = qacc converges with
    // error 0.0393^n: DAMPING_SWEEPS = 12 leaves 1e-17.  The base dofs are undamped: E x only needs the leg part of x.
    B::fence();
    lds.mark(8);
    D xb[3], xl[5];
    minv_apply(fc, gb, gl, xb, xl);
    B::fence();
    D hb[3], hl[5];
    if (integrate) {
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; hb[Bc] = xb[Bc]; });
      lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; hl[Dd] = xl[Dd]; });
      const D zb[3] = {D(0.0), D(0.0), D(0.0)};
      D hdamp[5];
      lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; hdamp[Dd] = LH * kc(K, LK_DOF_DAMPING + Dd); });
      for (int it = 0; it < DAMPING_SWEEPS; it++) {
        D dl[5], eb[3], el[5];
        lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; dl[Dd] = hdamp[Dd] * hl[Dd]; });
        minv_apply(fc, zb, dl, eb, el);
        lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; hb[Bc] = xb[Bc] - eb[Bc]; });
        lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; hl[Dd] = xl[Dd] - el[Dd]; });
      }
    }
    lfor<0, 3>([&](auto bb) {
      constexpr int Bc = decltype(bb)::value;
      st.wb[Bc] = B::sel(go, xb[Bc], st.wb[Bc]);
      if (integrate) {
        const D vn = st.vb[Bc] + LH * hb[Bc];
        st.vb[Bc] = B::sel(go, vn, st.vb[Bc]);
        st.qb[Bc] = B::sel(go, st.qb[Bc] + LH * vn, st.qb[Bc]);
      }
    });
    lfor<0, 5>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      st.wl[Dd] = B::sel(go, xl[Dd], st.wl[Dd]);
      if (integrate) {
        const D vn = st.vl[Dd] + LH * hl[Dd];
        st.vl[Dd] = B::sel(go, vn, st.vl[Dd]);
        st.ql[Dd] = B::sel(go, st.ql[Dd] + LH * vn, st.ql[Dd]);
      }
    });
    lds.mark(9);
  }

  template <int MODE, bool HF = false>
  static LEG_FN void substep(typename B::Lds& lds, Lane& st, bool from_rec, M live, bool integrate, SubOut& out, const Terrain* hf = nullptr) {
    Sub s;
    auto solve = [&]() { sub_sweeps(s); };   // (inside the branch of the set-up that knows `small`: the six- or the eight-row sweeps)
    sub_setup<MODE, HF>(lds, st, from_rec, live, integrate, out, s, hf, solve, solve);
    out.niter = s.niter;
    B::fence();
    lds.mark(7);
    sub_finish<HF>(lds, st, integrate, s);
  }

  // ------------------------------------------------------------------------------------------------ operational-space state
  // Cassie2d::GetOperationalSpaceState (Cassie2d.cpp:218-237) with the RBDL-semantics tables from the kinematics of the last
  // setState (quirks Q1/Q2): body site on both lanes, the own foot (mean of the two toe sites) per lane.
  static LEG_FN void opstate(const D (&kqb)[3], const D (&kql)[5], const D (&kvb)[3], const D (&kvl)[5], D (&body)[4], D (&foot)[4]) {
    const I leg = B::opq(B::leg());
    Kin k;
    fk<1>(kqb, kql, kvb, kvl, leg, B::kbase(leg), k);
    const D bx = kqb[0] - cp_qpos0[0] + cp_link_off[1][0][0], bz = kqb[1] - cp_qpos0[1] + cp_link_off[1][0][1];
    auto site = [&](auto jj, I sid, D (&o)[4]) {
      constexpr int J = decltype(jj)::value;
      const D dx = ldc(&cp_site_d[0][0][0], sid * 2 + CP_NSITE * 2), dz = ldc(&cp_site_d[0][0][0], sid * 2 + 1 + CP_NSITE * 2);
      const D rx = k.c[J] * dx + k.s[J] * dz, rz = -k.s[J] * dx + k.c[J] * dz;
      o[0] = bx + k.ox[J] + rx; o[1] = bz + k.oz[J] + rz;
      o[2] = kvb[0] + k.vx[J] + k.w[J] * rz; o[3] = kvb[1] + k.vz[J] - k.w[J] * rx;
    };
    site(LI<0>{}, I(1), body);
    D sa[4], sbv[4];
    site(LI<4>{}, leg * 2 + 2, sa);
    site(LI<4>{}, leg * 2 + 3, sbv);
    lfor<0, 4>([&](auto ii) { constexpr int Ii = decltype(ii)::value; foot[Ii] = (sa[Ii] + sbv[Ii]) / 2.0; });
  }

  static LEG_FN M in_range(D x) { return B::fabs(x) <= FINITE_BOUND; }  // false for NaN and +-inf

  // ------------------------------------------------------------------------------------------------ HBM <-> lane
  // Per-lane "pointers" (B::P: the lane's state record / action row / observation row ...); base-dof fields are read by both
  // lanes of an environment and written by the left lane only.
  struct Io {
    typename B::P rec, act, obs, tobs, rew;
    typename B::P8 done;
    bool has_act, has_tobs;
  };

  // ------------------------------------------------------------------------------------------------ masked reset
  // Cassie2dEnv.reset / Cassie2d::Reset for the environments of `want` (identical on the two lanes of a pair; CassieVecReset): state <-
  // (qin, vin) or the reset pose, mj_forward with the stale ctrl, no setState; observation = the 17 op-space values, trajectory slots zero
  // -- what env_step's reset pass does for a terminated environment, for any state.  o.pend = 1: the state needs more rows than this
  // tier holds and the environment is left untouched (the wave-per-environment reset kernel takes it).
  template <bool HF = false>
  static LEG_FN void env_reset(const EnvCfg& cfg, typename B::Lds& lds, const Io& io, M want, typename B::P qin, typename B::P vin, bool has_qv, Out& o,
                               const Terrain* hf = nullptr) {
    const I leg = B::leg();
    const I lo = leg * 5 + 3, ao = leg * 3;
    const M left = leg == 0;
    const M all = want | !want;
    Lane st;
    lfor<0, 3>([&](auto bb) {
      constexpr int Bc = decltype(bb)::value;
      st.wb[Bc] = B::pld(io.rec, I(ES_WS + Bc));
      st.qb[Bc] = has_qv ? B::pld(qin, I(Bc)) : D(cp_env_qinit[Bc]);
      st.vb[Bc] = has_qv ? B::pld(vin, I(Bc)) : D(0.0);
      lds.cst(C_CTRL + Bc, B::pld(io.rec, ao + (ES_CTRL + Bc)), all);
      lds.cst(C_ACT + Bc, D(0.0), all);
      lds.cst(C_KQ + Bc, B::pld(io.rec, I(ES_KQ + Bc)), all); lds.cst(C_KV + Bc, B::pld(io.rec, I(ES_KV + Bc)), all);
    });
    lfor<0, 5>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      st.wl[Dd] = B::pld(io.rec, lo + (ES_WS + Dd));
      st.ql[Dd] = has_qv ? B::pld(qin, lo + Dd) : ldc(cp_env_qinit, lo + Dd);
      st.vl[Dd] = has_qv ? B::pld(vin, lo + Dd) : D(0.0);
      lds.cst(C_QST + Dd, st.ql[Dd], all);
      lds.cst(C_KQ + 3 + Dd, B::pld(io.rec, lo + (ES_KQ + Dd)), all); lds.cst(C_KV + 3 + Dd, B::pld(io.rec, lo + (ES_KV + Dd)), all);
    });
    lds.cst(C_TIME, D(0.0), all);
    lds.cst(C_A2, D(0.0), all);
    SubOut so;
    substep<1, HF>(lds, st, true, want, false, so, hf);
    const M ovf = want & so.overflow;
    const M live = want & !ovf;
    o.pend = B::seli(ovf, I(1), I(0));
    o.niter = so.niter;
    o.do_reset = live; o.bad = (want & !want); o.set_state = o.bad;
    if (cfg.want_obs) {
      const bool fix_kin = (cfg.flags & FLAG_FIX_STALE_KIN) != 0;
      D body[4], foot[4];
      D kqb[3], kql[5], kvb[3], kvl[5];
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; kqb[Bc] = fix_kin ? st.qb[Bc] : lds.cld(C_KQ + Bc); kvb[Bc] = fix_kin ? st.vb[Bc] : lds.cld(C_KV + Bc); });
      lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; kql[Dd] = fix_kin ? st.ql[Dd] : lds.cld(C_KQ + 3 + Dd); kvl[Dd] = fix_kin ? st.vl[Dd] : lds.cld(C_KV + 3 + Dd); });
      opstate(kqb, kql, kvb, kvl, body, foot);
      D ob[5], of[6];
      ob[0] = body[1]; ob[1] = st.qb[2]; ob[2] = body[2]; ob[3] = body[3]; ob[4] = st.vb[2];
      of[0] = foot[0] - body[0]; of[1] = foot[1]; of[2] = 0.0; of[3] = foot[2]; of[4] = foot[3]; of[5] = 0.0;
      lfor<0, 5>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::pst(io.obs, I(Ii), ob[Ii], live & left); });
      lfor<0, 6>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::pst(io.obs, leg * 6 + (5 + Ii), of[Ii], live); });
      lfor<0, 9>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::pst(io.obs, I(17 + Ii), D(0.0), live & !left); });
    }
    // ---- state write-back (what store_state of the wave-per-environment reset kernel writes: q, v, warm start, qstate, clock, iteration
    // count, cold start of the OSC QP; ctrl and the setState copies stay)
    lfor<0, 3>([&](auto bb) {
      constexpr int Bc = decltype(bb)::value;
      const M lv = live & left;
      B::pst(io.rec, I(ES_Q + Bc), st.qb[Bc], lv); B::pst(io.rec, I(ES_V + Bc), st.vb[Bc], lv); B::pst(io.rec, I(ES_WS + Bc), st.wb[Bc], lv);
      B::pst(io.rec, I(ES_QSTATE + Bc), st.qb[Bc], lv);
    });
    lfor<0, 5>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      B::pst(io.rec, lo + (ES_Q + Dd), st.ql[Dd], live); B::pst(io.rec, lo + (ES_V + Dd), st.vl[Dd], live); B::pst(io.rec, lo + (ES_WS + Dd), st.wl[Dd], live);
      B::pst(io.rec, lo + (ES_QSTATE + Dd), st.ql[Dd], live);
    });
    B::pst(io.rec, I(ES_TIME), D(0.0), live & left);
    B::pst(io.rec, I(ES_NITER), B::toD(o.niter), live & left);
    B::pst(io.rec, I(ES_QPWSET), D(0.0), live & left);
  }

  // ------------------------------------------------------------------------------------------------ end of an Env.step
  // Observation / reward / termination of the environments of `live`, stores, and the state of a terminated environment set to the
  // reset pose.  Returns true when the caller has to run the reset pass (mj_forward only, for o.do_reset) and call this again with
  // reset_pass = true, which stores the reset observation (and returns false).
  static LEG_FN bool step_outputs(const EnvCfg& cfg, typename B::Lds& lds, const Io& io, Lane& st, M live, Out& o, bool reset_pass) {
    const I leg = B::leg();
    const I lo = leg * 5 + 3;
    const M left = leg == 0;
    const bool fix_kin = (cfg.flags & FLAG_FIX_STALE_KIN) != 0;
    // ---- end-of-step section: operational-space state from the kinematics of the last setState (quirks Q1/Q2)
    D body[4], foot[4];
    {
      D kqb[3], kql[5], kvb[3], kvl[5];
      // (both sources are read and the VALUE is chosen: with a backend whose snapshot slots are plain memory -- cassie_duo_core.h: the
      // HBM record -- "flag ? st.x : slot" becomes a load through a chosen POINTER, which keeps the whole lane state in scratch)
      lfor<0, 3>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        const D sq = st.qb[Bc], sv = st.vb[Bc], lq = lds.cld(C_KQ + Bc), lv = lds.cld(C_KV + Bc);
        kqb[Bc] = fix_kin ? sq : lq; kvb[Bc] = fix_kin ? sv : lv;
      });
      lfor<0, 5>([&](auto dd) {
        constexpr int Dd = decltype(dd)::value;
        const D sq = st.ql[Dd], sv = st.vl[Dd], lq = lds.cld(C_KQ + 3 + Dd), lv = lds.cld(C_KV + 3 + Dd);
        kql[Dd] = fix_kin ? sq : lq; kvl[Dd] = fix_kin ? sv : lv;
      });
      opstate(kqb, kql, kvb, kvl, body, foot);
    }
    const D bodyx = body[0], zz = body[1], pitch = st.qb[2];
    // obs[0..4] = z, pitch, xd, zd, pitchd ; own foot at obs[5 + 6 leg ..]: x - bodyx, z, 0 (Q4), xd, zd, 0 (Q4)
    D ob[5], of[6];
    ob[0] = zz; ob[1] = pitch; ob[2] = body[2]; ob[3] = body[3]; ob[4] = st.vb[2];
    of[0] = foot[0] - bodyx; of[1] = foot[1]; of[2] = 0.0; of[3] = foot[2]; of[4] = foot[3]; of[5] = 0.0;
    auto put = [&](typename B::P row, M m) {
      lfor<0, 5>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::pst(row, I(Ii), ob[Ii], m & left); });
      lfor<0, 6>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::pst(row, leg * 6 + (5 + Ii), of[Ii], m); });
    };
    if (reset_pass) {
      // Cassie2dEnv.reset returns the 17 op-space values; the trajectory slots of the observation are zero there
      put(io.obs, o.do_reset);
      lfor<0, 9>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::pst(io.obs, I(17 + Ii), D(0.0), o.do_reset & !left); });
      return false;
    }
    D ref[9];
    lfor<0, 9>([&](auto ii) { constexpr int Ii = decltype(ii)::value; ref[Ii] = 0.0; });
    D reward = 0.0;
    M done;
    if (cfg.env_kind == 0) {
      // reference-gait lookup (cassie2d_trajectory.py:16-19), reward (cassie2d.py:197-218); qstate is the reset pose unless
      // FLAG_FIX_STALE_QSTATE (quirk Q3)
      const double tmax = cfg.traj_tmax;
      const I idx = B::toint(B::fmod(lds.cld(C_TIME), tmax) / tmax * (double)cfg.traj_n);
      constexpr int COLS[9] = {0, 1, 2, 3, 4, 6, 8, 9, 11};
      lfor<0, 9>([&](auto ii) { constexpr int Ii = decltype(ii)::value; ref[Ii] = B::ldg(cfg.traj_qpos, idx * LNV + COLS[Ii]); });
      const bool fixq = (cfg.flags & FLAG_FIX_STALE_QSTATE) != 0;
      // joints 3,4,6 (left) and 8,9,11 (right): own hip + knee + toe, partner's by exchange; left first as in the reference
      const D mine3 = fixq ? st.ql[0] + st.ql[1] + st.ql[3] : lds.cld(C_QST + 0) + lds.cld(C_QST + 1) + lds.cld(C_QST + 3);
      const D other3 = B::swap(mine3);
      D j = B::sel(left, mine3, other3);
      j = j + B::sel(left, other3, mine3);
      D sum = 0.0;
      lfor<3, 9>([&](auto ii) { constexpr int Ii = decltype(ii)::value; sum += ref[Ii]; });
      j = j - sum; j = B::exp(-(j * j));
      D pp = bodyx + zz;
      pp = pp - (ref[0] + ref[1]); pp = B::exp(-(pp * pp));
      D oo = pitch;
      oo = oo - ref[2]; oo = B::exp(-(oo * oo));
      reward = 0.5 * j + 0.3 * pp + 0.1 * oo;
      done = (zz < 0.6) | (zz > 1.2) | (reward < 0.6);
    } else {
      const D a2own = lds.cld(C_A2);
      const D a2 = a2own + B::swap(a2own);
      // m = (left foot x - bodyx + right foot x - bodyx) / 2, left first
      const D fo = B::swap(of[0]);
      const D m = (B::sel(left, of[0], fo) + B::sel(left, fo, of[0])) / 2.0;
      reward = 0.0;
      reward = reward - 2.0 * (0.9 - zz) * (0.9 - zz);
      reward = reward - 2.0 * m * m;
      reward = reward + 1.0;
      reward = reward - 0.001 * a2;
      done = zz < 0.5;
    }
    // failure guard (MuJoCo's mj_checkPos / mj_checkVel): a state outside the finite range terminates the episode
    M okl = in_range(st.qb[0]) & in_range(st.qb[1]) & in_range(st.qb[2]) & in_range(st.vb[0]) & in_range(st.vb[1]) & in_range(st.vb[2]);
    lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; okl = okl & in_range(st.ql[Dd]) & in_range(st.vl[Dd]); });
    okl = okl & B::swapm(okl);
    const M bad = live & ((!okl) | (!in_range(reward)));
    o.bad = bad;
    lfor<0, 5>([&](auto ii) { constexpr int Ii = decltype(ii)::value; ob[Ii] = B::sel(bad, D(0.0), ob[Ii]); });
    lfor<0, 6>([&](auto ii) { constexpr int Ii = decltype(ii)::value; of[Ii] = B::sel(bad, D(0.0), of[Ii]); });
    lfor<0, 9>([&](auto ii) { constexpr int Ii = decltype(ii)::value; ref[Ii] = B::sel(bad, D(0.0), ref[Ii]); });
    reward = B::sel(bad, D(0.0), reward);
    done = done | bad;
    if (cfg.auto_reset) {
      // the reset below also clears every NaN carrier (warm start, ctrl, setState copies)
      lfor<0, 3>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        st.wb[Bc] = B::sel(bad, D(0.0), st.wb[Bc]);
        lds.cst(C_KQ + Bc, D(cp_env_qinit[Bc]), bad); lds.cst(C_KV + Bc, D(0.0), bad);
        lds.cst(C_CTRL + Bc, D(0.0), bad);
      });
      lfor<0, 5>([&](auto dd) {
        constexpr int Dd = decltype(dd)::value;
        st.wl[Dd] = B::sel(bad, D(0.0), st.wl[Dd]);
        lds.cst(C_KQ + 3 + Dd, ldc(cp_env_qinit, lo + Dd), bad); lds.cst(C_KV + 3 + Dd, D(0.0), bad);
      });
      o.set_state = o.set_state | bad;
    }
    if (io.has_tobs) {
      put(io.tobs, live);
      lfor<0, 9>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::pst(io.tobs, I(17 + Ii), ref[Ii], live & !left); });
    }
    put(io.obs, live);
    lfor<0, 9>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::pst(io.obs, I(17 + Ii), ref[Ii], live & !left); });
    B::pst(io.rew, I(0), reward, live & left);
    B::pst8(io.done, done, live & left);
    o.do_reset = live & done & (cfg.auto_reset != 0);
    if (!B::any(o.do_reset)) return false;
    // ---- Cassie2dEnv.reset for the terminated environments: qinit, mj_forward with the stale ctrl, no setState
    lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; st.qb[Bc] = B::sel(o.do_reset, D(cp_env_qinit[Bc]), st.qb[Bc]); st.vb[Bc] = B::sel(o.do_reset, D(0.0), st.vb[Bc]); });
    lfor<0, 5>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      const D qi = ldc(cp_env_qinit, lo + Dd);
      st.ql[Dd] = B::sel(o.do_reset, qi, st.ql[Dd]); st.vl[Dd] = B::sel(o.do_reset, D(0.0), st.vl[Dd]);
      lds.cst(C_QST + Dd, qi, o.do_reset);
    });
    lds.cst(C_TIME, D(0.0), o.do_reset);
    return true;   // the reset pose on the flat floor has 12 rows: never an overflow
  }

  // ------------------------------------------------------------------------------------------------ fused Env.step
  // MODE: 0 PD (Cassie2d::StepPd), 1 torque (Cassie2d::Step), 2 motor commands from the state record (StepOsc / StepJacobian:
  // the controller kernel wrote them).  valid: the lane's environment exists.  One loop, ONE copy of the substep code: passes
  // 0..n_sub-1 are the physics substeps; the end-of-step section computes observation / reward / termination and stores them; if
  // any environment of the wave terminated, one more pass (mj_forward only, on the reset pose, for those environments) leaves
  // the reset observation.  On return `o` says what the caller still has to do (pending count, failure-guard counter).
  template <int MODE, bool HF = false>
  static LEG_FN void env_step(const EnvCfg& cfg, typename B::Lds& lds, const Io& io, M valid, Out& o, const Terrain* hf = nullptr) {
    const I leg = B::leg();
    const I lo = leg * 5 + 3, ao = leg * 3;
    const M left = leg == 0;
    Lane st;
    // ---- load: hot part to registers, cold part to the lane's LDS slots (ES_KQ / ES_KV are not read: the first setState of the
    // step overwrites them)
    lfor<0, 3>([&](auto bb) {
      constexpr int Bc = decltype(bb)::value;
      st.qb[Bc] = B::pld(io.rec, I(ES_Q + Bc)); st.vb[Bc] = B::pld(io.rec, I(ES_V + Bc)); st.wb[Bc] = B::pld(io.rec, I(ES_WS + Bc));
      lds.cst(C_CTRL + Bc, B::pld(io.rec, ao + (ES_CTRL + Bc)), valid | !valid);
      D a = 0.0;
      if (io.has_act && MODE != 2) a = B::pld(io.act, ao + Bc);
      lds.cst(C_ACT + Bc, a, valid | !valid);
    });
    lfor<0, 5>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      st.ql[Dd] = B::pld(io.rec, lo + (ES_Q + Dd)); st.vl[Dd] = B::pld(io.rec, lo + (ES_V + Dd)); st.wl[Dd] = B::pld(io.rec, lo + (ES_WS + Dd));
      lds.cst(C_QST + Dd, B::pld(io.rec, lo + (ES_QSTATE + Dd)), valid | !valid);
    });
    lds.cst(C_TIME, B::pld(io.rec, I(ES_TIME)), valid | !valid);
    {
      // sum(action^2) of cassie_stand2d.py's reward: the own leg's three components; component 6 (OSC) rides on the left lane
      D a2own = 0.0;
      if (io.has_act && cfg.env_kind != 0) {
        lfor<0, 3>([&](auto aa) { constexpr int A_ = decltype(aa)::value; const D a = B::pld(io.act, ao + A_); a2own += a * a; });
        if (cfg.adim == 7) { const D a = B::pld(io.act, I(6)); a2own += B::sel(left, a * a, D(0.0)); }
      }
      lds.cst(C_A2, a2own, valid | !valid);
    }
    M live = valid;
    o.set_state = (valid & !valid);
    o.pend = 0; o.niter = 0;
    if (cfg.cont) o.niter = B::toint(B::pld(io.rec, I(ES_NITER)));
    o.do_reset = o.set_state; o.bad = o.set_state;
    const bool fix_kin = (cfg.flags & FLAG_FIX_STALE_KIN) != 0;
    bool reset_pass = false;
    int sub = 0;
    SubOut so;
    while (true) {
      substep<MODE, HF>(lds, st, reset_pass || MODE == 2, reset_pass ? o.do_reset : live, !reset_pass, so, hf);
      if (!reset_pass) {
        const M ovf = live & so.overflow;
        o.pend = B::seli(ovf, I(cfg.n_sub - sub + cfg.pend_extra), o.pend);   // hand the rest of this environment to the next kernel tier
        live = live & !ovf;
        o.niter = o.niter + B::seli(live, so.niter, I(0));
        o.set_state = o.set_state | live;
        sub++;
        if (sub < cfg.n_sub && B::any(live)) continue;
      }
      if (!cfg.want_obs) break;
      if (!step_outputs(cfg, lds, io, st, live, o, reset_pass)) break;
      reset_pass = true;
    }
    // ---- state write-back
    lfor<0, 3>([&](auto bb) {
      constexpr int Bc = decltype(bb)::value;
      const M lv = valid & left;
      B::pst(io.rec, I(ES_Q + Bc), st.qb[Bc], lv); B::pst(io.rec, I(ES_V + Bc), st.vb[Bc], lv); B::pst(io.rec, I(ES_WS + Bc), st.wb[Bc], lv);
      B::pst(io.rec, I(ES_KQ + Bc), lds.cld(C_KQ + Bc), lv & o.set_state); B::pst(io.rec, I(ES_KV + Bc), lds.cld(C_KV + Bc), lv & o.set_state);
      B::pst(io.rec, ao + (ES_CTRL + Bc), lds.cld(C_CTRL + Bc), valid);
      // qstate of the base dofs: reset writes qinit there too (only the general kernels' record copy ever reads it back)
      B::pst(io.rec, I(ES_QSTATE + Bc), D(cp_env_qinit[Bc]), lv & o.do_reset);
    });
    lfor<0, 5>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      B::pst(io.rec, lo + (ES_Q + Dd), st.ql[Dd], valid); B::pst(io.rec, lo + (ES_V + Dd), st.vl[Dd], valid); B::pst(io.rec, lo + (ES_WS + Dd), st.wl[Dd], valid);
      B::pst(io.rec, lo + (ES_KQ + Dd), lds.cld(C_KQ + 3 + Dd), valid & o.set_state); B::pst(io.rec, lo + (ES_KV + Dd), lds.cld(C_KV + 3 + Dd), valid & o.set_state);
      B::pst(io.rec, lo + (ES_QSTATE + Dd), lds.cld(C_QST + Dd), valid & o.do_reset);
    });
    B::pst(io.rec, I(ES_TIME), lds.cld(C_TIME), valid & left);
    B::pst(io.rec, I(ES_NITER), B::toD(o.niter), valid & left);
    B::pst(io.rec, I(ES_QPWSET), D(0.0), valid & left & o.do_reset);   // new episode: cold start of the OSC QP too
  }
};

}  // namespace leg
}  // namespace cassie
#endif
