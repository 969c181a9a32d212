// cassie3d_leg_core.h -- Cassie3d physics (model/cassie3d_stiff.xml, BASELINE.json configs[4]) with ONE LANE PER LEG: two lanes per
// environment, 32 environments per wavefront (r04; the counterpart of cassie_leg_core.h for the floating-base 3-D mechanism).
//
// Why: env_step3d_kernel (cassie3d_kernels.hip) spends one wavefront on one robot -- 255 k VALU instructions per environment-step,
// most of them with a handful of useful lanes (r03 PMC) -- and is bound by the latency of its own chains with LDS-limited residency.
// Here every phase of the substep is scalar-per-lane code over ONE LEG, so a wave instruction does useful work for 64 (environment,
// leg) pairs, and the two lanes of an environment meet only where the mechanism couples its legs: the six base dofs.
//
//   * M = [[B, C_L', C_R'], [C_L, L_L, 0], [C_R, 0, L_R]] with B 6x6 (3 world translations + 3 body-frame rotations), L_k 7x7
//     (hip roll / yaw / pitch, knee, tarsus, toe, achilles rod), C_k 7x6.  A lane runs the kinematics of its leg down the tree (world
//     frame, positions relative to the pelvis origin), accumulates composite inertias and inertial forces back up it, and gets L, C,
//     its share of B and the bias forces from the "momentum" vectors P_j, Q_j of each hinge's subtree:
//     M_ij = om_i . P_j + vv_i . Q_j for an ancestor i.  The whole-body sums meet by one lane-pair exchange.
//   * Block factorisation as in 2-D: L^-1 in place, Y = L^-1 C, S = B - sum_k C_k' Y_k = F F', G = F^-1; M^-1 is never formed.
//   * Constraint rows belong to a leg (its 3 connect rows, its 6 joint limits, its 8 collision spheres x 3 rows; the pelvis sphere
//     rides on the left lane).  Rows live in per-lane LDS slots ([slot][lane]: bank = lane), MATRIX-FREE: with z_i = L^-1 jl_i and
//     u~_i = G (jb_i - Y' jl_i), (A f)_i = jl_i . c + u~_i . a~ + R_i f_i where c = sum_j z_j f_j (own leg, 7 numbers, lane-local)
//     and a~ = sum_j u~_j f_j (6 numbers, shared by the pair); a Gauss-Seidel step reads its row and moves c and a~ -- no A is
//     stored but the 3x3 diagonal block of a contact.  qacc = qacc_smooth + [G' a~ ; c - Y G' a~] needs no second solve.
//   * PGS in MuJoCo's row order (connect L, R; limits L, R; contacts pelvis + L, R), elliptic cones with the QCQP of mju_QCQP2,
//     a~ exchanged once per block; every loop over rows is ROLLED (row data is in LDS, so run-time indices are free).
//   * Capacity: the 3 connect rows of a leg in registers + NSLOT3 = 160 per-lane LDS slots
//     at 18 per joint limit and 52 per contact (3 contacts; 2 contacts + 3 limits; 1 contact + 6 limits): 80 KB per wavefront, two wavefronts per CU.  An environment that
//     needs more is left untouched from that substep on and handed to env_step3d_kernel through `pending`.
//
// The arithmetic restates the same mj_forward / mj_Euler as cassie3d_kernels.hip (kinematics, CRB, RNE bias, plane-sphere and
// plane-capsule collision, connect / limit / elliptic contact rows, warm start, PGS, implicit joint damping); what differs is the
// factorisation and the grouping of sums.  Written against the same kind of backend as cassie_leg_core.h, so that the SAME source
// is compiled by hipcc for gfx950 (cassie3d_leg.hip -- the product) and by g++ with a lane emulation (oracle/leg_host/) that the CPU
// test-suite checks against the oracle (tests/test_leg3d_host.py).  The latter is test infrastructure: the library has no CPU path.
#ifndef CASSIE3D_LEG_CORE_H_
#define CASSIE3D_LEG_CORE_H_

#include "cassie3d_tables.h"
#include "cassie3d_legk.h"
#include "cassie3d_layout.h"

#ifndef LEG_FN
#define LEG_FN __device__ __forceinline__
#endif
#ifndef LEG3_SUBSTEP_FN   // the substep is one real function per kernel: the step loop around it stays small
#define LEG3_SUBSTEP_FN __device__ __forceinline__
#endif

namespace cassie3d {
namespace leg {

#ifndef LEG3_STAT   // instrumented CPU builds only (tools/leg3d_stats.py): counts executed sweeps / limit steps / contact steps / Newton iterations per wavefront
#define LEG3_STAT(k)
#endif
#ifndef LEG3_MARK   // profiling builds of the device kernel only (cassie3d_leg.hip, -DCASSIE3D_PHASE_TIMING; tools/phase_profile_3d.py): shader cycles per phase of the substep
#define LEG3_PHASE_BEGIN
#define LEG3_MARK(k)
#define LEG3_SW_MARK(k)
#define LEG3_SW_FLUSH
#endif
#ifndef LEG3_ITERS
#define LEG3_ITERS ITERATIONS   // (timing experiments only: -DLEG3_ITERS=n)
#endif
constexpr int NSLOT3 = 160;   // per-lane LDS slots (doubles): 80 KB per wavefront = two wavefronts per CU
// Rows of the matrix-free solver.  The three connect rows of a leg always exist: they stay in REGISTERS (static indices).  Joint
// limits and contacts are compacted into the per-lane LDS slots, limits first:
//   limit record   z(7) u~(6) R b A_ii f ks                                                       18 slots  (the Jacobian of a joint limit is
//                  s e_k: ks = s (k + 1) stands for it, jl . c = s c_k comes from a select chain, 1 / A_ii is recomputed)
//   contact record 3 x [jl(7) u~(6) b f] + R + the packed 3x3 diagonal block of A                  52 slots  (no z: a contact step
//                  applies c += L^-1 (sum_i jl_i d_i) instead -- 49 multiply-adds more per step, 21 slots less per contact)
enum { R3_Z = 0, R3_UT = 7, R3_R = 13, R3_B = 14, R3_AD = 15, R3_F = 16, R3_KS = 17, R3_N = 18,
       R3_POS = R3_R, R3_INVW = R3_B,   // a raw limit row carries its position and inverse weight there
       C3_JL = 0, C3_UT = 7, C3_B = 13, C3_F = 14, C3_ROW = 15, C3_R = 3 * C3_ROW, C3_BLK = C3_R + 1, C3_N = C3_BLK + 6,
       C3_JB = C3_UT, C3_POS = C3_B, C3_INVW = C3_F,   // raw contact rows likewise
       DYN0 = 0 };
// scratch layout of the first two passes (dead before any row is written): per link h(3) J(6) F(3) T(3); per dof axis(3) anchor(3)
enum { T3_LINK = 0, T3_DOF = 7 * 15, T3_END = 7 * 15 + 7 * 6 };
static_assert(T3_END <= NSLOT3 && DYN0 + 2 * C3_N + 3 * R3_N <= NSLOT3 && DYN0 + 3 * C3_N <= NSLOT3 && DYN0 + C3_N + 6 * R3_N <= NSLOT3, "slot budget");

template <int I_> struct LI { static constexpr int value = I_; };
template <int B_, int E_, class F> LEG_FN void lfor(F&& f) {
  if constexpr (B_ < E_) { f(LI<B_>{}); lfor<B_ + 1, E_>(f); }
}
constexpr int symidx(int n, int i, int j) { return i <= j ? i * n - i * (i - 1) / 2 + (j - i) : j * n - j * (j - 1) / 2 + (i - j); }
constexpr int lowidx(int i, int j) { return i * (i + 1) / 2 + j; }   // (i, j <= i) of a packed lower triangle

template <class B> struct Core3 {
  typedef typename B::D D;
  typedef typename B::I I;
  typedef typename B::M M;
  typedef typename B::K KP_;
  static LEG_FN D kc(KP_ K, int idx) { return B::kld(K, idx); }
  static LEG_FN D ldc(const double* t, int i) { return B::ldc(t, I(i)); }

  // qpos / qvel / qacc_warmstart: base (position, quaternion; world linear + body angular velocity) replicated on both lanes, own leg
  struct Lane { D qp[3], qq[4], ql[7], vb[6], vl[7], wb[6], wl[7]; };
  struct Frame { D pos[3], mat[9]; };              // of a link, position relative to the pelvis origin
  struct Vel { D w[3], vo[3], al[3], ao[3]; };     // angular / origin velocity, bias accelerations (RNE with qacc = 0)

  // ------------------------------------------------------------------------------------------------ small vector algebra
  static LEG_FN void cross(const D (&a)[3], const D (&b)[3], D (&r)[3]) {
    const D x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
  }
  static LEG_FN D dot(const D (&a)[3], const D (&b)[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
  static LEG_FN void mv(const D (&m)[9], const D (&v)[3], D (&r)[3]) {
    const D x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
  }
  static LEG_FN void mm(const D (&a)[9], const D (&b)[9], D (&r)[9]) {
    D t[9];
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; lfor<0, 3>([&](auto jj) { constexpr int j = decltype(jj)::value;
      t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j]; }); });
    lfor<0, 9>([&](auto ii) { constexpr int i = decltype(ii)::value; r[i] = t[i]; });
  }
  // symmetric 3x3 (xx yy zz xy xz yz) times vector
  static LEG_FN void sym_mv(const D (&s)[6], const D (&v)[3], D (&r)[3]) {
    const D x = s[0] * v[0] + s[3] * v[1] + s[4] * v[2], y = s[3] * v[0] + s[1] * v[1] + s[5] * v[2], z = s[4] * v[0] + s[5] * v[1] + s[2] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
  }

  // ------------------------------------------------------------------------------------------------ kinematics
  static LEG_FN void base_frame(const Lane& st, Frame& f, Vel& v) {
    D qw = st.qq[0], qx = st.qq[1], qy = st.qq[2], qz = st.qq[3];
    const D n = B::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw = qw / n; qx = qx / n; qy = qy / n; qz = qz / n;
    f.mat[0] = 1.0 - 2.0 * (qy * qy + qz * qz); f.mat[1] = 2.0 * (qx * qy - qw * qz); f.mat[2] = 2.0 * (qx * qz + qw * qy);
    f.mat[3] = 2.0 * (qx * qy + qw * qz); f.mat[4] = 1.0 - 2.0 * (qx * qx + qz * qz); f.mat[5] = 2.0 * (qy * qz - qw * qx);
    f.mat[6] = 2.0 * (qx * qz - qw * qy); f.mat[7] = 2.0 * (qy * qz + qw * qx); f.mat[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
    f.pos[0] = 0.0; f.pos[1] = 0.0; f.pos[2] = 0.0;
    const D wl[3] = {st.vb[3], st.vb[4], st.vb[5]};
    mv(f.mat, wl, v.w);
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; v.vo[i] = st.vb[i]; v.al[i] = 0.0; v.ao[i] = 0.0; });
  }
  // frame of leg link J (1..7) from its parent's; ax = the hinge axis in world axes, the hinge anchor is the link origin.
  // sn, cs: sine / cosine of (q - ref).  WITH_VEL: also the velocity recursion of mj_comVel / the bias accelerations of mj_rne.
  template <int J, bool WITH_VEL>
  static LEG_FN void child_frame(KP_ K, const Frame& pf, const Vel& pv, D sn, D cs, D qd, Frame& f, Vel& v, D (&ax)[3]) {
    D lp[3], lr[9], la[3], r[3], m0[9];
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; lp[i] = kc(K, LK3_LINK_POS + 3 * (J - 1) + i); la[i] = kc(K, LK3_DOF_AXIS + 3 * (J - 1) + i); });
    lfor<0, 9>([&](auto ii) { constexpr int i = decltype(ii)::value; lr[i] = kc(K, LK3_LINK_ROT + 9 * (J - 1) + i); });
    mv(pf.mat, lp, r);
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; f.pos[i] = pf.pos[i] + r[i]; });
    mm(pf.mat, lr, m0);
    mv(m0, la, ax);
    // Rodrigues rotation about the world axis, applied on the left (mj_kinematics)
    const D t1 = 1.0 - cs;
    const D R[9] = {cs + ax[0] * ax[0] * t1, ax[0] * ax[1] * t1 - ax[2] * sn, ax[0] * ax[2] * t1 + ax[1] * sn,
                    ax[1] * ax[0] * t1 + ax[2] * sn, cs + ax[1] * ax[1] * t1, ax[1] * ax[2] * t1 - ax[0] * sn,
                    ax[2] * ax[0] * t1 - ax[1] * sn, ax[2] * ax[1] * t1 + ax[0] * sn, cs + ax[2] * ax[2] * t1};
    mm(R, m0, f.mat);
    if constexpr (WITH_VEL) {
      D t[3], t2[3];
      const D axqd[3] = {ax[0] * qd, ax[1] * qd, ax[2] * qd};
      cross(pv.w, r, t);
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; v.vo[i] = pv.vo[i] + t[i]; });
      cross(pv.w, t, t2); cross(pv.al, r, t);
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; v.ao[i] = pv.ao[i] + t[i] + t2[i]; });
      cross(pv.w, axqd, t);
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; v.al[i] = pv.al[i] + t[i]; v.w[i] = pv.w[i] + axqd[i]; });
    }
  }
  // Contribution of one link to the composite sums about the pelvis origin: h = m c (3), J = I_world + m (c.c 1 - c c') (6),
  // inertial force F = m (a_c - g) (3) and its moment about the pelvis origin T = N + c x F (3), N = I alpha + w x I w.
  static LEG_FN void link_contrib(const Frame& f, const Vel& v, D m, const D (&ipos)[3], const D (&I6)[6], D (&o)[15]) {
    D rc[3], c[3];
    mv(f.mat, ipos, rc);
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; c[i] = f.pos[i] + rc[i]; });
    // I_world = mat I mat'
    D tm[9], Iw[6];
    lfor<0, 3>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      tm[3 * i + 0] = f.mat[3 * i] * I6[0] + f.mat[3 * i + 1] * I6[3] + f.mat[3 * i + 2] * I6[4];
      tm[3 * i + 1] = f.mat[3 * i] * I6[3] + f.mat[3 * i + 1] * I6[1] + f.mat[3 * i + 2] * I6[5];
      tm[3 * i + 2] = f.mat[3 * i] * I6[4] + f.mat[3 * i + 1] * I6[5] + f.mat[3 * i + 2] * I6[2];
    });
    auto rowdot = [&](auto ii, auto jj) { constexpr int i = decltype(ii)::value, j = decltype(jj)::value;
      return tm[3 * i] * f.mat[3 * j] + tm[3 * i + 1] * f.mat[3 * j + 1] + tm[3 * i + 2] * f.mat[3 * j + 2]; };
    Iw[0] = rowdot(LI<0>{}, LI<0>{}); Iw[1] = rowdot(LI<1>{}, LI<1>{}); Iw[2] = rowdot(LI<2>{}, LI<2>{});
    Iw[3] = rowdot(LI<0>{}, LI<1>{}); Iw[4] = rowdot(LI<0>{}, LI<2>{}); Iw[5] = rowdot(LI<1>{}, LI<2>{});
    D t[3], t2[3], ac[3], F[3], Ial[3], Iww[3], N[3];
    cross(v.w, rc, t); cross(v.w, t, t2); cross(v.al, rc, t);
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; ac[i] = v.ao[i] + t[i] + t2[i]; });
    ac[2] = ac[2] - GRAVITY_Z;
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; F[i] = m * ac[i]; });
    sym_mv(Iw, v.al, Ial); sym_mv(Iw, v.w, Iww); cross(v.w, Iww, t);
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; N[i] = Ial[i] + t[i]; });
    const D c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    cross(c, F, t);
    o[0] = m * c[0]; o[1] = m * c[1]; o[2] = m * c[2];
    o[3] = Iw[0] + m * (c2 - c[0] * c[0]); o[4] = Iw[1] + m * (c2 - c[1] * c[1]); o[5] = Iw[2] + m * (c2 - c[2] * c[2]);
    o[6] = Iw[3] - m * c[0] * c[1]; o[7] = Iw[4] - m * c[0] * c[2]; o[8] = Iw[5] - m * c[1] * c[2];
    o[9] = F[0]; o[10] = F[1]; o[11] = F[2];
    o[12] = N[0] + t[0]; o[13] = N[1] + t[1]; o[14] = N[2] + t[2];
  }

  // ------------------------------------------------------------------------------------------------ mass matrix blocks, bias
  struct Mass { D Ls[28], C[7][6], Bb[21], biasb[6], biasl[7]; };
  // leg dof i is an ancestor-or-self of leg dof j (0 hip roll .. 5 toe along the chain; 6 = achilles rod, a child of the thigh)
  static constexpr bool anc(int i, int j) { return i == j || (j <= 5 && i < j) || (j == 6 && i <= 2); }

  // First two passes of a substep: kinematics with velocities down the leg, composite sums back up it.  Leaves sin / cos of the
  // joint angles in sn, cs (the row pass runs the frames again without recomputing them).
  static LEG_FN void mass_bias(typename B::Lds& lds, const Lane& st, KP_ K, Mass& mm, D (&sn)[7], D (&cs)[7]) {
    const M all = M(true);
    Frame f0; Vel v0;
    base_frame(st, f0, v0);
    lfor<0, 7>([&](auto jj) { constexpr int J = decltype(jj)::value; B::sincos(st.ql[J] - kc(K, LK3_DOF_REF + J), sn[J], cs[J]); });
    auto do_link = [&](auto jj, const Frame& pf, const Vel& pv, Frame& f, Vel& v) {   // J = 1..7
      constexpr int J = decltype(jj)::value;
      D ax[3];
      child_frame<J, true>(K, pf, pv, sn[J - 1], cs[J - 1], st.vl[J - 1], f, v, ax);
      D ip[3], I6[6], o[15];
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; ip[i] = kc(K, LK3_IPOS + 3 * (J - 1) + i); });
      lfor<0, 6>([&](auto ii) { constexpr int i = decltype(ii)::value; I6[i] = kc(K, LK3_INERTIA + 6 * (J - 1) + i); });
      link_contrib(f, v, kc(K, LK3_MASS + J - 1), ip, I6, o);
      lfor<0, 15>([&](auto ii) { constexpr int i = decltype(ii)::value; lds.st(T3_LINK + 15 * (J - 1) + i, o[i], all); });
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; lds.st(T3_DOF + 6 * (J - 1) + i, ax[i], all); lds.st(T3_DOF + 6 * (J - 1) + 3 + i, f.pos[i], all); });
    };
    {
      Frame fa, fb, f3; Vel va, vb, v3;
      do_link(LI<1>{}, f0, v0, fa, va);
      do_link(LI<2>{}, fa, va, fb, vb);
      do_link(LI<3>{}, fb, vb, f3, v3);
      do_link(LI<7>{}, f3, v3, fa, va);
      do_link(LI<4>{}, f3, v3, fa, va);
      do_link(LI<5>{}, fa, va, fb, vb);
      do_link(LI<6>{}, fb, vb, fa, va);
    }
    B::fence();
    // ---- back up the tree: composite (mass, h, J) and inertial (F, T) sums of each hinge's subtree; M entries from the momentum
    // vectors of the subtree moving with hinge j:  P_j = J om_j + h x vv_j,  Q_j = m vv_j + om_j x h,  M_ij = om_i . P_j + vv_i . Q_j
    D cm = 0.0, acc[15], rodm = 0.0, rod[15];
    lfor<0, 15>([&](auto ii) { constexpr int i = decltype(ii)::value; acc[i] = 0.0; rod[i] = 0.0; });
    auto add_link = [&](auto jj, D& m_, D (&a)[15]) {
      constexpr int J = decltype(jj)::value;
      m_ = m_ + kc(K, LK3_MASS + J - 1);
      lfor<0, 15>([&](auto ii) { constexpr int i = decltype(ii)::value; a[i] = a[i] + lds.ld(T3_LINK + 15 * (J - 1) + i); });
    };
    auto dof_vec = [&](auto kk, D (&om)[3], D (&vv)[3], D (&an)[3]) {
      constexpr int Kd = decltype(kk)::value;
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; om[i] = lds.ld(T3_DOF + 6 * Kd + i); an[i] = lds.ld(T3_DOF + 6 * Kd + 3 + i); });
      cross(an, om, vv);
    };
    auto do_dof = [&](auto kk, D m_, const D (&a)[15]) {   // leg dof Kd with the sums of its subtree
      constexpr int Kd = decltype(kk)::value;
      D om[3], vv[3], an[3], P[3], Q[3], t[3];
      dof_vec(kk, om, vv, an);
      const D h[3] = {a[0], a[1], a[2]}, Jc[6] = {a[3], a[4], a[5], a[6], a[7], a[8]}, Fs[3] = {a[9], a[10], a[11]}, Ts[3] = {a[12], a[13], a[14]};
      sym_mv(Jc, om, P); cross(h, vv, t);
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; P[i] = P[i] + t[i]; });
      cross(om, h, t);
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; Q[i] = m_ * vv[i] + t[i]; });
      // bias: moment of the subtree's inertial forces about the hinge anchor, along the axis
      cross(an, Fs, t);
      mm.biasl[Kd] = om[0] * (Ts[0] - t[0]) + om[1] * (Ts[1] - t[1]) + om[2] * (Ts[2] - t[2]);
      lfor<0, 7>([&](auto ii) {
        constexpr int Ii = decltype(ii)::value;
        if constexpr (anc(Ii, Kd)) {
          D val;
          if constexpr (Ii == Kd) val = dot(om, P) + dot(vv, Q) + kc(K, LK3_ARMATURE + Kd);
          else { D oi[3], vi[3], ai[3]; dof_vec(ii, oi, vi, ai); val = dot(oi, P) + dot(vi, Q); }
          mm.Ls[symidx(7, Ii, Kd)] = val;
        } else if constexpr (Ii < Kd) mm.Ls[symidx(7, Ii, Kd)] = 0.0;   // chain dofs 3..5 against the rod: different branches
      });
      // coupling with the base: slides along world axes (om = 0, vv = e_b), rotations about the pelvis frame's axes through its origin
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; mm.C[Kd][Bc] = Q[Bc]; });
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; mm.C[Kd][3 + Bc] = f0.mat[Bc] * P[0] + f0.mat[3 + Bc] * P[1] + f0.mat[6 + Bc] * P[2]; });
    };
    add_link(LI<6>{}, cm, acc); do_dof(LI<5>{}, cm, acc);
    add_link(LI<5>{}, cm, acc); do_dof(LI<4>{}, cm, acc);
    add_link(LI<4>{}, cm, acc); do_dof(LI<3>{}, cm, acc);
    add_link(LI<7>{}, rodm, rod); do_dof(LI<6>{}, rodm, rod);
    cm = cm + rodm;
    lfor<0, 15>([&](auto ii) { constexpr int i = decltype(ii)::value; acc[i] = acc[i] + rod[i]; });
    add_link(LI<3>{}, cm, acc); do_dof(LI<2>{}, cm, acc);
    add_link(LI<2>{}, cm, acc); do_dof(LI<1>{}, cm, acc);
    add_link(LI<1>{}, cm, acc); do_dof(LI<0>{}, cm, acc);
    // ---- whole body: own leg + partner's leg (exchange) + pelvis
    D tot[15], pel[15];
    {
      D ip[3], I6[6];
      lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; ip[i] = ldc(&c3_link_ipos[0][0], i); });
      I6[0] = ldc(&c3_link_inertia[0][0][0], 0); I6[1] = ldc(&c3_link_inertia[0][0][0], 4); I6[2] = ldc(&c3_link_inertia[0][0][0], 8);
      I6[3] = ldc(&c3_link_inertia[0][0][0], 1); I6[4] = ldc(&c3_link_inertia[0][0][0], 2); I6[5] = ldc(&c3_link_inertia[0][0][0], 5);
      link_contrib(f0, v0, ldc(c3_link_mass, 0), ip, I6, pel);
    }
    const D tm = (cm + B::swap(cm)) + ldc(c3_link_mass, 0);
    lfor<0, 15>([&](auto ii) { constexpr int i = decltype(ii)::value; tot[i] = (acc[i] + B::swap(acc[i])) + pel[i]; });
    const D h[3] = {tot[0], tot[1], tot[2]}, Jc[6] = {tot[3], tot[4], tot[5], tot[6], tot[7], tot[8]};
    D col[3][3], Ja[3][3], ah[3][3];   // pelvis axes a_k (columns of its frame), J a_k, a_k x h
    lfor<0, 3>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      col[k][0] = f0.mat[k]; col[k][1] = f0.mat[3 + k]; col[k][2] = f0.mat[6 + k];
      sym_mv(Jc, col[k], Ja[k]); cross(col[k], h, ah[k]);
    });
    lfor<0, 3>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      lfor<i, 3>([&](auto jj) { constexpr int j = decltype(jj)::value; mm.Bb[symidx(6, i, j)] = i == j ? tm : D(0.0); });
      lfor<0, 3>([&](auto kk) { constexpr int k = decltype(kk)::value; mm.Bb[symidx(6, i, 3 + k)] = ah[k][i]; });
      lfor<i, 3>([&](auto jj) { constexpr int j = decltype(jj)::value; mm.Bb[symidx(6, 3 + i, 3 + j)] = dot(col[i], Ja[j]); });
      mm.biasb[i] = tot[9 + i];
      mm.biasb[3 + i] = col[i][0] * tot[12] + col[i][1] * tot[13] + col[i][2] * tot[14];
    });
  }

  // ------------------------------------------------------------------------------------------------ block factorisation
  template <int N> static LEG_FN void sym_inverse(D (&a)[N * (N + 1) / 2]) {   // in place, symmetric Gauss-Jordan, no pivoting (SPD)
    lfor<0, N>([&](auto kk) {
      constexpr int Kp = decltype(kk)::value;
      const D p = B::rcp(a[symidx(N, Kp, Kp)]);
      D col[N];
      lfor<0, N>([&](auto ii) { constexpr int Ii = decltype(ii)::value; col[Ii] = a[symidx(N, Ii, Kp)]; });
      lfor<0, N>([&](auto ii) {
        constexpr int Ii = decltype(ii)::value;
        if constexpr (Ii != Kp) {
          const D t = col[Ii] * p;
          lfor<Ii, N>([&](auto jj) { constexpr int Jj = decltype(jj)::value; if constexpr (Jj != Kp) a[symidx(N, Ii, Jj)] = a[symidx(N, Ii, Jj)] - t * col[Jj]; });
        }
      });
      lfor<0, N>([&](auto ii) { constexpr int Ii = decltype(ii)::value; if constexpr (Ii != Kp) a[symidx(N, Ii, Kp)] = -(col[Ii] * p); });
      a[symidx(N, Kp, Kp)] = -p;
    });
    lfor<0, N * (N + 1) / 2>([&](auto ii) { constexpr int Ii = decltype(ii)::value; a[Ii] = -a[Ii]; });
  }
  struct Fact { D Li[28], Y[7][6], G[21]; };   // L^-1 (packed symmetric), Y = L^-1 C, G = F^-1 (packed lower, S = F F')

  static LEG_FN void factor(const Mass& mm, Fact& fc) {
    lfor<0, 28>([&](auto ii) { constexpr int Ii = decltype(ii)::value; fc.Li[Ii] = mm.Ls[Ii]; });
    sym_inverse<7>(fc.Li);
    lfor<0, 7>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      lfor<0, 6>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        D a = 0.0;
        lfor<0, 7>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += fc.Li[symidx(7, Ii, Jj)] * mm.C[Jj][Bc]; });
        fc.Y[Ii][Bc] = a;
      });
    });
    D S[21];
    lfor<0, 6>([&](auto aa) {
      constexpr int A_ = decltype(aa)::value;
      lfor<A_, 6>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        D a = 0.0;
        lfor<0, 7>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += mm.C[Jj][A_] * fc.Y[Jj][Bc]; });
        S[symidx(6, A_, Bc)] = mm.Bb[symidx(6, A_, Bc)] - (a + B::swap(a));   // own + partner: commutative, identical on both lanes
      });
    });
    // Cholesky S = F F' (lower), then G = F^-1
    D F[21], inv[6];
    lfor<0, 6>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      D s = S[symidx(6, j, j)];
      lfor<0, j>([&](auto kk) { constexpr int k = decltype(kk)::value; s = s - F[lowidx(j, k)] * F[lowidx(j, k)]; });
      F[lowidx(j, j)] = B::sqrt(s);
      inv[j] = B::rcp(F[lowidx(j, j)]);
      lfor<j + 1, 6>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        D t = S[symidx(6, j, i)];
        lfor<0, j>([&](auto kk) { constexpr int k = decltype(kk)::value; t = t - F[lowidx(i, k)] * F[lowidx(j, k)]; });
        F[lowidx(i, j)] = t * inv[j];
      });
    });
    lfor<0, 6>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      fc.G[lowidx(j, j)] = inv[j];
      lfor<j + 1, 6>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        D t = 0.0;
        lfor<j, i>([&](auto kk) { constexpr int k = decltype(kk)::value; t = t + F[lowidx(i, k)] * fc.G[lowidx(k, j)]; });
        fc.G[lowidx(i, j)] = -(t * inv[i]);
      });
    });
  }
  static LEG_FN void Gmul(const Fact& fc, const D (&u)[6], D (&o)[6]) {
    lfor<0, 6>([&](auto ii) { constexpr int i = decltype(ii)::value; D a = 0.0;
      lfor<0, i + 1>([&](auto jj) { constexpr int j = decltype(jj)::value; a += fc.G[lowidx(i, j)] * u[j]; }); o[i] = a; });
  }
  static LEG_FN void GTmul(const Fact& fc, const D (&t)[6], D (&o)[6]) {
    lfor<0, 6>([&](auto jj) { constexpr int j = decltype(jj)::value; D a = 0.0;
      lfor<j, 6>([&](auto ii) { constexpr int i = decltype(ii)::value; a += fc.G[lowidx(i, j)] * t[i]; }); o[j] = a; });
  }
  // x = M^-1 g for g = (gb identical on both lanes, gl own leg)
  static LEG_FN void minv_apply(const Fact& fc, const D (&gb)[6], const D (&gl)[7], D (&xb)[6], D (&xl)[7]) {
    D t[6], gt[6];
    lfor<0, 6>([&](auto bb) {
      constexpr int Bc = decltype(bb)::value;
      D a = 0.0;
      lfor<0, 7>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += fc.Y[Jj][Bc] * gl[Jj]; });
      t[Bc] = gb[Bc] - (a + B::swap(a));
    });
    Gmul(fc, t, gt);
    GTmul(fc, gt, xb);
    lfor<0, 7>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      D a = 0.0;
      lfor<0, 7>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += fc.Li[symidx(7, Ii, Jj)] * gl[Jj]; });
      D y = 0.0;
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; y += fc.Y[Ii][Bc] * xb[Bc]; });
      xl[Ii] = a - y;
    });
  }

  static LEG_FN D impedance(D d0, D d1, D width, D x) {
    const M flat = (d0 == d1) | (width <= MINVAL);
    D xx = B::fabs(x / width);
    D y = B::sel(xx <= 0.5, 2.0 * xx * xx, 1.0 - 2.0 * (1.0 - xx) * (1.0 - xx));
    D r = d0 + y * (d1 - d0);
    r = B::sel(xx >= 1.0, d1, r);
    r = B::sel(xx <= 0.0, d0, r);
    return B::sel(flat, 0.5 * (d0 + d1), r);
  }

  struct SubOut { I niter, nrows; M overflow; };
  constexpr static int DAMPING_SWEEPS = 12;

  // ------------------------------------------------------------------------------------------------ one mj_forward (+ Euler)
  // cu: motor commands of the own leg's five actuators (hip roll, hip yaw, hip pitch, knee, toe), pre-clamp.  `live` masks the
  // environments that take part (identical on the two lanes of an environment); an environment over the row capacity is reported in
  // out.overflow and left untouched.  integrate = false: mj_forward only (Reset).
  static LEG3_SUBSTEP_FN void substep(typename B::Lds& lds, Lane& st, const D (&cu)[5], M live, bool integrate, SubOut& out) {
    const I leg = B::opq(B::leg());
    const KP_ K = B::kbase(leg);
    LEG3_PHASE_BEGIN
    Fact fc;
    D qsb[6], qsl[7], sn[7], cs[7];
    {
      Mass mm;
      mass_bias(lds, st, K, mm, sn, cs);
      LEG3_MARK(0)   // 0 = kinematics, mass matrix, bias
      B::fence();
      D taub[6], taul[7];
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; taub[Bc] = -mm.biasb[Bc]; });
      lfor<0, 7>([&](auto dd) {
        constexpr int Dd = decltype(dd)::value;
        D t = -kc(K, LK3_DAMPING + Dd) * st.vl[Dd] - mm.biasl[Dd];
        if constexpr (Dd <= 3 || Dd == 5) {
          constexpr int A_ = Dd == 5 ? 4 : Dd;
          const D lo = kc(K, LK3_ACT_RANGE + 2 * A_), hi = kc(K, LK3_ACT_RANGE + 2 * A_ + 1);
          const D u = B::sel(cu[A_] < lo, lo, B::sel(cu[A_] > hi, hi, cu[A_]));
          t = t + kc(K, LK3_ACT_GEAR + A_) * u;
        }
        taul[Dd] = t;
      });
      factor(mm, fc);
      B::fence();
      minv_apply(fc, taub, taul, qsb, qsl);
    }
    B::fence();
    LEG3_MARK(1)   // 1 = smooth force, factorisation, M^-1 tau
    // ---- raw rows (Jacobians, position, inverse weight).  Joint limits first (they only need q) and straight into their LDS
    // records: the contacts' records start behind them.  The connect rows stay in registers.
    I nlim = 0, ncon = 0;
    M ovf = live & !live;
    lfor<0, 6>([&](auto jj) {
      constexpr int Jj = decltype(jj)::value;
      const D qd = st.ql[Jj];
      const D dlo = qd - kc(K, LK3_LIM_RANGE + 2 * Jj), dhi = kc(K, LK3_LIM_RANGE + 2 * Jj + 1) - qd;
      const M act = live & ((dlo < 0.0) | (dhi < 0.0));
      if (B::any(act)) {
        const I base = nlim * R3_N + DYN0;
        const M fits = base + R3_N <= NSLOT3;
        ovf = ovf | (act & !fits);
        lds.stv(base + R3_KS, B::sel(dlo < 0.0, D(Jj + 1.0), D(-(Jj + 1.0))), act & fits);
        lds.stv(base + R3_POS, B::sel(dlo < 0.0, dlo, dhi), act & fits); lds.stv(base + R3_INVW, kc(K, LK3_DOF_INVW + Jj), act & fits);
      }
      nlim = nlim + B::toI(act);
    });
    D eq_jb[3][6], eq_jl[3][7], eq_pos[3];
    {
      // second run down the frames (no velocities); hinge axes and anchors stay in registers for the point Jacobians
      Frame f0, fa, fb, f3;
      Vel vd;
      D ax[7][3], an[7][3], a0[3][3];
      base_frame(st, f0, vd);
      lfor<0, 3>([&](auto kk) { constexpr int k = decltype(kk)::value; a0[k][0] = f0.mat[k]; a0[k][1] = f0.mat[3 + k]; a0[k][2] = f0.mat[6 + k]; });
      auto frame = [&](auto jj, const Frame& pf, Frame& f) {
        constexpr int J = decltype(jj)::value;
        child_frame<J, false>(K, pf, vd, sn[J - 1], cs[J - 1], D(0.0), f, vd, ax[J - 1]);
        lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; an[J - 1][i] = f.pos[i]; });
      };
      // Jacobian entry of hinge Kd for a point p and a direction: dir . (axis x (p - anchor))
      auto jent = [&](auto kk, const D (&p)[3], const D (&dir)[3]) {
        constexpr int Kd = decltype(kk)::value;
        const D r[3] = {p[0] - an[Kd][0], p[1] - an[Kd][1], p[2] - an[Kd][2]};
        D c[3];
        cross(ax[Kd], r, c);
        return dot(dir, c);
      };
      // the spheres of link LK (its frame f): candidates CA, CA + 1 (the pelvis: candidate 0 alone, left lane only)
      auto spheres = [&](auto lk_, auto ca_, auto nc_, const Frame& f) {
        constexpr int LK = decltype(lk_)::value, CA = decltype(ca_)::value, NC = decltype(nc_)::value;
        lfor<0, NC>([&](auto cc) {
          constexpr int Cc = CA + decltype(cc)::value;
          D sp[3], r[3], hint[3], hw[3], p[3];
          lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; sp[i] = kc(K, LK3_SPH_POS + 3 * Cc + i); hint[i] = kc(K, LK3_SPH_HINT + 3 * Cc + i); });
          mv(f.mat, sp, r);
          lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; p[i] = f.pos[i] + r[i]; });
          const D rad = kc(K, LK3_SPH_R + Cc);
          const D dist = (st.qp[2] + p[2]) - rad;
          M act = live & (dist < 0.0);
          if constexpr (Cc == 0) act = act & (leg == 0);
          if (B::any(act)) {
            p[2] = p[2] - rad - 0.5 * dist;   // contact point: half-way into the penetration
            // mju_makeFrame with normal +z: first tangent = the hint (capsule axis) minus its normal part (spheres: world y), second = n x t1
            mv(f.mat, hint, hw);
            const M has_hint = !((hint[0] == 0.0) & (hint[1] == 0.0) & (hint[2] == 0.0));
            D tx = B::sel(has_hint, hw[0], D(0.0)), ty = B::sel(has_hint, hw[1], D(1.0));
            const D n = B::sqrt(tx * tx + ty * ty);
            const M tiny = n < MINVAL;
            tx = B::sel(tiny, D(1.0), tx / n); ty = B::sel(tiny, D(0.0), ty / n);
            const I base = nlim * R3_N + ncon * C3_N + DYN0;
            const M fits = base + C3_N <= NSLOT3;
            ovf = ovf | (act & !fits);
            const M wr = act & fits;
            lfor<0, 3>([&](auto oo) {
              constexpr int Cmp = decltype(oo)::value;
              const D dir[3] = {Cmp == 0 ? D(0.0) : (Cmp == 1 ? tx : -ty), Cmp == 0 ? D(0.0) : (Cmp == 1 ? ty : tx), D(Cmp == 0 ? 1.0 : 0.0)};
              D c[3];
              const I rb = base + Cmp * C3_ROW;
              lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; lds.stv(rb + (C3_JB + Bc), dir[Bc], wr); cross(a0[Bc], p, c); lds.stv(rb + (C3_JB + 3 + Bc), dot(dir, c), wr); });
              lfor<0, 7>([&](auto dd) {
                constexpr int Dd = decltype(dd)::value;
                if constexpr (Dd < LK && Dd < 6) lds.stv(rb + (C3_JL + Dd), jent(dd, p, dir), wr);
                else lds.stv(rb + (C3_JL + Dd), D(0.0), wr);
              });
            });
            lds.stv(base + C3_POS, dist, wr); lds.stv(base + C3_INVW, kc(K, LK3_SPH_INVW + Cc), wr);
          }
          ncon = ncon + B::toI(act);
        });
      };
      D p1[3], p2[3];
      spheres(LI<0>{}, LI<0>{}, LI<1>{}, f0);
      frame(LI<1>{}, f0, fa);
      frame(LI<2>{}, fa, fb);
      frame(LI<3>{}, fb, f3);
      spheres(LI<3>{}, LI<1>{}, LI<2>{}, f3);
      frame(LI<7>{}, f3, fa);
      {
        D e1[3], r[3];
        lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; e1[i] = kc(K, LK3_EQ_P1 + i); });
        mv(fa.mat, e1, r);
        lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; p1[i] = fa.pos[i] + r[i]; });
      }
      frame(LI<4>{}, f3, fa);
      spheres(LI<4>{}, LI<3>{}, LI<2>{}, fa);
      frame(LI<5>{}, fa, fb);
      spheres(LI<5>{}, LI<5>{}, LI<2>{}, fb);
      {
        D e2[3], r[3];
        lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; e2[i] = kc(K, LK3_EQ_P2 + i); });
        mv(fb.mat, e2, r);
        lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; p2[i] = fb.pos[i] + r[i]; });
      }
      frame(LI<6>{}, fb, fa);
      spheres(LI<6>{}, LI<7>{}, LI<2>{}, fa);
      // connect rows: rod end (link 7: hinges 0 1 2 6) against the heel-spring anchor on the tarsus (link 5: hinges 0 .. 4)
      lfor<0, 3>([&](auto oo) {
        constexpr int Cmp = decltype(oo)::value;
        const D dir[3] = {D(Cmp == 0 ? 1.0 : 0.0), D(Cmp == 1 ? 1.0 : 0.0), D(Cmp == 2 ? 1.0 : 0.0)};
        const D dp[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        D c[3];
        lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; eq_jb[Cmp][Bc] = 0.0; cross(a0[Bc], dp, c); eq_jb[Cmp][3 + Bc] = dot(dir, c); });
        lfor<0, 3>([&](auto dd) { constexpr int Dd = decltype(dd)::value; eq_jl[Cmp][Dd] = jent(dd, p1, dir) - jent(dd, p2, dir); });
        eq_jl[Cmp][3] = -jent(LI<3>{}, p2, dir); eq_jl[Cmp][4] = -jent(LI<4>{}, p2, dir); eq_jl[Cmp][5] = 0.0; eq_jl[Cmp][6] = jent(LI<6>{}, p1, dir);
        eq_pos[Cmp] = dp[Cmp];
      });
    }
    ovf = ovf | B::swapm(ovf);
    out.overflow = ovf;
    const M go = live & !ovf;
    out.nrows = nlim + ncon * 3 + 3;
    out.nrows = out.nrows + B::swapi(out.nrows);
    B::fence();
    LEG3_MARK(2)   // 2 = raw rows: joint limits, collision, connect
    // ---- finish the rows: impedance, R, reference acceleration, z, u~, diagonal, warm start; c and a~ of the warm start
    const D mu = MU;
    struct KindPar { D kk, bb, d0, d1, w; };
    auto kind_par = [&](D solref0, D solref1, D d0, D d1, D w) {
      KindPar k_;
      const D tc = B::sel(solref0 < 2.0 * H, D(2.0 * H), solref0);
      k_.kk = 1.0 / (d1 * d1 * tc * tc * solref1 * solref1); k_.bb = 2.0 / (d1 * tc);
      k_.d0 = d0; k_.d1 = d1; k_.w = w;
      return k_;
    };
    const KindPar kp_eq = kind_par(kc(K, LK3_EQ_SOLREF), kc(K, LK3_EQ_SOLREF + 1), kc(K, LK3_EQ_SOLIMP), kc(K, LK3_EQ_SOLIMP + 1), kc(K, LK3_EQ_SOLIMP + 2));
    const KindPar kp_lim = kind_par(D(c3_limit_solref[0]), D(c3_limit_solref[1]), D(c3_limit_solimp[0]), D(c3_limit_solimp[1]), D(c3_limit_solimp[2]));
    const KindPar kp_con = kind_par(D(c3_contact_solref[0]), D(c3_contact_solref[1]), D(c3_contact_solimp[0]), D(c3_contact_solimp[1]), D(c3_contact_solimp[2]));
    D c[7], at[6];
    lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; c[Dd] = 0.0; });
    lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; at[Bc] = 0.0; });
    struct RowW { D jb[6], jl[7], z[7], ut[6], R, bv, jar, ad; };
    // from the raw row (w.jb, w.jl, position, inverse weight) everything but the force; tangent: the regulariser is the normal row's
    auto finish = [&](D pos, D invw, const KindPar& kp, bool tangent, D R_in, RowW& w) {
      D vel = 0.0, bq = 0.0, jw = 0.0;
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; vel += w.jb[Bc] * st.vb[Bc]; bq += w.jb[Bc] * qsb[Bc]; jw += w.jb[Bc] * st.wb[Bc]; });
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; vel += w.jl[Dd] * st.vl[Dd]; bq += w.jl[Dd] * qsl[Dd]; jw += w.jl[Dd] * st.wl[Dd]; });
      const D imp = impedance(kp.d0, kp.d1, kp.w, pos);
      D R = (1.0 - imp) / imp * invw;
      R = B::sel(R > MINVAL, R, D(MINVAL));
      w.R = tangent ? R_in : R;
      const D aref = -kp.bb * vel - kp.kk * imp * pos;   // (a tangent row's own position is 0)
      w.bv = bq - aref; w.jar = jw - aref;
      lfor<0, 7>([&](auto ii) {
        constexpr int Ii = decltype(ii)::value;
        D a = 0.0;
        lfor<0, 7>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += fc.Li[symidx(7, Ii, Jj)] * w.jl[Jj]; });
        w.z[Ii] = a;
      });
      D u[6];
      lfor<0, 6>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        D a = 0.0;
        lfor<0, 7>([&](auto jj) { constexpr int Jj = decltype(jj)::value; a += fc.Y[Jj][Bc] * w.jl[Jj]; });
        u[Bc] = w.jb[Bc] - a;
      });
      Gmul(fc, u, w.ut);
      D ad = w.R;
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; ad += w.ut[Bc] * w.ut[Bc]; });
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; ad += w.jl[Dd] * w.z[Dd]; });
      w.ad = ad;
    };
    auto warm = [&](const RowW& w, D f) {   // the row's share of c and a~ for the warm-start force f
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; c[Dd] += w.z[Dd] * f; });
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; at[Bc] += w.ut[Bc] * f; });
    };
    struct EqRow { D jl[7], z[7], ut[6], R, b, ad, ai, f; };
    EqRow eq[3];
    lfor<0, 3>([&](auto ss) {   // connect rows: registers
      constexpr int S = decltype(ss)::value;
      RowW w;
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; w.jb[Bc] = eq_jb[S][Bc]; });
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; w.jl[Dd] = eq_jl[S][Dd]; });
      finish(eq_pos[S], kc(K, LK3_EQ_INVW), kp_eq, false, D(0.0), w);
      const D f = B::sel(go, -(B::rcp(w.R) * w.jar), D(0.0));
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; eq[S].jl[Dd] = w.jl[Dd]; eq[S].z[Dd] = w.z[Dd]; });
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; eq[S].ut[Bc] = w.ut[Bc]; });
      eq[S].R = w.R; eq[S].b = w.bv; eq[S].ad = w.ad; eq[S].ai = B::rcp(w.ad); eq[S].f = f;
      warm(w, f);
      B::fence();
    });
    LEG3_MARK(5)   // 5 = connect rows finished
    for (int j = 0; j < 6; j++) {   // joint limits (compacted: limit j exists only if limit j - 1 does)
      const M valid = go & (nlim > j);
      if (!B::any(valid)) break;
      RowW w;
      const I base = I(DYN0 + j * R3_N);
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; w.jb[Bc] = 0.0; });
      {
        const D ks = B::sel(valid, lds.ldv(base + R3_KS), D(0.0)), ak = B::fabs(ks);
        lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; w.jl[Dd] = B::sel(ak == D(Dd + 1.0), B::sel(ks > 0.0, D(1.0), D(-1.0)), D(0.0)); });
      }
      finish(B::sel(valid, lds.ldv(base + R3_POS), D(0.0)), B::sel(valid, lds.ldv(base + R3_INVW), D(1.0)), kp_lim, false, D(0.0), w);
      const D f = B::sel(valid & (w.jar < 0.0), -(B::rcp(w.R) * w.jar), D(0.0));
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; lds.stv(base + (R3_Z + Dd), w.z[Dd], valid); });
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; lds.stv(base + (R3_UT + Bc), w.ut[Bc], valid); });
      lds.stv(base + R3_R, w.R, valid); lds.stv(base + R3_B, w.bv, valid); lds.stv(base + R3_AD, w.ad, valid);
      lds.stv(base + R3_F, f, valid);
      warm(w, f);
    }
    LEG3_MARK(6)   // 6 = limit rows finished
    for (int p = 0; p < 9; p++) {   // contacts: normal, tangent 1, tangent 2
      const M valid = go & (ncon > p);
      if (!B::any(valid)) break;
      const I base = nlim * R3_N + (DYN0 + p * C3_N);
      RowW wn, w1, w2;
      auto raw = [&](I rb, RowW& w) {
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; w.jb[Bc] = B::sel(valid, lds.ldv(rb + (C3_JB + Bc)), D(0.0)); });
        lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; w.jl[Dd] = B::sel(valid, lds.ldv(rb + (C3_JL + Dd)), D(0.0)); });
      };
      const D dist = B::sel(valid, lds.ldv(base + C3_POS), D(0.0)), invw = B::sel(valid, lds.ldv(base + C3_INVW), D(1.0));
      raw(base, wn); raw(base + C3_ROW, w1); raw(base + 2 * C3_ROW, w2);
      finish(dist, invw, kp_con, false, D(0.0), wn);
      finish(D(0.0), invw, kp_con, true, wn.R, w1);
      finish(D(0.0), invw, kp_con, true, wn.R, w2);
      // 3x3 diagonal block of A (the R of the contact on its diagonal: already in ad)
      auto cross_term = [&](const RowW& a, const RowW& b) {
        D t = 0.0;
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; t += a.ut[Bc] * b.ut[Bc]; });
        lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; t += a.jl[Dd] * b.z[Dd]; });
        return t;
      };
      lds.stv(base + C3_R, wn.R, valid);
      lds.stv(base + (C3_BLK + 0), wn.ad, valid); lds.stv(base + (C3_BLK + 1), cross_term(wn, w1), valid); lds.stv(base + (C3_BLK + 2), cross_term(wn, w2), valid);
      lds.stv(base + (C3_BLK + 3), w1.ad, valid); lds.stv(base + (C3_BLK + 4), cross_term(w1, w2), valid); lds.stv(base + (C3_BLK + 5), w2.ad, valid);
      // warm start of the cone (mj_constraintUpdate): top zone 0, bottom zone -D jar, middle zone on the cone
      const D Dn = B::rcp(wn.R);
      const D Nn = wn.jar * mu, U1 = w1.jar * mu, U2 = w2.jar * mu, Tt = B::sqrt(U1 * U1 + U2 * U2);
      const M top = (Nn >= mu * Tt) | ((Tt <= 0.0) & (Nn >= 0.0));
      const M bot = (mu * Nn + Tt <= 0.0) | ((Tt <= 0.0) & (Nn < 0.0));
      const D Dm = Dn / (mu * mu * (1.0 + mu * mu)), NmT = Nn - mu * Tt;
      const D fnm = -Dm * NmT * mu;
      const D fn = B::sel(valid, B::sel(top, D(0.0), B::sel(bot, -Dn * wn.jar, fnm)), D(0.0));
      const D f1 = B::sel(valid, B::sel(top, D(0.0), B::sel(bot, -Dn * w1.jar, -fnm / Tt * U1 * mu)), D(0.0));
      const D f2 = B::sel(valid, B::sel(top, D(0.0), B::sel(bot, -Dn * w2.jar, -fnm / Tt * U2 * mu)), D(0.0));
      auto put = [&](I rb, const RowW& w, D f) {
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; lds.stv(rb + (C3_UT + Bc), w.ut[Bc], valid); });
        lds.stv(rb + C3_B, w.bv, valid); lds.stv(rb + C3_F, f, valid);
        warm(w, f);
      };
      put(base, wn, fn); put(base + C3_ROW, w1, f1); put(base + 2 * C3_ROW, w2, f2);
    }
    lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; at[Bc] = at[Bc] + B::swap(at[Bc]); });
    LEG3_MARK(7)   // 7 = contact rows finished (+ their 3 x 3 blocks and cone warm start)
    // (A f)_i + b_i pieces shared by the cost of the warm start and the sweeps: two accumulation chains instead of one
    auto dot_c = [&](const D (&jl)[7]) {
      D x = jl[0] * c[0], y = jl[1] * c[1];
      x = B::fma(jl[2], c[2], x); y = B::fma(jl[3], c[3], y); x = B::fma(jl[4], c[4], x); y = B::fma(jl[5], c[5], y); x = B::fma(jl[6], c[6], x);
      return x + y;
    };
    auto dot_a = [&](const D (&ut)[6]) {
      D x = ut[0] * at[0], y = ut[1] * at[1];
      x = B::fma(ut[2], at[2], x); y = B::fma(ut[3], at[3], y); x = B::fma(ut[4], at[4], x); y = B::fma(ut[5], at[5], y);
      return x + y;
    };
    struct CRow { D jl[7], ut[6], b, f; };
    auto cload = [&](I rb, CRow& r) {
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; r.jl[Dd] = lds.ldv(rb + (C3_JL + Dd)); });
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; r.ut[Bc] = lds.ldv(rb + (C3_UT + Bc)); });
      r.b = lds.ldv(rb + C3_B); r.f = lds.ldv(rb + C3_F);
    };
    struct LRow { D z[7], ut[6], R, b, f, jc; };   // jc = jl . c = s c_k
    auto lload = [&](I base, LRow& r) {
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; r.z[Dd] = lds.ldv(base + (R3_Z + Dd)); });
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; r.ut[Bc] = lds.ldv(base + (R3_UT + Bc)); });
      r.R = lds.ldv(base + R3_R); r.b = lds.ldv(base + R3_B); r.f = lds.ldv(base + R3_F);
      const D ks = lds.ldv(base + R3_KS), ak = B::fabs(ks);
      D ck = c[0];
      lfor<1, 6>([&](auto dd) { constexpr int Dd = decltype(dd)::value; ck = B::sel(ak == D(Dd + 1.0), c[Dd], ck); });
      r.jc = B::sel(ks > 0.0, ck, -ck);
    };
    {
      // cost of the warm start, 1/2 f'Af + f'b: kept only if negative
      D cost = 0.0;
      lfor<0, 3>([&](auto ss) { constexpr int S = decltype(ss)::value;
        cost += B::sel(go, eq[S].f * (0.5 * (eq[S].R * eq[S].f + dot_c(eq[S].jl) + dot_a(eq[S].ut)) + eq[S].b), D(0.0)); });
      for (int j = 0; j < 6; j++) {
        const M v = go & (nlim > j);
        if (!B::any(v)) break;
        LRow r; lload(I(DYN0 + j * R3_N), r);
        cost += B::sel(v, r.f * (0.5 * (r.R * r.f + r.jc + dot_a(r.ut)) + r.b), D(0.0));
      }
      for (int p = 0; p < 9; p++) {
        const M v = go & (ncon > p);
        if (!B::any(v)) break;
        const I base = nlim * R3_N + (DYN0 + p * C3_N);
        const D R = lds.ldv(base + C3_R);
        for (int i = 0; i < 3; i++) {
          CRow r; cload(base + i * C3_ROW, r);
          cost += B::sel(v, r.f * (0.5 * (R * r.f + dot_c(r.jl) + dot_a(r.ut)) + r.b), D(0.0));
        }
      }
      cost = cost + B::swap(cost);
      const M drop = cost > 0.0;
      if (B::any(drop)) {
        lfor<0, 3>([&](auto ss) { constexpr int S = decltype(ss)::value; eq[S].f = B::sel(drop, D(0.0), eq[S].f); });
        for (int j = 0; j < 6; j++) { const M v = go & (nlim > j); if (!B::any(v)) break; lds.stv(I(DYN0 + j * R3_N + R3_F), D(0.0), v & drop); }
        for (int p = 0; p < 9; p++) {
          const M v = go & (ncon > p);
          if (!B::any(v)) break;
          const I base = nlim * R3_N + (DYN0 + p * C3_N);
          for (int i = 0; i < 3; i++) lds.stv(base + (i * C3_ROW + C3_F), D(0.0), v & drop);
        }
      }
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; c[Dd] = B::sel(drop, D(0.0), c[Dd]); });
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; at[Bc] = B::sel(drop, D(0.0), at[Bc]); });
    }
    LEG3_MARK(3)   // 3 = cost of the warm start (kept only if negative)
    // ---- PGS sweeps (mj_solPGS, elliptic cones), MuJoCo's row order; a~ = `at` is shared by the two lanes of an environment.
    // A lane that does not own a step executes it all the same with its deltas masked to zero; what it reads from its own LDS slots
    // is whatever it last wrote there (finite: the kernel clears the slots once), so 0 x it cannot poison c or a~.
    I niter = 0;
    {
      const D scale = 1.0 / (MEANINERTIA * NV);
      M sweeping = go;
      const M isL = leg == 0;
      D acc = 0.0;
      auto sync = [&](int w) {
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value;
          at[Bc] = w == 0 ? B::template pair_bcast<0>(at[Bc]) : B::template pair_bcast<1>(at[Bc]); });
      };
      auto apply_z = [&](const D (&z)[7], const D (&ut)[6], D d) {
        lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; c[Dd] = B::fma(z[Dd], d, c[Dd]); });
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; at[Bc] = B::fma(ut[Bc], d, at[Bc]); });
      };
      auto eq_step = [&](auto ss, M mine) {   // a connect row: unclamped
        typename B::OwnerScope scope_(mine);   // op-counting builds of the CPU emulation only (tools/count_flops.py); empty on the device
        constexpr int S = decltype(ss)::value;
        const D res = (B::fma(eq[S].R, eq[S].f, eq[S].b) + dot_c(eq[S].jl)) + dot_a(eq[S].ut);
        D d = -(res * eq[S].ai);
        D chg = d * B::fma(0.5 * eq[S].ad, d, res);
        const M keep = mine & (chg <= 1e-10);
        d = B::sel(keep, d, D(0.0)); chg = B::sel(keep, chg, D(0.0));
        apply_z(eq[S].z, eq[S].ut, d);
        acc = acc + chg;
        eq[S].f = eq[S].f + d;
      };
      // A limit row's LDS record (LRaw) is read one step AHEAD (r05): the step of row j starts from registers while the record of row j + 1 is on
      // its way (another record: the store of f_j does not touch it), instead of every step opening with ~100 cycles of LDS latency.
      struct LRaw { D z[7], ut[6], R, b, f, ks, ad; };
      auto lraw = [&](I base, LRaw& r) {
        lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; r.z[Dd] = lds.ldv(base + (R3_Z + Dd)); });
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; r.ut[Bc] = lds.ldv(base + (R3_UT + Bc)); });
        r.R = lds.ldv(base + R3_R); r.b = lds.ldv(base + R3_B); r.f = lds.ldv(base + R3_F); r.ks = lds.ldv(base + R3_KS); r.ad = lds.ldv(base + R3_AD);
      };
      auto lim_step = [&](I base_, M mine, const LRaw& r) {   // a joint limit: f >= 0; r = the record of `base_` (of slot DYN0 on a bystander lane)
        typename B::OwnerScope scope_(mine);
        LEG3_STAT(1);
        const I base = B::seli(mine, base_, I(DYN0));   // (a bystander stays inside its slots)
        const D ak = B::fabs(r.ks);
        D ck = c[0];
        lfor<1, 6>([&](auto dd) { constexpr int Dd = decltype(dd)::value; ck = B::sel(ak == D(Dd + 1.0), c[Dd], ck); });   // (a select tree instead of this chain: 5.60 -> 5.67 ms per step, not kept)
        const D jc = B::sel(r.ks > 0.0, ck, -ck);
        const D ad = r.ad, ai = B::rcp(ad);
        const D res = (B::fma(r.R, r.f, r.b) + jc) + dot_a(r.ut);
        const D nf = B::fmax(B::fma(-res, ai, r.f), D(0.0));
        D d = nf - r.f;
        D chg = d * B::fma(0.5 * ad, d, res);
        const M keep = mine & (chg <= 1e-10);
        d = B::sel(keep, d, D(0.0)); chg = B::sel(keep, chg, D(0.0));
        apply_z(r.z, r.ut, d);
        acc = acc + chg;
        lds.stv(base + R3_F, r.f + d, keep);
      };
      auto contact = [&](I base_, M mine) {
        typename B::OwnerScope scope_(mine);
        LEG3_STAT(2);
        const I base = B::seli(mine, base_, I(DYN0));   // (a bystander stays inside its slots)
        CRow r0, r1, r2;
        cload(base, r0); cload(base + C3_ROW, r1); cload(base + 2 * C3_ROW, r2);
        const D R = lds.ldv(base + C3_R);
        const D A00 = lds.ldv(base + (C3_BLK + 0)), A01 = lds.ldv(base + (C3_BLK + 1)), A02 = lds.ldv(base + (C3_BLK + 2));
        const D A11 = lds.ldv(base + (C3_BLK + 3)), A12 = lds.ldv(base + (C3_BLK + 4)), A22 = lds.ldv(base + (C3_BLK + 5));
        const D q0 = (B::fma(R, r0.f, r0.b) + dot_c(r0.jl)) + dot_a(r0.ut);
        const D q1 = (B::fma(R, r1.f, r1.b) + dot_c(r1.jl)) + dot_a(r1.ut);
        const D q2 = (B::fma(R, r2.f, r2.b) + dot_c(r2.jl)) + dot_a(r2.ut);
        const D o0 = r0.f, o1 = r1.f, o2 = r2.f;
        // normal-only update (taken when the normal force is ~0)
        const D fn_n = B::fmax(o0 - q0 * B::rcp(A00), D(0.0));
        // ray update: scale the force vector by (1 + x), x clamped so that the normal force stays >= 0
        const D v0 = A00 * o0 + A01 * o1 + A02 * o2, v1_ = A01 * o0 + A11 * o1 + A12 * o2, v2_ = A02 * o0 + A12 * o1 + A22 * o2;
        const D denom = o0 * v0 + o1 * v1_ + o2 * v2_;
        D x = -(o0 * q0 + o1 * q1 + o2 * q2) * B::rcp(denom);
        x = B::fmax(x, D(-1.0));
        x = B::sel(denom >= MINVAL, x, D(0.0));
        const M use_n = o0 < MINVAL;
        const D f0 = B::sel(use_n, fn_n, o0 + x * o0);
        D f1 = B::sel(use_n, D(0.0), o1 + x * o1), f2 = B::sel(use_n, D(0.0), o2 + x * o2);
        {  // friction: QCQP on the cone given the normal force (mju_QCQP2; result used only if f0 >= MINVAL)
          const D bc1 = q1 - (A11 * o1 + A12 * o2) + A01 * (f0 - o0);
          const D bc2 = q2 - (A12 * o1 + A22 * o2) + A02 * (f0 - o0);
          const D b1 = bc1 * mu, b2 = bc2 * mu, Q11 = A11 * (mu * mu), Q22 = A22 * (mu * mu), Q12 = A12 * (mu * mu);
          // Newton iterations on the multiplier la of |v| <= f0, v = -(Q + la)^-1 b, in mju_QCQP2's order (v, test the violation,
          // step, test the step, update; 20 rounds at most, v is the one of the last round) but with everything multiplied through by
          // det = |Q + la|: w = det v = -adj(Q + la) b, val det^2 = w.w - f0^2 det^2, step = -val / val' = (val det^2) det / (2 w'adj w):
          // ONE reciprocal per round on a chain half as long.  Only la is carried per lane: a lane that has stopped keeps its la, so
          // every later round recomputes ITS w and det unchanged and they are still right behind the loop.
          D la = 0.0, w1 = 0.0, w2 = 0.0, det = 1.0;
          M degenerate = mine & !mine;
          M run = mine & (f0 >= MINVAL);
          for (int it = 0; it < 20; it++) {
            if (!B::any(run)) break;
            LEG3_STAT(3);
            const D qa = Q11 + la, qd = Q22 + la;
            det = qa * qd - Q12 * Q12;
            const M sing = run & (det < 1e-10);
            degenerate = degenerate | sing;
            run = run & !sing;
            w1 = Q12 * b2 - qd * b1; w2 = Q12 * b1 - qa * b2;
            const D det2 = det * det;
            const D num = (w1 * w1 + w2 * w2) - (f0 * f0) * det2;
            run = run & !(num < 1e-10 * det2);   // inside the cone (la = 0) or on it: done
            const D den = 2.0 * (qd * w1 * w1 - 2.0 * Q12 * w1 * w2 + qa * w2 * w2);
            const D delta = num * det * B::rcp(den);
            run = run & !(delta < 1e-10);
            la = B::sel(run, la + delta, la);
          }
          const D di = B::rcp(det);
          const D v1 = w1 * di, v2 = w2 * di;
          D g1 = B::sel(degenerate, D(0.0), v1 * mu), g2 = B::sel(degenerate, D(0.0), v2 * mu);
          {  // active constraint: put the friction exactly on the cone
            D s = (g1 * g1 + g2 * g2) * (1.0 / (MU * MU));
            s = B::sqrt(f0 * f0 * B::rcp(B::sel(s > MINVAL, s, D(MINVAL))));
            const M oncone = !(la == 0.0) & !degenerate;
            g1 = B::sel(oncone, g1 * s, g1); g2 = B::sel(oncone, g2 * s, g2);
          }
          const M fr = f0 >= MINVAL;
          f1 = B::sel(fr, g1, f1); f2 = B::sel(fr, g2, f2);
        }
        D d0 = f0 - o0, d1 = f1 - o1, d2 = f2 - o2;
        D chg = 0.5 * (d0 * (A00 * d0 + A01 * d1 + A02 * d2) + d1 * (A01 * d0 + A11 * d1 + A12 * d2) + d2 * (A02 * d0 + A12 * d1 + A22 * d2)) +
                d0 * q0 + d1 * q1 + d2 * q2;
        const M keep = mine & (chg <= 1e-10);
        d0 = B::sel(keep, d0, D(0.0)); d1 = B::sel(keep, d1, D(0.0)); d2 = B::sel(keep, d2, D(0.0)); chg = B::sel(keep, chg, D(0.0));
        // c += L^-1 (jl_0 d0 + jl_1 d1 + jl_2 d2), a~ += u~_0 d0 + u~_1 d1 + u~_2 d2
        D gw[7];
        lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; gw[Dd] = B::fma(r2.jl[Dd], d2, B::fma(r1.jl[Dd], d1, r0.jl[Dd] * d0)); });
        lfor<0, 7>([&](auto ii) {
          constexpr int Ii = decltype(ii)::value;
          D xa = fc.Li[symidx(7, Ii, 0)] * gw[0], xb = fc.Li[symidx(7, Ii, 1)] * gw[1];
          xa = B::fma(fc.Li[symidx(7, Ii, 2)], gw[2], xa); xb = B::fma(fc.Li[symidx(7, Ii, 3)], gw[3], xb);
          xa = B::fma(fc.Li[symidx(7, Ii, 4)], gw[4], xa); xb = B::fma(fc.Li[symidx(7, Ii, 5)], gw[5], xb);
          xa = B::fma(fc.Li[symidx(7, Ii, 6)], gw[6], xa);
          c[Ii] = c[Ii] + (xa + xb);
        });
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; at[Bc] = B::fma(r2.ut[Bc], d2, B::fma(r1.ut[Bc], d1, B::fma(r0.ut[Bc], d0, at[Bc]))); });
        acc = acc + chg;
        lds.stv(base + C3_F, o0 + d0, keep); lds.stv(base + (C3_ROW + C3_F), o1 + d1, keep); lds.stv(base + (2 * C3_ROW + C3_F), o2 + d2, keep);
      };
      for (int iter = 0; iter < LEG3_ITERS; iter++) {
        if (!B::any(sweeping)) break;
        LEG3_STAT(0);
        acc = 0.0;
        for (int w = 0; w < 2; w++) {
          const M side = (w == 0 ? isL : !isL) & sweeping;
          eq_step(LI<0>{}, side); eq_step(LI<1>{}, side); eq_step(LI<2>{}, side);
          sync(w);
        }
        LEG3_SW_MARK(0)   // (inside the sweeps: 0 = connect steps, 1 = limit steps, 2 = contact steps + the stopping test)
        for (int w = 0; w < 2; w++) {
          const M side = (w == 0 ? isL : !isL) & sweeping;
          if (!B::any(side & (nlim > 0))) continue;
          LRaw cur;
          lraw(I(DYN0), cur);   // row 0 (a lane without limits reads its slot DYN0 too: masked below)
          for (int j = 0; j < 6; j++) {
            const M mine = side & (nlim > j);
            if (!B::any(mine)) break;
            LRaw nxt = cur;
            if (j < 5) lraw(B::seli(side & (nlim > j + 1), I(DYN0 + (j + 1) * R3_N), I(DYN0)), nxt);
            lim_step(I(DYN0 + j * R3_N), mine, cur);
            cur = nxt;
          }
          sync(w);
        }
        LEG3_SW_MARK(1)
        for (int w = 0; w < 2; w++) {
          const M side = (w == 0 ? isL : !isL) & sweeping;
          if (!B::any(side & (ncon > 0))) continue;
          for (int p = 0; p < 9; p++) {
            const M mine = side & (ncon > p);
            if (!B::any(mine)) break;
            contact(nlim * R3_N + (DYN0 + p * C3_N), mine);
          }
          sync(w);
        }
        const D improvement = -(acc + B::swap(acc));
        niter = niter + B::toI(sweeping);
        sweeping = sweeping & !(improvement * scale < TOLERANCE);
        LEG3_SW_MARK(2)
      }
    }
    out.niter = niter;
    B::fence();
    LEG3_SW_FLUSH
    // ---- qacc = qacc_smooth + M^-1 J' f = qs + [G' a~ ; c - Y G' a~]; implicit joint damping of mj_Euler as in cassie_leg_core.h:
    // (M + h B) qacc' = M qacc  <=>  qacc' = (I + E)^-1 qacc, E = M^-1 h B a contraction (tests/test_implicit_damping_bound.py)
    D xb[6], xl[7];
    {
      D gx[6];
      GTmul(fc, at, gx);
      lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; xb[Bc] = qsb[Bc] + gx[Bc]; });
      lfor<0, 7>([&](auto ii) {
        constexpr int Ii = decltype(ii)::value;
        D y = 0.0;
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; y += fc.Y[Ii][Bc] * gx[Bc]; });
        xl[Ii] = qsl[Ii] + (c[Ii] - y);
      });
    }
    D hb[6], hl[7];
    lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; hb[Bc] = xb[Bc]; });
    lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; hl[Dd] = xl[Dd]; });
    if (integrate) {
      const D zb[6] = {D(0.0), D(0.0), D(0.0), D(0.0), D(0.0), D(0.0)};
      D hdamp[7];
      lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; hdamp[Dd] = H * kc(K, LK3_DAMPING + Dd); });
      for (int it = 0; it < DAMPING_SWEEPS; it++) {
        D dl[7], eb[6], el[7];
        lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; dl[Dd] = hdamp[Dd] * hl[Dd]; });
        minv_apply(fc, zb, dl, eb, el);
        lfor<0, 6>([&](auto bb) { constexpr int Bc = decltype(bb)::value; hb[Bc] = xb[Bc] - eb[Bc]; });
        lfor<0, 7>([&](auto dd) { constexpr int Dd = decltype(dd)::value; hl[Dd] = xl[Dd] - el[Dd]; });
      }
    }
    lfor<0, 6>([&](auto bb) {
      constexpr int Bc = decltype(bb)::value;
      st.wb[Bc] = B::sel(go, xb[Bc], st.wb[Bc]);
      if (integrate) st.vb[Bc] = B::sel(go, st.vb[Bc] + H * hb[Bc], st.vb[Bc]);
    });
    lfor<0, 7>([&](auto dd) {
      constexpr int Dd = decltype(dd)::value;
      st.wl[Dd] = B::sel(go, xl[Dd], st.wl[Dd]);
      if (integrate) {
        const D vn = st.vl[Dd] + H * hl[Dd];
        st.vl[Dd] = B::sel(go, vn, st.vl[Dd]);
        st.ql[Dd] = B::sel(go, st.ql[Dd] + H * vn, st.ql[Dd]);
      }
    });
    if (integrate) {
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; st.qp[Bc] = B::sel(go, st.qp[Bc] + H * st.vb[Bc], st.qp[Bc]); });
      // mju_quatIntegrate: quat <- normalize(quat) * axisangle(omega_body, h |omega|)
      const D wx = st.vb[3], wy = st.vb[4], wz = st.vb[5], wn = B::sqrt(wx * wx + wy * wy + wz * wz);
      const M spin = wn >= MINVAL;
      const D ax = B::sel(spin, wx / wn, D(1.0)), ay = B::sel(spin, wy / wn, D(0.0)), az = B::sel(spin, wz / wn, D(0.0));
      const D ang = B::sel(spin, H * wn, D(0.0));
      D sh, r0;
      B::sincos(0.5 * ang, sh, r0);
      const D r1 = ax * sh, r2 = ay * sh, r3 = az * sh;
      const D n = B::sqrt(st.qq[0] * st.qq[0] + st.qq[1] * st.qq[1] + st.qq[2] * st.qq[2] + st.qq[3] * st.qq[3]);
      const D a0 = st.qq[0] / n, a1 = st.qq[1] / n, a2 = st.qq[2] / n, a3 = st.qq[3] / n;
      st.qq[0] = B::sel(go, a0 * r0 - a1 * r1 - a2 * r2 - a3 * r3, st.qq[0]);
      st.qq[1] = B::sel(go, a0 * r1 + a1 * r0 + a2 * r3 - a3 * r2, st.qq[1]);
      st.qq[2] = B::sel(go, a0 * r2 - a1 * r3 + a2 * r0 + a3 * r1, st.qq[2]);
      st.qq[3] = B::sel(go, a0 * r3 + a1 * r2 - a2 * r1 + a3 * r0, st.qq[3]);
    }
    LEG3_MARK(4)   // 4 = qacc, implicit damping, integration
  }

  // ------------------------------------------------------------------------------------------------ HBM <-> lane, n_sub substeps
  struct Io {
    typename B::P rec, act;   // the lane's state record [ENV3_STRIDE], its environment's action row [NU] (or the record's ctrl)
    bool has_act;
  };
  struct Out { I pend, niter, nrows; };

  // n_sub torque-mode substeps (Cassie2d::Step semantics on the 3-D mechanism).  On return o.pend = substeps NOT done because the
  // environment left the row capacity (its state is untouched from that substep on; the next kernel tier finishes it).
  static LEG_FN void env_step(typename B::Lds& lds, const Io& io, M valid, int n_sub, bool integrate, Out& o) {
    const I leg = B::leg();
    const I lq = leg * 7 + 7, lv = leg * 7 + 6, la = leg * 5;
    const M left = leg == 0;
    Lane st;
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; st.qp[i] = B::pld(io.rec, I(E3_Q + i)); });
    lfor<0, 4>([&](auto ii) { constexpr int i = decltype(ii)::value; st.qq[i] = B::pld(io.rec, I(E3_Q + 3 + i)); });
    lfor<0, 6>([&](auto ii) { constexpr int i = decltype(ii)::value; st.vb[i] = B::pld(io.rec, I(E3_V + i)); st.wb[i] = B::pld(io.rec, I(E3_WS + i)); });
    lfor<0, 7>([&](auto ii) { constexpr int i = decltype(ii)::value;
      st.ql[i] = B::pld(io.rec, lq + (E3_Q + i)); st.vl[i] = B::pld(io.rec, lv + (E3_V + i)); st.wl[i] = B::pld(io.rec, lv + (E3_WS + i)); });
    D cu[5];
    lfor<0, 5>([&](auto aa) { constexpr int A_ = decltype(aa)::value; cu[A_] = io.has_act ? B::pld(io.act, la + A_) : B::pld(io.rec, la + (E3_CTRL + A_)); });
    M live = valid;
    o.pend = 0; o.niter = 0; o.nrows = 0;
    D time = B::pld(io.rec, I(E3_TIME));
    SubOut so;
    for (int sub = 0; sub < n_sub; sub++) {
      if (!B::any(live)) break;
      substep(lds, st, cu, live, integrate, so);
      const M ovf = live & so.overflow;
      o.pend = B::seli(ovf, I(n_sub - sub), o.pend);
      live = live & !ovf;
      o.niter = o.niter + B::seli(live, so.niter, I(0));
      o.nrows = B::seli(live, so.nrows, o.nrows);
      if (integrate) time = B::sel(live, time + H, time);
    }
    const M wr = valid;
    lfor<0, 3>([&](auto ii) { constexpr int i = decltype(ii)::value; B::pst(io.rec, I(E3_Q + i), st.qp[i], wr & left); });
    lfor<0, 4>([&](auto ii) { constexpr int i = decltype(ii)::value; B::pst(io.rec, I(E3_Q + 3 + i), st.qq[i], wr & left); });
    lfor<0, 6>([&](auto ii) { constexpr int i = decltype(ii)::value; B::pst(io.rec, I(E3_V + i), st.vb[i], wr & left); B::pst(io.rec, I(E3_WS + i), st.wb[i], wr & left); });
    lfor<0, 7>([&](auto ii) { constexpr int i = decltype(ii)::value;
      B::pst(io.rec, lq + (E3_Q + i), st.ql[i], wr); B::pst(io.rec, lv + (E3_V + i), st.vl[i], wr); B::pst(io.rec, lv + (E3_WS + i), st.wl[i], wr); });
    lfor<0, 5>([&](auto aa) { constexpr int A_ = decltype(aa)::value; B::pst(io.rec, la + (E3_CTRL + A_), cu[A_], wr); });
    B::pst(io.rec, I(E3_TIME), time, wr & left);
  }
};

}  // namespace leg
}  // namespace cassie3d
#endif
