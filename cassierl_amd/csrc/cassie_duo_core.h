// cassie_duo_core.h -- Cassie2d Env.step with 64 ENVIRONMENTS PER WAVEFRONT: the set-up of cassie_leg_core.h (one lane per leg) for two
// groups of 32 environments, ONE joint Gauss-Seidel sweep with a lane per ENVIRONMENT, the two finishes (r05).
//
// Why.  In the two-lanes-per-environment kernel a Gauss-Seidel step is executed by both lanes of an environment and kept by the leg
// that owns the row: half of the wavefront's lanes do nothing useful in the PGS sweeps, which are 74 % of that kernel's time (r03/r04
// PMC: issued FP64 0.35 of peak, useful 0.18).  An environment's sweep is one dependent chain (MuJoCo's row order: connect L,
// connect R, contacts L, contacts R -- every block waits for the previous one through the shared 3-vector a~), so the instruction
// stream of a sweep cannot be shortened; what can change is how many environments one instruction serves.  Here the lane that was
// "left leg of environment e of group A" becomes "environment e of group A" for the duration of the sweeps (it takes the right leg's
// row data from its partner lane by one DPP exchange per word), and its partner lane becomes "environment e of group B": the same
// ~500 instructions per sweep now advance 64 environments instead of 32, every lane the owner of every step it executes, no a~
// exchange at all (a~ is a lane-local 3-vector).
//
//   per substep:   set-up(A)  set-up(B)  | transpose (DPP, ~130 doubles per lane) | joint sweeps (<= 50) | forces back | finish(A) finish(B)
//
// The phases are INDEPENDENT: everything a group carries from its set-up to the joint sweep and on to its finish (state, rows, factorisation,
// forces) goes through a per-wavefront workspace in global memory, stored at the end of a phase and loaded in one batch at the head of the next
// (the set-up needs the whole register file; see "hand-over" below and DESIGN.md section 5 K1d for the builds that tried otherwise).  Set-up and
// finish exist once, in a loop over the two groups.
//
// The arithmetic of an environment is the arithmetic of cassie_leg_core.h operation for operation: the set-up and the finish ARE its
// functions (`sub_setup`, `sub_finish`), the joint sweep executes the owner lane's sequence of the pair sweep (same step functions
// written for one lane, same order, same fused multiply-adds), and the cost accumulators of the two legs are kept apart and added in
// the order the pair sweep adds them -- so the state trajectories are BIT-IDENTICAL to the two-lanes-per-environment kernel
// (tests/test_gpu_duo.py; on the CPU: tests/test_duo_host.py through oracle/leg_host).  The translation units that instantiate the two cores are
// compiled with -ffp-contract=on for that (cassierl_amd/build.py: UNIT_FLAGS).
//
// Capacity of the joint sweep: six rows per leg (2 connect + 2 contact pairs: robots on their feet, the bench's and TRPO's regime).
// A group in which some environment has a joint limit active or a third contact pair on a leg runs the pair sweep of
// cassie_leg_core.h for that substep (eight rows per leg), exactly as the two-lanes kernel would; environments over eight rows per
// leg are handed to the lower tiers through `pending` as before.
//
// Storage.  Two groups share one wavefront's LDS budget (40 KB at four wavefronts per CU), so the per-lane cold block shrinks to what a
// substep itself produces and consumes (clock, smooth force, link origins, contact descriptors; `Lds` of the backend): the setState
// snapshot, the motor commands and qstate live in the HBM record (written when they change, read where the end of the step needs them),
// the action is read from its row.  Reference call sites as cassie_leg_core.h (Cassie2d::Step/StepPd, Cassie2d.cpp:86-117; mj_step;
// Cassie2dEnv.step, rllab/envs/cassie2d.py:97-225, cassie_stand2d.py:86-137).
#ifndef CASSIE_DUO_CORE_H_
#define CASSIE_DUO_CORE_H_

#include "cassie_leg_core.h"

#ifndef CASSIE_DUO_VIEW_FLAG
#define CASSIE_DUO_VIEW_FLAG 0x20000000   // (DUO_VIEW_EXPERIMENT builds only; no caller sets it)
#endif
#ifndef DUO_JOINT8
#define DUO_JOINT8 1   // groups that leave the six-row path take the eight-row JOINT sweep (r06); 0: the pair sweep inline, as in r05
#endif
#ifndef LEG_NOUNROLL
#define LEG_NOUNROLL _Pragma("clang loop unroll(disable)")
#endif

namespace cassie {
namespace leg {

template <class B> struct Duo : Core<B> {
  typedef Core<B> C;
  typedef typename B::D D;
  typedef typename B::I I;
  typedef typename B::M M;
  typedef typename C::Lane Lane;
  typedef typename C::Sub Sub;
  typedef typename C::SubOut SubOut;
  typedef typename C::Out Out;
  typedef typename C::Io Io;

  static constexpr int NR = 6;                   // row slots of a leg in the joint sweep: 2 connect + 2 contact pairs
  static constexpr int NA = NR * (NR + 1) / 2;
  static constexpr int NP = 2;

  // rows of ONE LEG of the lane's environment
  struct LegRows {
    D r[NR], f[NR], ut[NR][3], Al[NA], Adiag[NR], Ainv[NR], Ant[NP];
    M pair[NP];   // contact pair P exists (its two rows are K_CN / K_CT)
  };

  // Lane roles in the joint sweep: an EVEN lane (left-leg lane of the pair set-up) holds environment e of group A, an ODD lane environment
  // e of group B.  sa / sb = the two groups' values on THIS lane (its own leg); X = 0: the environment's LEFT leg, 1: its RIGHT leg.
  //   even lane, left leg  = own sa           even lane, right leg = partner's sa
  //   odd lane,  left leg  = partner's sb     odd lane,  right leg = own sb
  template <int X> static LEG_FN D pick(M even, D sa, D sb) {
    if constexpr (X == 0) return B::sel(even, sa, B::swap(sb));
    else return B::sel(even, B::swap(sa), sb);
  }
  template <int X> static LEG_FN M pickm(M even, M ma, M mb) {
    if constexpr (X == 0) return (even & ma) | ((!even) & B::swapm(mb));
    else return (even & B::swapm(ma)) | ((!even) & mb);
  }
  template <int X> static LEG_FN void gather(M even, const Sub& sa, const Sub& sb, LegRows& g) {
    lfor<0, NR>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      g.r[Ii] = pick<X>(even, sa.r[Ii], sb.r[Ii]);
      g.f[Ii] = pick<X>(even, sa.f[Ii], sb.f[Ii]);
      g.Adiag[Ii] = pick<X>(even, sa.Adiag[Ii], sb.Adiag[Ii]);
      g.Ainv[Ii] = pick<X>(even, sa.Ainv[Ii], sb.Ainv[Ii]);
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; g.ut[Ii][Bc] = pick<X>(even, sa.ut[Ii][Bc], sb.ut[Ii][Bc]); });
      lfor<Ii, NR>([&](auto jj) {
        constexpr int Jj = decltype(jj)::value;
        g.Al[symidx(NR, Ii, Jj)] = pick<X>(even, sa.Al[symidx(CAP, Ii, Jj)], sb.Al[symidx(CAP, Ii, Jj)]);
      });
    });
    lfor<0, NP>([&](auto pp) {
      constexpr int P = decltype(pp)::value;
      g.Ant[P] = pick<X>(even, sa.Ant[P], sb.Ant[P]);
      g.pair[P] = pickm<X>(even, sa.go & (sa.ncon > I(P)), sb.go & (sb.ncon > I(P)));
    });
  }

  // ---- the PGS sweeps of cassie_leg_core.h (`sub_sweeps`, six-row instantiation) for a lane that holds BOTH legs of its environment.
  // Same step functions, same order (connect L, connect R, contacts L, contacts R), every lane the owner of every step; the cost
  // accumulators of the two legs are separate (the pair sweep keeps one per lane and adds the partner's at the end).
  static LEG_FN void joint_sweeps(LegRows& L, LegRows& R, D& a0, D& a1, D& a2, M go, I& niter_out) {
    const D mu = CP_CONTACT_MU;
    const D scale = 1.0 / (CP_MEANINERTIA * LNV);
    M sweeping = go;
    bool anyPair[2][NP];
    lfor<0, NP>([&](auto pp) { constexpr int P = decltype(pp)::value; anyPair[0][P] = B::any(go & L.pair[P]); anyPair[1][P] = B::any(go & R.pair[P]); });
    D accL = 0.0, accR = 0.0;
    D rdenL[NP], rdenR[NP];
    // FULL (compile time): every lane of the wavefront is sweeping (64 robots, none converged yet: the common case) -- the "still
    // sweeping" selects are identities and are left out; a lane without a pair runs the pair's step on its all-zero rows with the update
    // masked off (the step adds exact zeros).  Same arithmetic.
    auto eq_step = [&](LegRows& g, D& acc, auto ss, auto full_) {
      LEG_FP_CONTRACT_OFF
      constexpr int S = decltype(ss)::value;
      constexpr bool FULL = decltype(full_)::value != 0;
      const D res = B::fma(g.ut[S][2], a2, B::fma(g.ut[S][1], a1, B::fma(g.ut[S][0], a0, g.r[S])));
      D d = -(res * g.Ainv[S]);
      D chg = d * B::fma(g.Adiag[S], d, res);   // (Adiag of a connect row holds A_ii / 2: sub_setup)
      if constexpr (!FULL) {
        const M mine = sweeping;   // slots 0, 1 are K_EQ in every environment that goes
        d = B::sel(mine, d, D(0.0)); chg = B::sel(mine, chg, D(0.0));
      }
      a0 = B::fma(g.ut[S][0], d, a0); a1 = B::fma(g.ut[S][1], d, a1); a2 = B::fma(g.ut[S][2], d, a2);
      acc = acc + chg;
      g.f[S] = g.f[S] + d;
      lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; g.r[Ii] = B::fma(g.Al[symidx(NR, Ii, S)], d, g.r[Ii]); });
    };
    auto pair_step = [&](LegRows& g, D& acc, const D (&rden)[NP], auto pp, auto full_) {
      LEG_FP_CONTRACT_OFF
      constexpr int P = decltype(pp)::value;
      constexpr bool FULL = decltype(full_)::value != 0;
      constexpr int N = 2 + 2 * P, T = 3 + 2 * P;
      const D rn = B::fma(g.ut[N][2], a2, B::fma(g.ut[N][1], a1, B::fma(g.ut[N][0], a0, g.r[N])));
      const D rt = B::fma(g.ut[T][2], a2, B::fma(g.ut[T][1], a1, B::fma(g.ut[T][0], a0, g.r[T])));
      const D on = g.f[N], ot = g.f[T];
      const D Ann = g.Adiag[N], Att = g.Adiag[T], Ant_ = g.Ant[P];
      const D fn_n = B::fmax(B::fma(-rn, g.Ainv[N], on), D(0.0));
      D x = -B::fma(ot, rt, on * rn) * rden[P];
      x = B::fmax(x, D(-1.0));
      const M use_n = on < LMINVAL;
      D fn = B::sel(use_n, fn_n, B::fma(x, on, on));
      D ft = B::sel(use_n, D(0.0), B::fma(x, ot, ot));
      const D bc = B::fma(Ant_, fn - on, B::fma(-Att, ot, rt));
      const D x0 = -bc * g.Ainv[T];
      const D v1 = x0 * (1.0 / mu);
      const D val = B::fma(v1, v1, -(fn * fn));
      const M on_cone = (val >= 1e-10) & (val * Att * (mu * mu) >= 2e-10 * (v1 * v1));
      const D ftc = B::sel(on_cone, B::copysign(mu * fn, x0), x0);
      ft = B::sel(fn >= LMINVAL, ftc, ft);
      D dn = fn - on, dt = ft - ot;
      D chg = B::fma(dt, B::fma(0.5 * Att, dt, B::fma(Ant_, dn, rt)), dn * B::fma(0.5 * Ann, dn, rn));
      M keep = chg <= 1e-10;
      if constexpr (FULL) keep = keep & g.pair[P];
      else keep = keep & (sweeping & g.pair[P]);
      dn = B::sel(keep, dn, D(0.0)); dt = B::sel(keep, dt, D(0.0)); chg = B::sel(keep, chg, D(0.0));
      a0 = B::fma(g.ut[T][0], dt, B::fma(g.ut[N][0], dn, a0)); a1 = B::fma(g.ut[T][1], dt, B::fma(g.ut[N][1], dn, a1)); a2 = B::fma(g.ut[T][2], dt, B::fma(g.ut[N][2], dn, a2));
      acc = acc + chg;
      g.f[N] = g.f[N] + dn; g.f[T] = g.f[T] + dt;
      lfor<0, NR>([&](auto ii) {
        constexpr int Ii = decltype(ii)::value;
        g.r[Ii] = B::fma(g.Al[symidx(NR, Ii, T)], dt, B::fma(g.Al[symidx(NR, Ii, N)], dn, g.r[Ii]));
      });
    };
    auto ray_den = [&](const LegRows& g, D (&rden)[NP], auto pp) {
      LEG_FP_CONTRACT_OFF
      constexpr int P = decltype(pp)::value;
      constexpr int N = 2 + 2 * P, T = 3 + 2 * P;
      const D on = g.f[N], ot = g.f[T];
      const D denom = B::fma(ot, B::fma(g.Adiag[T], ot, g.Ant[P] * on), on * B::fma(g.Ant[P], ot, g.Adiag[N] * on));
      rden[P] = B::sel(denom >= LMINVAL, B::rcp(denom), D(0.0));
    };
    I niter = 0;
    int iter = 0;
    // two loops, not one loop with two bodies (one loop with both bodies: the allocator shuffles the row data between the register files,
    // 228 moves per pass instead of 80): sweeps while every lane is a sweeping robot, then the general sweeps
    if (!B::any(!go) && anyPair[0][0] && anyPair[0][1] && anyPair[1][0] && anyPair[1][1]) {   // (every pair slot is somebody's: no step is run for nobody)
      for (; iter < LEG_ITERS; iter++) {
        if (B::any(!sweeping)) break;
        accL = 0.0; accR = 0.0;
        lfor<0, NP>([&](auto pp) { ray_den(L, rdenL, pp); });
        lfor<0, NP>([&](auto pp) { ray_den(R, rdenR, pp); });
        eq_step(L, accL, LI<0>{}, LI<1>{}); eq_step(L, accL, LI<1>{}, LI<1>{});
        eq_step(R, accR, LI<0>{}, LI<1>{}); eq_step(R, accR, LI<1>{}, LI<1>{});
        lfor<0, NP>([&](auto pp) { pair_step(L, accL, rdenL, pp, LI<1>{}); });
        lfor<0, NP>([&](auto pp) { pair_step(R, accR, rdenR, pp, LI<1>{}); });
        const D improvement = -(accL + accR);
        niter = niter + 1;
        sweeping = sweeping & !(improvement * scale < CP_TOLERANCE);
      }
    }
    for (; iter < LEG_ITERS; iter++) {
      if (!B::any(sweeping)) break;
      accL = 0.0; accR = 0.0;
      lfor<0, NP>([&](auto pp) { ray_den(L, rdenL, pp); });
      lfor<0, NP>([&](auto pp) { ray_den(R, rdenR, pp); });
      eq_step(L, accL, LI<0>{}, LI<0>{}); eq_step(L, accL, LI<1>{}, LI<0>{});
      eq_step(R, accR, LI<0>{}, LI<0>{}); eq_step(R, accR, LI<1>{}, LI<0>{});
      if (anyPair[0][0]) {
        lfor<0, NP>([&](auto pp) { constexpr int P = decltype(pp)::value; if (anyPair[0][P]) pair_step(L, accL, rdenL, pp, LI<0>{}); });
      }
      if (anyPair[1][0]) {
        lfor<0, NP>([&](auto pp) { constexpr int P = decltype(pp)::value; if (anyPair[1][P]) pair_step(R, accR, rdenR, pp, LI<0>{}); });
      }
      const D improvement = -(accL + accR);
      niter = niter + B::toI(sweeping);
      sweeping = sweeping & !(improvement * scale < CP_TOLERANCE);
    }
    niter_out = niter;
  }

  // ------------------------------------------------------------------------------------------------ hand-over (per-wavefront workspace)
  // The set-up of a group needs the whole register file (it is the two-lanes kernel's: 256 + 256 registers), so nothing of group A can
  // wait in registers while group B is set up.  Left to the register allocator the hand-over was ~650 scratch accesses per substep pass,
  // each reload waited for where it is used (r05 PMC of the first build: 41 % of the wavefront's cycles in s_waitcnt, 1.46 ms per
  // 65 536-env step against 1.40 for the two-lanes kernel although it issued a third fewer instructions).  So the phases of a pass --
  // set-up(g), joint sweep, finish(g) -- are made independent: EVERYTHING a group carries from one phase to the next (state, rows,
  // factorisation, forces) goes through a per-wavefront workspace in global memory, [slot][lane] (one coalesced 512-byte access per slot),
  // stored at the end of a phase and loaded in ONE batch at the head of the next, and the set-up / finish code exists once, in a loop over
  // the two groups.  ~600 accesses of 512 bytes per pass and wavefront; measured beside FP64 work in isolation (1024 wavefronts, 135 KB
  // each, the pattern of this kernel): +0..5 % -- the workspace of a wavefront is re-used every ~80 us and lives in L2 / Infinity Cache.
  enum {
    W_ST = 0,                                           // 24: the lane state (q, v, warm start)
    W_ROWS = W_ST + 24, W_NROWS = 4 * NR + 3 * NR + NA + NP + 3,   // 68: the six-row subset the joint sweep takes (the forces come back in their slots)
    W_NROWS8 = 4 * CAP + 3 * CAP + CAP * (CAP + 1) / 2 + 3 + 3,     // 98: all eight rows of a leg, for the eight-row joint sweep (r06): the block's size
    W_FC = W_ROWS + W_NROWS8,                           // 36: block factorisation
    W_ANCH = W_FC + 36,                                 // 4: connect anchors
    W_MISC = W_ANCH + 4,                                // go, ncon, sweeps done, nlim
    W_DESC = W_MISC + 4,                                // 10: what the finish of an eight-row group needs of the descriptors that exist once per wavefront in LDS:
                                                        //     4 joint limits (sign x (dof + 1)), third contact pair (x, z, depth), its terrain normal; [8] experiment marker
    W_GROUP = W_DESC + 10,                              // slots per group (even: every block starts on an even slot, see put_block)
    W_N = 2 * W_GROUP
  };
  static_assert(W_NROWS8 >= W_NROWS && W_NROWS8 % 2 == 0, "one rows block for both formats");
  static_assert(W_ROWS % 2 == 0 && W_FC % 2 == 0 && W_ANCH == W_FC + 36 && W_MISC % 2 == 0 && W_DESC % 2 == 0 && W_GROUP % 2 == 0, "blocks start on even slots; the anchors follow the factorisation");
  typedef typename B::W W;
  template <class F> static LEG_FN void rows_each(Sub& s, int base, F&& f) {
    int k = base + W_ROWS;
    lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; f(k++, s.f[Ii]); });   // forces first: slots base + W_ROWS + i
    lfor<0, NR>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      f(k++, s.r[Ii]); f(k++, s.Adiag[Ii]); f(k++, s.Ainv[Ii]);
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; f(k++, s.ut[Ii][Bc]); });
      lfor<Ii, NR>([&](auto jj) { constexpr int Jj = decltype(jj)::value; f(k++, s.Al[symidx(CAP, Ii, Jj)]); });
    });
    lfor<0, NP>([&](auto pp) { constexpr int P = decltype(pp)::value; f(k++, s.Ant[P]); });
    f(k++, s.a0); f(k++, s.a1); f(k++, s.a2);
  }
  template <class F> static LEG_FN void keep_each(Sub& s, int base, F&& f) {
    int k = base + W_FC;
    lfor<0, 15>([&](auto ii) { constexpr int Ii = decltype(ii)::value; f(k++, s.fc.Li[Ii]); });
    lfor<0, 5>([&](auto ii) { constexpr int Ii = decltype(ii)::value; lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; f(k++, s.fc.Y[Ii][Bc]); }); });
    lfor<0, 6>([&](auto ii) { constexpr int Ii = decltype(ii)::value; f(k++, s.fc.G[Ii]); });
    f(k++, s.p1x); f(k++, s.p1z); f(k++, s.p2x); f(k++, s.p2z);
  }
  template <class F> static LEG_FN void lane_each(Lane& st, int base, F&& f) {
    int k = base + W_ST;
    lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; f(k++, st.qb[Bc]); f(k++, st.vb[Bc]); f(k++, st.wb[Bc]); });
    lfor<0, 5>([&](auto dd) { constexpr int Dd = decltype(dd)::value; f(k++, st.ql[Dd]); f(k++, st.vl[Dd]); f(k++, st.wl[Dd]); });
  }
  // A block of N consecutive slots (first slot even) moves as N / 2 sixteen-byte accesses per lane (B::wld2 / wst2: the slots 2p, 2p + 1 of
  // a lane are adjacent in the workspace) + one eight-byte access if N is odd.  Why pairs: a wavefront has at most 63 memory instructions
  // in flight (vmcnt), the workspace answers from the Infinity Cache in ~1 us, and at 512 bytes per instruction that is ~10 bytes per
  // cycle and wavefront -- what the eight-byte form of this hand-over measured (r05 phase clocks: ~60 cycles per access).
  template <int N> static LEG_FN void put_block(W ws, int first, D (&t)[N]) {
    lfor<0, N / 2>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::wst2(ws, first + 2 * Ii, t[2 * Ii], t[2 * Ii + 1]); });
    if constexpr (N & 1) B::wst(ws, first + N - 1, t[N - 1]);
  }
  template <int N> static LEG_FN void get_block(W ws, int first, D (&t)[N]) {
    lfor<0, N / 2>([&](auto ii) { constexpr int Ii = decltype(ii)::value; B::wld2(ws, first + 2 * Ii, t[2 * Ii], t[2 * Ii + 1]); });
    if constexpr (N & 1) t[N - 1] = B::wld(ws, first + N - 1);
  }
  static LEG_FN void put_rows(W ws, int base, Sub& s) { D t[W_NROWS]; rows_each(s, 0, [&](int k, D& v) { t[k - W_ROWS] = v; }); put_block(ws, base + W_ROWS, t); }
  static LEG_FN void get_rows(W ws, int base, Sub& s) { D t[W_NROWS]; get_block(ws, base + W_ROWS, t); rows_each(s, 0, [&](int k, D& v) { v = t[k - W_ROWS]; }); }
  static LEG_FN void put_keep(W ws, int base, Sub& s) { D t[40]; keep_each(s, 0, [&](int k, D& v) { t[k - W_FC] = v; }); put_block(ws, base + W_FC, t); }
  static LEG_FN void get_keep(W ws, int base, Sub& s) { D t[40]; get_block(ws, base + W_FC, t); keep_each(s, 0, [&](int k, D& v) { v = t[k - W_FC]; }); }
  static LEG_FN void put_lane(W ws, int base, Lane& st) { D t[24]; lane_each(st, 0, [&](int k, D& v) { t[k - W_ST] = v; }); put_block(ws, base + W_ST, t); }
  static LEG_FN void get_lane(W ws, int base, Lane& st) { D t[24]; get_block(ws, base + W_ST, t); lane_each(st, 0, [&](int k, D& v) { v = t[k - W_ST]; }); }
  static LEG_FN void put_forces(W ws, int base, Sub& s) { D t[NR]; lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; t[Ii] = s.f[Ii]; }); put_block(ws, base + W_ROWS, t); }
  static LEG_FN void get_forces(W ws, int base, Sub& s) { D t[NR]; get_block(ws, base + W_ROWS, t); lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; s.f[Ii] = t[Ii]; }); }
  // masks and counts travel as doubles
  static LEG_FN void put_misc(W ws, int base, M go, I ncon) { B::wst2(ws, base + W_MISC, B::sel(go, D(1.0), D(0.0)), B::toD(ncon)); }
  static LEG_FN void get_misc(W ws, int base, M& go, I& ncon) { D a, b; B::wld2(ws, base + W_MISC, a, b); go = a > D(0.5); ncon = B::toint(b); }

  // rows of a group that does not run in this pass (no environment of it is live): defined values for the lanes the joint sweep masks off
  static LEG_FN void idle_rows(Sub& s) {
    const M none = (B::leg() == I(0)) & !(B::leg() == I(0));
    s.go = none; s.ncon = 0; s.nlim = 0; s.niter = 0;
    s.a0 = 0.0; s.a1 = 0.0; s.a2 = 0.0;
    lfor<0, NR>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      s.r[Ii] = 0.0; s.f[Ii] = 0.0; s.Adiag[Ii] = Ii < 2 ? 0.5 : 1.0; s.Ainv[Ii] = 1.0;
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; s.ut[Ii][Bc] = 0.0; });
      lfor<Ii, NR>([&](auto jj) { constexpr int Jj = decltype(jj)::value; s.Al[symidx(CAP, Ii, Jj)] = 0.0; });
    });
    lfor<0, NP>([&](auto pp) { constexpr int P = decltype(pp)::value; s.Ant[P] = 0.0; });
  }

  // One joint solve for the groups of `join` (wave-uniform flags): rows in, forces and iteration counts back in the groups' Sub.
  static LEG_FN void joint_solve(Sub (&S)[2], const bool (&join)[2]) {
    const M even = B::leg() == I(0);
    const M none = even & !even;
    // environments of a group that does not take part never sweep (their lanes hold whatever that group's set-up left)
    const M goA = join[0] ? S[0].go : none, goB = join[1] ? S[1].go : none;
    LegRows L, R;
    gather<0>(even, S[0], S[1], L);
    gather<1>(even, S[0], S[1], R);
    D a0 = B::sel(even, S[0].a0, S[1].a0), a1 = B::sel(even, S[0].a1, S[1].a1), a2 = B::sel(even, S[0].a2, S[1].a2);   // a~ is the same on both lanes of a pair
    const M go = (even & goA) | ((!even) & goB);
    I niter;
    B::fence();
    joint_sweeps(L, R, a0, a1, a2, go, niter);
    B::fence();
    // forces back to the leg lanes: group A's left legs sit on the even lanes (own L), its right legs on the odd lanes (partner's R);
    // group B's left legs on the even lanes (partner's L), its right legs on the odd lanes (own R)
    const I nsw = B::swapi(niter);
    if (join[0]) {
      lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; S[0].f[Ii] = B::sel(even, L.f[Ii], B::swap(R.f[Ii])); });
      S[0].niter = B::seli(even, niter, nsw);
    }
    if (join[1]) {
      lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; S[1].f[Ii] = B::sel(even, B::swap(L.f[Ii]), R.f[Ii]); });
      S[1].niter = B::seli(even, nsw, niter);
    }
  }

#ifdef DUO_VIEW_EXPERIMENT
  // ---- r05 EXPERIMENT, kept as source for the build guard (tests/test_gpu_build_guard.py, DESIGN.md section 5 K1d "dead-code dependence"): the joint
  // sweep's rows transposed by the ADDRESSES of the hand-over instead of lane exchanges.  Joint lane 2e + k (environment e of group k) reads its LEFT
  // leg's rows from column 2e of group k's block and its RIGHT leg's from column 2e + 1 (B::wview), and writes the forces back the same way.  Slower
  // (every load touches 2 KB of workspace and uses half of it), and in r05 the GPU build that merely CONTAINED this branch -- never taken -- gave wrong
  // results.  -DDUO_VIEW_EXPERIMENT compiles the branch in behind a flag no caller sets (CASSIE_DUO_VIEW_FLAG).
  template <int X> static LEG_FN void view_rows(W wv, LegRows& g, D& a0, D& a1, D& a2, M& go_) {
    D t[W_NROWS];
    get_block(wv, W_ROWS, t);
    int k = 0;
    lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; g.f[Ii] = t[k++]; });
    lfor<0, NR>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      g.r[Ii] = t[k++]; g.Adiag[Ii] = t[k++]; g.Ainv[Ii] = t[k++];
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; g.ut[Ii][Bc] = t[k++]; });
      lfor<Ii, NR>([&](auto jj) { constexpr int Jj = decltype(jj)::value; g.Al[symidx(NR, Ii, Jj)] = t[k++]; });
    });
    lfor<0, NP>([&](auto pp) { constexpr int P = decltype(pp)::value; g.Ant[P] = t[k++]; });
    if constexpr (X == 0) { a0 = t[k]; a1 = t[k + 1]; a2 = t[k + 2]; }
    M go; I ncon;
    get_misc(wv, 0, go, ncon);
    lfor<0, NP>([&](auto pp) { constexpr int P = decltype(pp)::value; g.pair[P] = go & (ncon > I(P)); });
    if constexpr (X == 0) go_ = go;
  }
  static LEG_FN void joint_solve_view(W ws, const bool (&join)[2]) {
    const M even = B::leg() == I(0);
    const M none = even & !even;
    const M mine = (even & (join[0] ? !none : none)) | ((!even) & (join[1] ? !none : none));   // this lane's group takes part
    const W wl = B::wview(ws, W_GROUP, 0), wr = B::wview(ws, W_GROUP, 1);
    LegRows L, R;
    D a0, a1, a2, d0, d1, d2;
    M goL, goR;
    view_rows<0>(wl, L, a0, a1, a2, goL);
    view_rows<1>(wr, R, d0, d1, d2, goR);
    const M go = mine & goL;
    I niter;
    B::fence();
    joint_sweeps(L, R, a0, a1, a2, go, niter);
    B::fence();
    D tl[NR], tr[NR];
    lfor<0, NR>([&](auto ii) { constexpr int Ii = decltype(ii)::value; tl[Ii] = L.f[Ii]; tr[Ii] = R.f[Ii]; });
    if (B::any(mine)) {
      B::wput_if(wl, W_ROWS, tl, mine); B::wput_if(wr, W_ROWS, tr, mine);
      B::wst_if(wl, W_MISC + 2, B::toD(niter), mine); B::wst_if(wr, W_MISC + 2, B::toD(niter), mine);
      B::wst_if(wl, W_DESC + 8, D(1.0), mine);   // the group's marker slot: "this path ran" (the guard test reads it back; nothing else writes it)
    }
  }
#endif

  // ------------------------------------------------------------------------------------------------ eight-row joint sweep (r06)
  // Until r05 a group in which some environment had a joint limit active or a third contact pair on a leg ran the PAIR sweep of cassie_leg_core.h
  // for that substep (both lanes of an environment execute every step, one keeps it) -- under random torques 95-99.6 % of the groups
  // (profiles: tools/small_stats.py).  Here such a group hands ALL EIGHT row slots of its legs to the workspace and the wavefront runs one JOINT sweep
  // with a lane per environment over eight rows per leg: the step functions of `sub_sweeps` (connect, joint limit, contact pair) for one lane, in the
  // pair sweep's order -- connect L, connect R, limits L, limits R, pairs L, pairs R -- with the row kinds as per-lane masks (limit j at slot 7 - j while
  // nlim > j, pair P at slots 2 + 2P / 3 + 2P while ncon > P: what sub_setup assigns).  ~190 doubles of row data per lane: the rows are transposed
  // STRAIGHT from the workspace in chunks (never both groups' Sub in registers), and the allocator keeps part of them in the accumulator file.
  // A group on its feet that shares the wavefront with such a group is swept with it (its slots 6, 7 are empty rows); results are bit-identical to
  // the pair sweep either way (same step functions, contraction off, same order, the two legs' cost accumulators added as the pair sweep adds them).
  struct LegRows8 {
    D r[CAP], f[CAP], ut[CAP][3], Al[CAP * (CAP + 1) / 2], Adiag[CAP], Ainv[CAP], Ant[3];
    M pair[3], lim[4];
  };
  // order of the 98 slots of the eight-row block: f 0..7 | r 8..15 | Adiag 16..23 | Ainv 24..31 | ut 32..55 | Al 56..91 (packed as in Sub) | Ant 92..94 | a~ 95..97
  template <int K> static LEG_FN D& sfld8(Sub& s) {
    if constexpr (K < 8) return s.f[K];
    else if constexpr (K < 16) return s.r[K - 8];
    else if constexpr (K < 24) return s.Adiag[K - 16];
    else if constexpr (K < 32) return s.Ainv[K - 24];
    else if constexpr (K < 56) return s.ut[(K - 32) / 3][(K - 32) % 3];
    else if constexpr (K < 92) return s.Al[K - 56];
    else if constexpr (K < 95) return s.Ant[K - 92];
    else if constexpr (K == 95) return s.a0;
    else if constexpr (K == 96) return s.a1;
    else return s.a2;
  }
  template <int K> static LEG_FN D& lfld8(LegRows8& g, D (&a)[3]) {
    if constexpr (K < 8) return g.f[K];
    else if constexpr (K < 16) return g.r[K - 8];
    else if constexpr (K < 24) return g.Adiag[K - 16];
    else if constexpr (K < 32) return g.Ainv[K - 24];
    else if constexpr (K < 56) return g.ut[(K - 32) / 3][(K - 32) % 3];
    else if constexpr (K < 92) return g.Al[K - 56];
    else if constexpr (K < 95) return g.Ant[K - 92];
    else return a[K - 95];
  }
  static LEG_FN void put_rows8(W ws, int base, Sub& s) {
    D t[W_NROWS8];
    lfor<0, W_NROWS8>([&](auto kk) { constexpr int K = decltype(kk)::value; t[K] = sfld8<K>(s); });
    put_block(ws, base + W_ROWS, t);
  }
  // a group on its feet (six-row format in the workspace) re-written in the eight-row format: its slots 6, 7 are empty rows
  static LEG_FN void widen_rows(W ws, int base) {
    Sub s;
    get_rows(ws, base, s);
    lfor<NR, CAP>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      s.r[Ii] = 0.0; s.f[Ii] = 0.0; s.Adiag[Ii] = 1.0; s.Ainv[Ii] = 1.0;
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; s.ut[Ii][Bc] = 0.0; });
      lfor<0, CAP>([&](auto jj) { constexpr int Jj = decltype(jj)::value; s.Al[symidx(CAP, Ii, Jj)] = 0.0; });
    });
    s.Ant[2] = 0.0;
    B::fence();
    put_rows8(ws, base, s);
    B::wst(ws, base + W_MISC + 3, D(0.0));   // nlim
  }
  // a group that does not run in this pass: empty rows, nobody goes
  static LEG_FN void put_idle8(W ws, int base) {
    Sub s;
    idle_rows(s);
    lfor<NR, CAP>([&](auto ii) {
      constexpr int Ii = decltype(ii)::value;
      s.r[Ii] = 0.0; s.f[Ii] = 0.0; s.Adiag[Ii] = 1.0; s.Ainv[Ii] = 1.0;
      lfor<0, 3>([&](auto bb) { constexpr int Bc = decltype(bb)::value; s.ut[Ii][Bc] = 0.0; });
    });
    lfor<0, CAP>([&](auto ii) { constexpr int Ii = decltype(ii)::value; lfor<Ii, CAP>([&](auto jj) { constexpr int Jj = decltype(jj)::value; s.Al[symidx(CAP, Ii, Jj)] = 0.0; }); });
    s.Ant[2] = 0.0;
    put_rows8(ws, base, s);
    B::wst2(ws, base + W_MISC, D(0.0), D(0.0));
    B::wst(ws, base + W_MISC + 3, D(0.0));
  }
  // the descriptors that exist once per wavefront in LDS (joint limits, third contact pair): parked for the group's finish, restored before it
  template <bool HF> static LEG_FN void put_desc(typename B::Lds& lds, W ws, int base) {
    D t[8];
    lfor<0, 4>([&](auto jj) {
      constexpr int Jj = decltype(jj)::value;
      D pos, sgn, invw; I lj;
      lds.ld_lim(Jj, pos, sgn, invw, lj);
      t[Jj] = sgn * B::toD(lj + I(1));
    });
    D px, pz, dist, invw; I depth;
    lds.ld_pair(2, px, pz, dist, invw, depth);
    t[4] = px; t[5] = pz; t[6] = B::toD(depth); t[7] = 0.0;
    if constexpr (HF) t[7] = lds.ld_nrm(2);
    put_block(ws, base + W_DESC, t);
  }
  template <bool HF> static LEG_FN void get_desc(typename B::Lds& lds, W ws, int base) {
    D t[8];
    get_block(ws, base + W_DESC, t);
    const M all = (B::leg() == I(0)) | !(B::leg() == I(0));
    lfor<0, 4>([&](auto jj) {
      constexpr int Jj = decltype(jj)::value;
      const D c = t[Jj];
      lds.st_lim(Jj, D(0.0), B::sel(c < 0.0, D(-1.0), D(1.0)), D(0.0), B::toint(B::fabs(c)) - I(1), all);
    });
    lds.st_pair(2, t[4], t[5], D(0.0), D(0.0), B::toint(t[6]), all);
    if constexpr (HF) lds.st_nrm(2, t[7], all);
  }

  static LEG_FN void joint_sweeps8(LegRows8& L, LegRows8& R, D& a0, D& a1, D& a2, M go, I& niter_out) {
    const D mu = CP_CONTACT_MU;
    const D scale = 1.0 / (CP_MEANINERTIA * LNV);
    M sweeping = go;
    bool anyLim[2][4], anyPair[2][3];
    lfor<0, 4>([&](auto jj) { constexpr int Jj = decltype(jj)::value; anyLim[0][Jj] = B::any(go & L.lim[Jj]); anyLim[1][Jj] = B::any(go & R.lim[Jj]); });
    lfor<0, 3>([&](auto pp) { constexpr int P = decltype(pp)::value; anyPair[0][P] = B::any(go & L.pair[P]); anyPair[1][P] = B::any(go & R.pair[P]); });
    D accL = 0.0, accR = 0.0;
    D rdenL[3], rdenR[3];
    auto eq_step = [&](LegRows8& g, D& acc, auto ss) {
      LEG_FP_CONTRACT_OFF
      constexpr int S = decltype(ss)::value;
      const D res = B::fma(g.ut[S][2], a2, B::fma(g.ut[S][1], a1, B::fma(g.ut[S][0], a0, g.r[S])));
      D d = -(res * g.Ainv[S]);
      D chg = d * B::fma(g.Adiag[S], d, res);   // (Adiag of a connect row holds A_ii / 2: sub_setup)
      d = B::sel(sweeping, d, D(0.0)); chg = B::sel(sweeping, chg, D(0.0));
      a0 = B::fma(g.ut[S][0], d, a0); a1 = B::fma(g.ut[S][1], d, a1); a2 = B::fma(g.ut[S][2], d, a2);
      acc = acc + chg;
      g.f[S] = g.f[S] + d;
      lfor<0, CAP>([&](auto ii) { constexpr int Ii = decltype(ii)::value; g.r[Ii] = B::fma(g.Al[symidx(CAP, Ii, S)], d, g.r[Ii]); });
    };
    auto lim_step = [&](LegRows8& g, D& acc, auto jj) {
      LEG_FP_CONTRACT_OFF
      constexpr int Jj = decltype(jj)::value;
      constexpr int S = 7 - Jj;
      const M mine = sweeping & g.lim[Jj];
      const D res = B::fma(g.ut[S][2], a2, B::fma(g.ut[S][1], a1, B::fma(g.ut[S][0], a0, g.r[S])));
      const D cand = B::fmax(B::fma(-res, g.Ainv[S], g.f[S]), D(0.0));
      D d = cand - g.f[S];
      D chg = d * B::fma(0.5 * g.Adiag[S], d, res);
      const M keep = mine & (chg <= 1e-10);
      d = B::sel(keep, d, D(0.0)); chg = B::sel(keep, chg, D(0.0));
      a0 = B::fma(g.ut[S][0], d, a0); a1 = B::fma(g.ut[S][1], d, a1); a2 = B::fma(g.ut[S][2], d, a2);
      acc = acc + chg;
      g.f[S] = g.f[S] + d;
      lfor<0, CAP>([&](auto ii) { constexpr int Ii = decltype(ii)::value; g.r[Ii] = B::fma(g.Al[symidx(CAP, Ii, S)], d, g.r[Ii]); });
    };
    auto pair_step = [&](LegRows8& g, D& acc, const D (&rden)[3], auto pp) {
      LEG_FP_CONTRACT_OFF
      constexpr int P = decltype(pp)::value;
      constexpr int N = 2 + 2 * P, T = 3 + 2 * P;
      const M mine = sweeping & g.pair[P];
      const D rn = B::fma(g.ut[N][2], a2, B::fma(g.ut[N][1], a1, B::fma(g.ut[N][0], a0, g.r[N])));
      const D rt = B::fma(g.ut[T][2], a2, B::fma(g.ut[T][1], a1, B::fma(g.ut[T][0], a0, g.r[T])));
      const D on = g.f[N], ot = g.f[T];
      const D Ann = g.Adiag[N], Att = g.Adiag[T], Ant_ = g.Ant[P];
      const D fn_n = B::fmax(B::fma(-rn, g.Ainv[N], on), D(0.0));
      D x = -B::fma(ot, rt, on * rn) * rden[P];
      x = B::fmax(x, D(-1.0));
      const M use_n = on < LMINVAL;
      D fn = B::sel(use_n, fn_n, B::fma(x, on, on));
      D ft = B::sel(use_n, D(0.0), B::fma(x, ot, ot));
      const D bc = B::fma(Ant_, fn - on, B::fma(-Att, ot, rt));
      const D x0 = -bc * g.Ainv[T];
      const D v1 = x0 * (1.0 / mu);
      const D val = B::fma(v1, v1, -(fn * fn));
      const M on_cone = (val >= 1e-10) & (val * Att * (mu * mu) >= 2e-10 * (v1 * v1));
      const D ftc = B::sel(on_cone, B::copysign(mu * fn, x0), x0);
      ft = B::sel(fn >= LMINVAL, ftc, ft);
      D dn = fn - on, dt = ft - ot;
      D chg = B::fma(dt, B::fma(0.5 * Att, dt, B::fma(Ant_, dn, rt)), dn * B::fma(0.5 * Ann, dn, rn));
      const M keep = mine & (chg <= 1e-10);
      dn = B::sel(keep, dn, D(0.0)); dt = B::sel(keep, dt, D(0.0)); chg = B::sel(keep, chg, D(0.0));
      a0 = B::fma(g.ut[T][0], dt, B::fma(g.ut[N][0], dn, a0)); a1 = B::fma(g.ut[T][1], dt, B::fma(g.ut[N][1], dn, a1)); a2 = B::fma(g.ut[T][2], dt, B::fma(g.ut[N][2], dn, a2));
      acc = acc + chg;
      g.f[N] = g.f[N] + dn; g.f[T] = g.f[T] + dt;
      lfor<0, CAP>([&](auto ii) {
        constexpr int Ii = decltype(ii)::value;
        g.r[Ii] = B::fma(g.Al[symidx(CAP, Ii, T)], dt, B::fma(g.Al[symidx(CAP, Ii, N)], dn, g.r[Ii]));
      });
    };
    auto ray_den = [&](const LegRows8& g, D (&rden)[3], auto pp) {
      LEG_FP_CONTRACT_OFF
      constexpr int P = decltype(pp)::value;
      constexpr int N = 2 + 2 * P, T = 3 + 2 * P;
      const D on = g.f[N], ot = g.f[T];
      const D denom = B::fma(ot, B::fma(g.Adiag[T], ot, g.Ant[P] * on), on * B::fma(g.Ant[P], ot, g.Adiag[N] * on));
      rden[P] = B::sel(denom >= LMINVAL, B::rcp(denom), D(0.0));
    };
    I niter = 0;
    for (int iter = 0; iter < LEG_ITERS; iter++) {
      if (!B::any(sweeping)) break;
      accL = 0.0; accR = 0.0;
      lfor<0, 3>([&](auto pp) { ray_den(L, rdenL, pp); });
      lfor<0, 3>([&](auto pp) { ray_den(R, rdenR, pp); });
      eq_step(L, accL, LI<0>{}); eq_step(L, accL, LI<1>{});
      eq_step(R, accR, LI<0>{}); eq_step(R, accR, LI<1>{});
      if (anyLim[0][0]) { lfor<0, 4>([&](auto jj) { constexpr int Jj = decltype(jj)::value; if (anyLim[0][Jj]) lim_step(L, accL, jj); }); }
      if (anyLim[1][0]) { lfor<0, 4>([&](auto jj) { constexpr int Jj = decltype(jj)::value; if (anyLim[1][Jj]) lim_step(R, accR, jj); }); }
      if (anyPair[0][0]) { lfor<0, 3>([&](auto pp) { constexpr int P = decltype(pp)::value; if (anyPair[0][P]) pair_step(L, accL, rdenL, pp); }); }
      if (anyPair[1][0]) { lfor<0, 3>([&](auto pp) { constexpr int P = decltype(pp)::value; if (anyPair[1][P]) pair_step(R, accR, rdenR, pp); }); }
      const D improvement = -(accL + accR);
      niter = niter + B::toI(sweeping);
      sweeping = sweeping & !(improvement * scale < CP_TOLERANCE);
    }
    niter_out = niter;
  }

  // Both groups' eight-row blocks (written by put_rows8 / widen_rows / put_idle8) -> one joint sweep -> the forces back into the blocks' force slots,
  // the sweep counts into W_MISC + 2.  `wr[G]`: the group takes part (its forces are wanted).
  static LEG_FN void joint_solve8(W ws, const bool (&wr)[2]) {
    const M even = B::leg() == I(0);
    LegRows8 L, R;
    D a[3], adummy[3];
    constexpr int CH = 14;   // slots per chunk (seven sixteen-byte accesses per group): what is in flight between two fences
    static_assert(W_NROWS8 % CH == 0, "whole chunks");
    lfor<0, W_NROWS8 / CH>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      D ta[CH], tb[CH];
      lfor<0, CH / 2>([&](auto pp) {
        constexpr int P = decltype(pp)::value;
        B::wld2(ws, W_ROWS + CH * C + 2 * P, ta[2 * P], ta[2 * P + 1]);
        B::wld2(ws, W_GROUP + W_ROWS + CH * C + 2 * P, tb[2 * P], tb[2 * P + 1]);
      });
      lfor<0, CH>([&](auto qq) {
        constexpr int Q = decltype(qq)::value;
        constexpr int K = CH * C + Q;
        if constexpr (K < 95) {
          lfld8<K>(L, a) = pick<0>(even, ta[Q], tb[Q]);
          lfld8<K>(R, adummy) = pick<1>(even, ta[Q], tb[Q]);
        } else {
          a[K - 95] = B::sel(even, ta[Q], tb[Q]);   // a~ is the same on both lanes of a pair
        }
      });
      B::fence();
    });
    M goA, goB; I nconA, nconB;
    get_misc(ws, 0, goA, nconA); get_misc(ws, W_GROUP, goB, nconB);
    const I nlimA = B::toint(B::wld(ws, W_MISC + 3)), nlimB = B::toint(B::wld(ws, W_GROUP + W_MISC + 3));
    lfor<0, 3>([&](auto pp) {
      constexpr int P = decltype(pp)::value;
      L.pair[P] = pickm<0>(even, goA & (nconA > I(P)), goB & (nconB > I(P)));
      R.pair[P] = pickm<1>(even, goA & (nconA > I(P)), goB & (nconB > I(P)));
    });
    lfor<0, 4>([&](auto jj) {
      constexpr int Jj = decltype(jj)::value;
      L.lim[Jj] = pickm<0>(even, goA & (nlimA > I(Jj)), goB & (nlimB > I(Jj)));
      R.lim[Jj] = pickm<1>(even, goA & (nlimA > I(Jj)), goB & (nlimB > I(Jj)));
    });
    const M go = (even & goA) | ((!even) & goB);
    I niter;
    B::fence();
    joint_sweeps8(L, R, a[0], a[1], a[2], go, niter);
    B::fence();
    const I nsw = B::swapi(niter);
    if (wr[0]) {
      D t[CAP];
      lfor<0, CAP>([&](auto ii) { constexpr int Ii = decltype(ii)::value; t[Ii] = B::sel(even, L.f[Ii], B::swap(R.f[Ii])); });
      put_block(ws, W_ROWS, t);
      B::wst(ws, W_MISC + 2, B::toD(B::seli(even, niter, nsw)));
    }
    if (wr[1]) {
      D t[CAP];
      lfor<0, CAP>([&](auto ii) { constexpr int Ii = decltype(ii)::value; t[Ii] = B::sel(even, B::swap(L.f[Ii]), R.f[Ii]); });
      put_block(ws, W_GROUP + W_ROWS, t);
      B::wst(ws, W_GROUP + W_MISC + 2, B::toD(B::seli(even, nsw, niter)));
    }
  }

  // ------------------------------------------------------------------------------------------------ fused Env.step, two groups
  // io_of(g): the group's per-lane pointers (record, action row, observation row ...), g wave-uniform at run time; valid / o: per group.
  // The backend's Lds is switched to a group with lds.select(g, io) before any of that group's code runs; lds.snapshot(bool): stores to the
  // setState slots (C_KQ / C_KV) reach the record (the last substep of the step, and the end-of-step section), or are dropped (every
  // earlier substep: only the LAST setState of a step is ever read -- by this step's observation, by the reset pass, by the next launch's
  // controllers; an environment that leaves this tier is finished by a lower tier, which does its own setState on every substep it
  // carries out).
  // HF: the height-field instantiation (terrain collision stage in the set-up, contact frames along the local normal; the joint sweep itself does
  // not know: rows are rows).
  template <int MODE, bool HF = false, class IoOf>
  static LEG_FN void env_step2(const EnvCfg& cfg, typename B::Lds& lds, W ws, IoOf&& io_of, const M (&valid)[2], Out (&o)[2], const Terrain* hf = nullptr) {
    const I leg = B::leg();
    const I lo = leg * 5 + 3, ao = leg * 3;
    const M left = leg == 0;
    const M none = left & !left;
    M live[2];
    lfor<0, 2>([&](auto gg) {
      constexpr int G = decltype(gg)::value;
      const Io io = io_of(G);
      lds.select(G, io);
      Lane st;
      lfor<0, 3>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        st.qb[Bc] = B::pld(io.rec, I(ES_Q + Bc)); st.vb[Bc] = B::pld(io.rec, I(ES_V + Bc)); st.wb[Bc] = B::pld(io.rec, I(ES_WS + Bc));
      });
      lfor<0, 5>([&](auto dd) {
        constexpr int Dd = decltype(dd)::value;
        st.ql[Dd] = B::pld(io.rec, lo + (ES_Q + Dd)); st.vl[Dd] = B::pld(io.rec, lo + (ES_V + Dd)); st.wl[Dd] = B::pld(io.rec, lo + (ES_WS + Dd));
      });
      put_lane(ws, G * W_GROUP, st);
      const M all = valid[G] | !valid[G];
      lds.cst(C::C_TIME, B::pld(io.rec, I(ES_TIME)), all);
      {
        D a2own = 0.0;
        if (io.has_act && cfg.env_kind != 0) {
          lfor<0, 3>([&](auto aa) { constexpr int A_ = decltype(aa)::value; const D a = B::pld(io.act, ao + A_); a2own += a * a; });
          if (cfg.adim == 7) { const D a = B::pld(io.act, I(6)); a2own += B::sel(left, a * a, D(0.0)); }
        }
        lds.cst(C::C_A2, a2own, all);
      }
      live[G] = valid[G];
      o[G].set_state = none;
      o[G].pend = 0; o[G].niter = 0;
      o[G].do_reset = none; o[G].bad = none;
    });
    B::fence();
    lds.mark(16);   // (16..21: pieces of the glue, bucket 0 unless the build splits it) 16 = records in, state parked
    bool reset_pass = false;
    int sub = 0;
    while (true) {
      bool join[2] = {false, false}, ran[2] = {false, false};
      bool join8[2] = {false, false};   // the group's rows are in the workspace in the eight-row format (r06: DUO_JOINT8)
      M ovf[2] = {none, none};
      I nit[2] = {I(0), I(0)};
      lds.snapshot(reset_pass || sub == cfg.n_sub - 1);
      const M lv0 = reset_pass ? o[0].do_reset : live[0], lv1 = reset_pass ? o[1].do_reset : live[1];
      // ---- phase 1, per group: set-up; a group on its feet hands rows / factorisation to the workspace; a group that needs the eight-row
      // sweep is carried through at once (pair layout), finish included
      LEG_NOUNROLL
      for (int g = 0; g < 2; g++) {
        const M lv = g == 0 ? lv0 : lv1;
        bool ran_ = false, join_ = false, join8_ = false;
        M ovf_ = none;
        I nit_ = 0;
        if (B::any(lv)) {
          ran_ = true;
          const Io io = io_of(g);
          lds.select(g, io);
          const int base = g * W_GROUP;
          Lane st;
          lds.mark(0);    // (profiling builds: the time up to a mark goes to its bucket; 0 = glue, bookkeeping, outputs)
          get_lane(ws, base, st);
          B::fence();
          lds.mark(15);   // 15 = state in, before the set-up (sub_setup's own marks: 1 kinematics .. 5 rows)
          Sub S;
          SubOut so;
          // The rows leave for the workspace INSIDE the branch of the set-up that built them (six slots on their feet / eight): with B::SPLIT_TAIL the two
          // cases never meet again, so nothing of a row has to survive a merge of the two paths (and a group on its feet does not build its two empty slots).
          auto rows_small = [&]() {
            B::fence();
            put_rows(ws, base, S); put_keep(ws, base, S); put_misc(ws, base, S.go, S.ncon);
            lds.mark(10);   // 10 = rows / factorisation out
          };
          auto rows_general = [&]() {
#if DUO_JOINT8
            // some environment of the group has a joint limit active or a third pair: all eight row slots go to the workspace, with what the finish
            // needs of the once-per-wavefront descriptors, for the eight-row joint sweep
            B::fence();
            put_rows8(ws, base, S); put_keep(ws, base, S); put_misc(ws, base, S.go, S.ncon);
            B::wst(ws, base + W_MISC + 3, B::toD(S.nlim));
            put_desc<HF>(lds, ws, base);
#endif
          };
          C::template sub_setup<MODE, HF>(lds, st, reset_pass || MODE == 2, lv, !reset_pass, so, S, hf, rows_small, rows_general);
          ovf_ = so.overflow;
          B::fence();
          if (S.small) {
            join_ = true;
          } else {
#if DUO_JOINT8
            join8_ = true;
#else
            C::sub_sweeps(S);
            B::fence();
            nit_ = S.niter;
            C::template sub_finish<HF>(lds, st, !reset_pass, S);
            put_lane(ws, base, st);
#endif
          }
          B::fence();
        }
        if (g == 0) { ran[0] = ran_; join[0] = join_; join8[0] = join8_; ovf[0] = ovf_; nit[0] = nit_; }
        else { ran[1] = ran_; join[1] = join_; join8[1] = join8_; ovf[1] = ovf_; nit[1] = nit_; }
      }
      // ---- phase 2: one joint sweep for the groups on their feet; the forces go back into the rows' force slots
#if DUO_JOINT8
      const bool eight = join8[0] || join8[1];
      if (eight) {
        // the eight-row joint sweep: every group of the wavefront in the eight-row format (a group on its feet is widened, a group that does not
        // run gets empty rows), one sweep, forces back
        lds.mark(0);
        LEG_NOUNROLL
        for (int g = 0; g < 2; g++) {
          const bool j6 = g == 0 ? join[0] : join[1], j8 = g == 0 ? join8[0] : join8[1];
          if (j6) widen_rows(ws, g * W_GROUP);
          else if (!j8) put_idle8(ws, g * W_GROUP);
          B::fence();
        }
        lds.mark(11);
        const bool wr[2] = {join[0] || join8[0], join[1] || join8[1]};
        joint_solve8(ws, wr);
        B::fence();
        lds.mark(7);
      }
      if (!eight && (join[0] || join[1])) {
#else
      if (join[0] || join[1]) {
#endif
        Sub S[2];
        lds.mark(0);
        lfor<0, 2>([&](auto gg) {
          constexpr int G = decltype(gg)::value;
          if (join[G]) { get_rows(ws, G * W_GROUP, S[G]); get_misc(ws, G * W_GROUP, S[G].go, S[G].ncon); }
          else idle_rows(S[G]);
        });
        B::fence();
        lds.mark(11);   // 11 = rows of both groups in
#ifdef DUO_VIEW_EXPERIMENT
        if (cfg.flags & CASSIE_DUO_VIEW_FLAG) joint_solve_view(ws, join);
        else
#endif
        {
        joint_solve(S, join);
        B::fence();
        lds.mark(7);    // 7 = transpose + joint sweeps + forces back
        lfor<0, 2>([&](auto gg) {
          constexpr int G = decltype(gg)::value;
          if (join[G]) {
            put_forces(ws, G * W_GROUP, S[G]);
            B::wst(ws, G * W_GROUP + W_MISC + 2, B::toD(S[G].niter));
          }
        });
        }
        B::fence();
        lds.mark(12);   // 12 = forces out
      }
      if (join[0] || join[1] || join8[0] || join8[1]) {
        // ---- phase 3, per group: finish
        LEG_NOUNROLL
        for (int g = 0; g < 2; g++) {
          const bool j8 = g == 0 ? join8[0] : join8[1];
          if (!((g == 0 ? join[0] : join[1]) || j8)) continue;
          const Io io = io_of(g);
          lds.select(g, io);
          const int base = g * W_GROUP;
          Sub S1;
          Lane st;
          get_lane(ws, base, st);
          get_keep(ws, base, S1);
          get_misc(ws, base, S1.go, S1.ncon);
          const I nit_ = B::toint(B::wld(ws, base + W_MISC + 2));
          S1.nlim = 0;
#if DUO_JOINT8
          if (j8) {
            // a group of the eight-row sweep: all eight forces, its joint limits, and the once-per-wavefront descriptors back in LDS
            D t[CAP];
            get_block(ws, base + W_ROWS, t);
            lfor<0, CAP>([&](auto ii) { constexpr int Ii = decltype(ii)::value; S1.f[Ii] = t[Ii]; });
            S1.nlim = B::toint(B::wld(ws, base + W_MISC + 3));
            get_desc<HF>(lds, ws, base);
          } else
#endif
          {
          get_forces(ws, base, S1);
          lfor<NR, CAP>([&](auto ii) { constexpr int Ii = decltype(ii)::value; S1.f[Ii] = 0.0; });   // slots 6, 7 are empty in a group on its feet
          }
          S1.leg = B::opq(B::leg());
          S1.K = B::kbase(S1.leg);
          // row kinds as sub_setup assigns them: slots 0, 1 connect; pair P at slots 2 + 2P / 3 + 2P while ncon > P; joint limit j at slot 7 - j while
          // nlim > j (a group on its feet: nlim = 0, ncon <= 2)
          S1.kind[0] = B::seli(S1.go, I(K_EQ), I(K_NONE)); S1.kind[1] = S1.kind[0];
          lfor<2, CAP>([&](auto ss) {
            constexpr int S_ = decltype(ss)::value;
            constexpr int P = (S_ - 2) >> 1, ODD = (S_ - 2) & 1, LJ = 7 - S_;
            const M isc = S1.ncon > I(P), isl = S1.nlim > I(LJ);
            S1.kind[S_] = B::seli(S1.go, B::seli(isc, I(ODD ? K_CT : K_CN), B::seli(isl, I(K_LIM), I(K_NONE))), I(K_NONE));
          });
          B::fence();
          lds.mark(13);   // 13 = state / factorisation / forces in, before the finish (sub_finish's own marks: 8 generalised force, 9 M^-1, damping, integration)
          C::template sub_finish<HF>(lds, st, !reset_pass, S1);
          put_lane(ws, base, st);
          B::fence();
          lds.mark(14);   // 14 = state out
          if (g == 0) nit[0] = nit_; else nit[1] = nit_;
        }
      }
      if (!reset_pass) {
        bool more = false;
        lfor<0, 2>([&](auto gg) {
          constexpr int G = decltype(gg)::value;
          if (ran[G]) {
            const M ov = live[G] & ovf[G];
            o[G].pend = B::seli(ov, I(cfg.n_sub - sub + cfg.pend_extra), o[G].pend);
            live[G] = live[G] & !ov;
            o[G].niter = o[G].niter + B::seli(live[G], nit[G], I(0));
            o[G].set_state = o[G].set_state | live[G];
            more = more || B::any(live[G]);
          }
        });
        sub++;
        if (sub < cfg.n_sub && more) continue;
      }
      if (!cfg.want_obs) break;
      lds.snapshot(true);
      bool again = false;
      lfor<0, 2>([&](auto gg) {
        constexpr int G = decltype(gg)::value;
        const Io io = io_of(G);
        lds.select(G, io);
        Lane st;
        lds.mark(17);   // 17 = bookkeeping after the last substep
        get_lane(ws, G * W_GROUP, st);
        if (C::step_outputs(cfg, lds, io, st, live[G], o[G], reset_pass)) again = true;
        put_lane(ws, G * W_GROUP, st);
        B::fence();
        lds.mark(reset_pass ? 19 : 18);   // 18 = outputs of the step, 19 = outputs of the reset pass
      });
      if (reset_pass || !again) break;
      reset_pass = true;
    }
    // ---- state write-back: q, v, warm start, clock, iteration count; the setState snapshot, the motor commands and qstate are in
    // the record already (written where they changed)
    lds.mark(20);
    lfor<0, 2>([&](auto gg) {
      constexpr int G = decltype(gg)::value;
      const Io io = io_of(G);
      lds.select(G, io);
      Lane st;
      get_lane(ws, G * W_GROUP, st);
      lfor<0, 3>([&](auto bb) {
        constexpr int Bc = decltype(bb)::value;
        const M lv = valid[G] & left;
        B::pst(io.rec, I(ES_Q + Bc), st.qb[Bc], lv); B::pst(io.rec, I(ES_V + Bc), st.vb[Bc], lv); B::pst(io.rec, I(ES_WS + Bc), st.wb[Bc], lv);
        B::pst(io.rec, I(ES_QSTATE + Bc), D(cp_env_qinit[Bc]), lv & o[G].do_reset);
      });
      lfor<0, 5>([&](auto dd) {
        constexpr int Dd = decltype(dd)::value;
        B::pst(io.rec, lo + (ES_Q + Dd), st.ql[Dd], valid[G]); B::pst(io.rec, lo + (ES_V + Dd), st.vl[Dd], valid[G]); B::pst(io.rec, lo + (ES_WS + Dd), st.wl[Dd], valid[G]);
      });
      B::pst(io.rec, I(ES_TIME), lds.cld(C::C_TIME), valid[G] & left);
      B::pst(io.rec, I(ES_NITER), B::toD(o[G].niter), valid[G] & left);
      B::pst(io.rec, I(ES_QPWSET), D(0.0), valid[G] & left & o[G].do_reset);
    });
    lds.mark(21);   // 21 = write-back
  }
};

}  // namespace leg
}  // namespace cassie
#endif
