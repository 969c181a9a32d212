// cassie_kernels_g16.hip -- 4 environments per wavefront (one per 16-lane DPP row).
//
// Round-1 PMC profile of the wave-per-environment kernel (profiles/r01_b_pmc, r01_c_pmc): the SIMDs are ~70 % busy issuing
// FP64 VALU instructions, and in the dominant PGS loop each of those instructions carries ONE useful lane (the owner of the
// constraint row being updated).  This kernel packs four environments into a wavefront so that one instruction stream
// updates four rows -- one per environment -- at a time:
//   * the 16 lanes of a DPP row own one environment; every cross-lane move inside an environment is a DPP row operation
//     (row_newbcast of the pivot row / of a force delta, quad_perm for the normal<->tangent partner, row_ror reductions);
//   * active constraint rows are COMPACTED onto the 16 lanes in MuJoCo order (4 connect rows, active joint limits, a pad
//     row so that contact pairs start on an even lane, active contact (normal,tangent) pairs); lane i keeps the 16 entries
//     of row i of A in registers;
//   * step K of a PGS sweep updates row K of all four environments; row kinds may differ between environments, so both the
//     single-row and the contact-pair code run under wave-uniform branches with per-environment predicates;
//   * an environment with more than 16 active rows (robot on the ground with many contacts/limits) is NOT handled here: its
//     state is left untouched from that substep on and `pending[env]` tells the wave-per-environment kernel
//     (env_step_kernel, clean-up pass) how many substeps are left.  Results are identical either way.
// Same reference call sites as cassie_kernels.hip (Cassie2d::StepPd/Step + mj_step + Cassie2dEnv.step).
#ifndef CASSIE_KERNELS_G16_HIP_
#define CASSIE_KERNELS_G16_HIP_

namespace cassie {
namespace g16 {

constexpr int MAXR = 16;
enum { RK_SKIP = 0, RK_EQ = 1, RK_LIM = 2, RK_CN = 3, RK_CT = 4 };

// LDS of one environment (4.5 KB; 4 per wavefront, 8 wavefronts per CU).  Buffers whose lifetimes do not overlap inside a
// substep share storage.  kq2..tim are per-lane values that live across all substeps of an Env.step; they sit in LDS rather
// than in VGPRs because the register allocator would otherwise spill exactly those (long-lived, rarely used) to scratch.
struct EnvLds {
  double q[16], v[16], ws[16], ctrl[8];
  double lc[12], ls[12], lw[12], lox[12], loz[12], lcx[12], lcz[12], lfx[12], lfz[12];
  double qs[16];
  double kq2[16], kv2[16];  // qpos/qvel at the last DynamicModel::setState (ES_KQ/ES_KV)
  double qst[16], actl[16], ctl[16], tim[2];  // self.qstate, this step's action per dof lane, last ctrl per dof lane, env time
  union {
    struct { double s1x[16], s1z[16], s2[16]; };          // mass_rows exchange (dead once the M rows are built)
    struct { double rowJ[MAXR][8]; int rowleg[MAXR]; };   // constraint rows (from the row build to the end of the substep)
  };
  union {
    struct { double minv[NV * NV + 7]; };                 // M^-1 (from the Gauss-Jordan to the end of the substep)
    struct { double lvx[12], lvz[12], lax[12], laz[12], site[2][6][4], s18[18], kq[16], kv[16]; };  // FK by-products / post-step
  };
};

template <int K> __device__ __forceinline__ int row_bcast_int(int x) {
  return __builtin_amdgcn_mov_dpp(x, 0x150 + K, 0xF, 0xF, false);
}

// position of the n-th set bit of mask (n counted from 0); -1 if there is none
__device__ __forceinline__ int nth_set_bit(unsigned mask, int n) {
  int found = -1, cnt = 0;
#pragma unroll
  for (int i = 0; i < 17; i++) {
    if ((mask >> i) & 1u) { if (cnt == n) found = i; cnt++; }
  }
  return found;
}

struct G16Out { int niter; bool overflow; };

#ifdef CASSIE_MFMA
// -DCASSIE_MFMA: A = J M^-1 J' through v_mfma_f64_16x16x4 (A/B experiment asked for by the north star).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
typedef unsigned mfma_u2 __attribute__((ext_vector_type(2)));
// a' = [a.lanes0-31, b.lanes0-31], b' = [a.lanes32-63, b.lanes32-63]   (v_permlane32_swap, gfx950)
__device__ __forceinline__ void swap32(double& a, double& b) {
  mfma_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  mfma_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
}
// a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3]   (v_permlane16_swap, gfx950)
__device__ __forceinline__ void swap16(double& a, double& b) {
  mfma_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  mfma_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
}
// 4x4 transpose between (register index) and (16-lane row index): afterwards register k of row G holds what register G of row k held
__device__ __forceinline__ void transpose_regs_rows(double& r0, double& r1, double& r2, double& r3) {
  swap32(r0, r2); swap32(r1, r3); swap16(r0, r1); swap16(r2, r3);
}
#endif

// ---------------------------------------------------------------- one mj_forward (+ optional Euler step) for 4 envs
// l = lane & 15, g = lane >> 4.  `live` (uniform inside a row) masks environments that must not be touched.
// `integrate` (wave-uniform) = false gives mj_forward only (Cassie2d::Reset); it is a run-time flag so that a kernel carries ONE
// copy of this code (two copies doubled the code size and the register spills around the second one).
// (s0, s1) -= E x for the ten damped dofs 3..12: s -= me[j] * (lane 3+j of x's row), alternating between two accumulators.
// Assembly for the reason given at gj_elim7 (cassie_kernels.hip); x may have been written by the instruction before, hence
// the leading s_nop; s0/s1 are consumed by an ordinary add.
constexpr int IMPLICIT_DAMPING_SWEEPS = 12;
#define DM_FMAC(acc, j, lane) "v_fmac_f64_dpp %" #acc ", %[x], -%" #j " row_newbcast:" #lane " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void damping_matvec(double& s0, double& s1, double x, const double (&me)[10]) {
  asm("s_nop 1\n\t" DM_FMAC(0, 2, 3) DM_FMAC(1, 3, 4) DM_FMAC(0, 4, 5) DM_FMAC(1, 5, 6) DM_FMAC(0, 6, 7) DM_FMAC(1, 7, 8) DM_FMAC(0, 8, 9)
      DM_FMAC(1, 9, 10) DM_FMAC(0, 10, 11) DM_FMAC(1, 11, 12)
      : "+v"(s0), "+v"(s1)
      : "v"(me[0]), "v"(me[1]), "v"(me[2]), "v"(me[3]), "v"(me[4]), "v"(me[5]), "v"(me[6]), "v"(me[7]), "v"(me[8]), "v"(me[9]), [x] "v"(x));
}
#undef DM_FMAC
// acc += y * (lane K of x's 16-lane row).  (The fused v_fmac_f64_dpp form used by the Gauss-Jordan was tried here too: 4 % fewer
// instructions, no time -- a sweep is one dependent chain, not issue-bound.)
template <int K> __device__ __forceinline__ void fmac_bcast(double& acc, double x, double y) {
#pragma clang fp contract(off)
  acc = __builtin_fma(row_bcast<K>(x), y, acc);
}
template <class SM, bool HF = false>
__device__ __forceinline__ void substep(SM& sm, const LaneConst& c_in, int l, int g, double ctrl, bool live, bool integrate, G16Out& out,
                                        const Terrain* terrain = nullptr, PhaseClock* pc = nullptr) {
  // Per-lane model constants are re-read from constant memory in every substep (K$/L1 hits).  Without these barriers the
  // compiler hoists ~50 loop-invariant table loads out of the substep loop and then spills them to scratch, which costs
  // HBM write traffic at every kernel boundary (profiles/r01_c_pmc: 32 MB per launch against 3.7 MB algorithmic).
  l = opaque(l);
  LaneConst c = c_in;
  c.ancmask = opaque(c.ancmask); c.d = opaque(c.d); c.dlink = opaque(c.dlink); c.submask = opaque(c.submask);
  c.rel = opaque(c.rel); c.act = opaque(c.act); c.kL = opaque(c.kL); c.kR = opaque(c.kR);
  // ---- kinematics, mass matrix, both inverses
  PHASE_MARK(*pc, 0);
  planar_fk<0>(sm, sm.q, sm.v, c, l);
  PHASE_MARK(*pc, 1);
  double tau, qs;
  {
    double Mi[NV];
    DofConst dc;
    load_dof_const(dc, c);
    double bias;
    mass_rows<0>(sm, c, dc, l, Mi, bias, false);
    gauss_jordan_rows_legs<true>(Mi, l);
    if (c.dvalid) { static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; sm.minv[c.d * NV + C] = Mi[C]; }); }
    double v_d = sm.v[c.d < NV ? c.d : 0];
    double u = ctrl < dc.clo ? dc.clo : (ctrl > dc.chi ? dc.chi : ctrl);
    tau = -dc.damping * v_d - bias + dc.gear * u;
    double qs0 = 0.0, qs1 = 0.0;  // two partial sums: halves the dependent-FMA chain
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; if constexpr (C & 1) qs1 += Mi[C] * row_bcast<C>(tau); else qs0 += Mi[C] * row_bcast<C>(tau); });
    qs = qs0 + qs1;
  }
  if (c.dvalid) sm.qs[c.d] = qs;
  lds_sync();
  PHASE_MARK(*pc, 2);
  // ---- which constraints are active: limits on lanes 0..7, collision spheres on lanes 0..15 (+ sphere 16 on lane 0)
  const double basez = sm.q[1] - cp_qpos0[1] + cp_link_off[0][0][1];
  bool lim_act = false;
  if (l < 8) {
    const int dof = opaque(cp_slot_dof[SLOT_LIM + l]);
    double qd = sm.q[dof];
    lim_act = (qd - cp_jnt_range[dof][0] < 0) || (cp_jnt_range[dof][1] - qd < 0);
  }
  const double basex = HF ? sm.q[0] - cp_qpos0[0] + cp_link_off[0][0][0] : 0.0;
  auto sphere_active = [&](int sph) {
    const int so = opaque(sph);
    const int lk = cp_sph_link[so];
    double cx, cz;
    link_point(sm, lk, cp_sph_d[so][0], cp_sph_d[so][1], cx, cz);
    if constexpr (HF) {
      double dist, nx, nz;
      terrain_sphere(*terrain, basex + cx, cp_sph_y[so], basez + cz, cp_sph_r[so], dist, nx, nz);
      return dist < 0;
    } else {
      return basez + cz - cp_sph_r[so] < 0;
    }
  };
  const bool con_act0 = sphere_active(l);
  const bool con_act1 = (l == 0) ? sphere_active(16) : false;
  const unsigned long long bl = __ballot(lim_act), b0 = __ballot(con_act0), b1 = __ballot(con_act1);
  const unsigned lim_mask = (unsigned)(bl >> (16 * g)) & 0xFFu;
  const unsigned con_mask = ((unsigned)(b0 >> (16 * g)) & 0xFFFFu) | ((((unsigned)(b1 >> (16 * g))) & 1u) << 16);
  const int nlim = __popc(lim_mask), ncon = __popc(con_mask);
  // Contact pairs start on an even row behind the limits.  If the environments of a wave had their pairs on different rows, a
  // sweep would run a pair step for every row ANY of them uses, so pairs are put on row 8 wherever that is possible: an
  // environment with 1..4 limits and <= 4 contacts always uses row 8 (pad rows behind its limits) -- its layout, and with it the
  // grouping of its partial sums, depends on nothing but its own state; an environment without limits moves from row 4 to row 8
  // when a neighbour needs it, which changes no bit of its result (the pad rows add exact zeros to the same partial sums, the
  // Gauss-Seidel order is the same).  Everything else is compacted.
  const bool own8 = live && nlim >= 1 && nlim <= 4 && ncon <= 4;
  const bool shift8 = live && nlim == 0 && ncon <= 4 && __ballot(own8) != 0;
  const int cbase = (own8 || shift8) ? 8 : ((4 + nlim + 1) & ~1);
  const int nrows = cbase + 2 * ncon;
  const bool ovf_here = live && nrows > MAXR;
  out.overflow = ovf_here;
  const bool go = live && !ovf_here;  // this environment is processed in this substep
  // ---- the row owned by this lane
  int kind = RK_SKIP, slot = 0;
  if (go) {
    if (l < 4) { kind = RK_EQ; slot = l; }
    else if (l < 4 + nlim) { kind = RK_LIM; slot = SLOT_LIM + nth_set_bit(lim_mask, l - 4); }
    else if (l >= cbase && l < nrows) {
      int j = (l - cbase) >> 1, odd = (l - cbase) & 1;
      kind = odd ? RK_CT : RK_CN;
      slot = SLOT_CON + 2 * nth_set_bit(con_mask, j) + odd;
    }
  }
  PHASE_MARK(*pc, 3);
  double b = 0.0, jar = 0.0, R = 1.0;
  const bool active = kind != RK_SKIP;
#ifndef CASSIE_MFMA
  double X[NV];
#endif
  int leg = 0;
  {
    const int so = opaque(slot);
    leg = cp_slot_leg[so];
    const int comp = cp_slot_comp[so], rdof = cp_slot_dof[so], link1 = cp_slot_link1[so], link2 = cp_slot_link2[so];
    const int pm1 = cp_link_pathmask8[link1], pm2 = cp_link_pathmask8[link2];
    const double d1x = cp_slot_d1[0][so][0], d1z = cp_slot_d1[0][so][1], d2x = cp_slot_d2[0][so][0], d2z = cp_slot_d2[0][so][1];
    const double radius = cp_slot_radius[so], invw = cp_slot_invweight[so];
    const double* sr = kind == RK_EQ ? cp_eq_solref[so >> 1] : (kind == RK_LIM ? cp_limit_solref : cp_contact_solref);
    const double* si = kind == RK_EQ ? cp_eq_solimp[so >> 1] : (kind == RK_LIM ? cp_limit_solimp : cp_contact_solimp);
    const double solref0 = sr[0], solref1 = sr[1], simp0 = si[0], simp1 = si[1], simp2 = si[2];
    double J[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double pos = 0.0;
    const int legbase = leg == 0 ? 1 : 6;
    const int vbase = leg == 0 ? 3 : 8;
    if (kind == RK_EQ) {
      double p1x, p1z, p2x, p2z;
      link_point(sm, link1, d1x, d1z, p1x, p1z);
      link_point(sm, link2, d2x, d2z, p2x, p2z);
      jac_compact(sm, pm1, legbase, comp, p1x, p1z, 1.0, J);
      jac_compact(sm, pm2, legbase, comp, p2x, p2z, -1.0, J);
      pos = comp == 0 ? p1x - p2x : p1z - p2z;
    } else if (kind == RK_LIM) {
      const int rd = rdof >= 0 ? rdof : 0;
      double qd = sm.q[rd];
      double dlo = qd - cp_jnt_range[rd][0], dhi = cp_jnt_range[rd][1] - qd;
      int k = 3 + rd - vbase;
      double sgn = 0.0;
      if (dlo < 0) { pos = dlo; sgn = 1.0; }
      else if (dhi < 0) { pos = dhi; sgn = -1.0; }
      static_for<3, 8>([&](auto kk) { constexpr int K = decltype(kk)::value; J[K] = (k == K) ? sgn : 0.0; });
    } else if (kind == RK_CN || kind == RK_CT) {
      double cx, cz;
      link_point(sm, link1, d1x, d1z, cx, cz);
      if constexpr (HF) {
        // terrain: contact frame from the cell under the sphere; normal row along (nx, nz), tangent row along (nz, -nx)
        const int sph = opaque(so >= SLOT_CON ? (so - SLOT_CON) >> 1 : 0);
        double dist, nx, nz;
        terrain_sphere(*terrain, basex + cx, cp_sph_y[sph], basez + cz, radius, dist, nx, nz);
        const double back = radius + 0.5 * dist;
        const double px = cx - nx * back, pz = cz - nz * back;
        const double dirx = kind == RK_CN ? nx : nz, dirz = kind == RK_CN ? nz : -nx;
        jac_compact(sm, pm1, legbase, 0, px, pz, dirx, J);
        jac_compact(sm, pm1, legbase, 1, px, pz, dirz, J);
        pos = dist;
      } else {
        double dist = basez + cz - radius;
        double pz = 0.5 * dist - basez;
        jac_compact(sm, pm1, legbase, comp, cx, pz, 1.0, J);
        pos = dist;
      }
    }
    double vel = J[0] * sm.v[0] + J[1] * sm.v[1] + J[2] * sm.v[2];
    double bq = J[0] * sm.qs[0] + J[1] * sm.qs[1] + J[2] * sm.qs[2];
    double jw = J[0] * sm.ws[0] + J[1] * sm.ws[1] + J[2] * sm.ws[2];
    static_for<0, 5>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      vel += J[3 + K] * sm.v[vbase + K]; bq += J[3 + K] * sm.qs[vbase + K]; jw += J[3 + K] * sm.ws[vbase + K];
    });
    double tc = solref0 < 2.0 * H ? 2.0 * H : solref0;
    double kk_ = 1.0 / (simp1 * simp1 * tc * tc * solref1 * solref1), bb_ = 2.0 / (simp1 * tc);
    double imp = impedance(simp0, simp1, simp2, pos);
    R = (1.0 - imp) / imp * invw;
    R = R > MINVAL ? R : MINVAL;
    double own_pos = kind == RK_CT ? 0.0 : pos;
    double imp_own = kind == RK_CT ? impedance(simp0, simp1, simp2, 0.0) : imp;
    double aref = -bb_ * vel - kk_ * imp_own * own_pos;
    b = active ? bq - aref : 0.0;
    jar = jw - aref;
    static_for<0, 8>([&](auto kk) { constexpr int K = decltype(kk)::value; sm.rowJ[l][K] = active ? J[K] : 0.0; });
    sm.rowleg[l] = leg;
#ifndef CASSIE_MFMA
    const double* mi = sm.minv;
    static_for<0, NV>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      double sx = mi[C * NV + 0] * J[0] + mi[C * NV + 1] * J[1] + mi[C * NV + 2] * J[2];
      static_for<0, 5>([&](auto kk) { constexpr int K = decltype(kk)::value; sx += mi[C * NV + vbase + K] * J[3 + K]; });
      X[C] = sx;
    });
#endif
  }
  lds_sync();
  // ---- row of A = J M^-1 J' + R in registers (16 columns = the 16 row lanes of this environment)
  double Ac[MAXR];
  double Adiag = 1.0, Ant = 0.0;
#ifdef CASSIE_MFMA
  {
    // Per environment gg of the wave, two 16x16x16 products on the matrix cores, K = dof index padded 13 -> 16 (4 MFMAs each):
    //   X' = Minv J'  (M index = dof, N index = constraint row)   then   A = X J'  (M = row, N = row),
    // all 64 lanes reading the operands of environment gg from ITS LDS.  Operand layouts of v_mfma_f64_16x16x4 (probed on
    // gfx950): A-operand lane i = A[i % 16][i / 16], B-operand lane i = B[i / 16][i % 16], result register j of lane i =
    // D[4 j + i / 16][i % 16].  With that layout the result of the first product IS the A-operand of the second
    // (X[r][4 s + i/16] sits in register s of lane 16 (i/16) + r), so nothing moves in between.  The second result holds, on
    // lane 16 k + c, A_gg[4 j + k][c] = A_gg[c][4 j + k]: row c's entries are spread over the four 16-lane rows, and one 4x4
    // (register <-> row) transpose per j -- v_permlane32_swap + v_permlane16_swap, no LDS -- brings every environment's
    // rows back to its own lanes.
    SM* const base = &sm - g;
    mfma_d4 Dres[4];
    static_for<0, 4>([&](auto ggc) {
      constexpr int GG = decltype(ggc)::value;
      const SM& se = base[GG];
      const int legr = se.rowleg[l];
      double bop[4];
      mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
      static_for<0, 4>([&](auto sc) {
        constexpr int S = decltype(sc)::value;
        const int j = 4 * S + g;  // dof index carried by this lane in K-chunk S
        const int idx = j < 3 ? j : (legr == 0 ? (j < 8 ? j : -1) : ((j >= 8 && j < NV) ? j - 5 : -1));
        bop[S] = idx >= 0 ? se.rowJ[l][idx < 0 ? 0 : idx] : 0.0;
        const double aop = (l < NV && j < NV) ? se.minv[(l < NV ? l : 0) * NV + (j < NV ? j : 0)] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop[S], acc, 0, 0, 0);
      });
      mfma_d4 acc2 = {0.0, 0.0, 0.0, 0.0};
      static_for<0, 4>([&](auto sc) {
        constexpr int S = decltype(sc)::value;
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[S], bop[S], acc2, 0, 0, 0);
      });
      Dres[GG] = acc2;
    });
    static_for<0, 4>([&](auto jc) {
      constexpr int Jj = decltype(jc)::value;
      double t0 = Dres[0][Jj], t1 = Dres[1][Jj], t2 = Dres[2][Jj], t3 = Dres[3][Jj];
      transpose_regs_rows(t0, t1, t2, t3);
      Ac[4 * Jj + 0] = t0; Ac[4 * Jj + 1] = t1; Ac[4 * Jj + 2] = t2; Ac[4 * Jj + 3] = t3;
    });
    static_for<0, MAXR>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      double a = active ? Ac[K] : 0.0;
      if (l == K && active) { a += R; Adiag = a; }
      if (l + 1 == K && kind == RK_CN) Ant = a;
      Ac[K] = a;
    });
  }
#else
  PHASE_MARK(*pc, 4);
  static_for<0, MAXR>([&](auto kk) {
    constexpr int K = decltype(kk)::value;
    const double* js = sm.rowJ[K];
    const int lg = sm.rowleg[K];
    double a = X[0] * js[0] + X[1] * js[1] + X[2] * js[2];
    double al = X[3] * js[3] + X[4] * js[4] + X[5] * js[5] + X[6] * js[6] + X[7] * js[7];
    double ar = X[8] * js[3] + X[9] * js[4] + X[10] * js[5] + X[11] * js[6] + X[12] * js[7];
    a += lg == 0 ? al : ar;
    if (!active) a = 0.0;
    if (l == K && active) { a += R; Adiag = a; }
    if (l + 1 == K && kind == RK_CN) Ant = a;
    Ac[K] = a;
  });
#endif
  PHASE_MARK(*pc, 5);
  const double Ainv = 1.0 / Adiag;
  const double Apart = swap1(Adiag);
  // ---- warm start (mj_constraintUpdate) kept only if its dual cost beats zero force
  const double mu = CP_CONTACT_MU;
  double f = 0.0;
  {
    double D = 1.0 / R;
    double pj = swap1(jar);
    if (kind == RK_EQ) f = -D * jar;
    else if (kind == RK_LIM) f = jar < 0 ? -D * jar : 0.0;
    else if (kind == RK_CN || kind == RK_CT) {
      double jn = kind == RK_CN ? jar : pj, jt = kind == RK_CN ? pj : jar;
      double N = jn * mu, U1 = jt * mu, T = fabs(U1);
      double fn, ft;
      if (N >= mu * T || (T <= 0 && N >= 0)) { fn = 0; ft = 0; }
      else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { fn = -D * jn; ft = -D * jt; }
      else {
        double Dm = D / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
        fn = -Dm * NmT * mu;
        ft = -fn / T * U1 * mu;
      }
      f = kind == RK_CN ? fn : ft;
    }
  }
  double res = 0.0;
  {
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
    static_for<0, MAXR>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      double t = Ac[K] * row_bcast<K>(f);
      if constexpr ((K & 3) == 0) r0 += t; else if constexpr ((K & 3) == 1) r1 += t; else if constexpr ((K & 3) == 2) r2 += t; else r3 += t;
    });
    res = (r0 + r1) + (r2 + r3);
  }
  {
    double cost = row_sum(active ? f * (0.5 * res + b) : 0.0);
    if (cost > 0) { f = 0.0; res = 0.0; }
  }
  res += b;
  PHASE_MARK(*pc, 6);
  // ---- PGS sweeps; step K updates row K of every environment of the wave
  const double scale = 1.0 / (CP_MEANINERTIA * NV);
  const double AttInv = 1.0 / Apart;
  bool sweeping = go;       // uniform inside a row: this environment still iterates
  // wave-uniform "some environment has a single row / a contact pair at step K" masks: lane 16 g + K owns row K of environment g,
  // so the four 16-bit fields of one ballot are OR-ed together
  unsigned anyS, anyP;
  {
    const unsigned long long ms = __ballot(kind == RK_EQ || kind == RK_LIM), mp = __ballot(kind == RK_CN);
    anyS = (unsigned)((ms | (ms >> 16) | (ms >> 32) | (ms >> 48)) & 0xFFFFull);
    anyP = (unsigned)((mp | (mp >> 16) | (mp >> 32) | (mp >> 48)) & 0xFFFFull);
  }
  double acc = 0.0;  // cost change of the current sweep, summed over the environment's rows (the same value on all 16 lanes)
  // Every lane runs the update of ITS OWN row on its own (f, residual); at step K only lane K of a row owns a live result, and
  // its deltas are zeroed everywhere else (`mine`), so `f += d` needs no further selection and the deltas / the cost change
  // reach the other lanes by one 64-bit row_newbcast each.
  const bool isE = kind == RK_EQ, isLim = kind == RK_LIM, isN = kind == RK_CN;
  const double hAdiag = 0.5 * Adiag, hApart = 0.5 * Apart;
  // "my environment still iterates" folded into two per-lane constants, so that no step has to test it: a connect row of a
  // finished environment gets a zero step (AinvE = 0), a limit / contact row never passes the cost test (threshold -inf);
  // the thresholds are per row kind because single_step and pair_step both run on lane K when the environments of a wave differ.
  double AinvE = (sweeping & isE) ? Ainv : 0.0;
  double thrL = (sweeping & isLim) ? 1e-10 : -__builtin_inf(), thrN = (sweeping & isN) ? 1e-10 : -__builtin_inf();
  // The step functions are inlined at two call sites each (the straight-line sweep and the general sweep below).  Which of
  // the two a wavefront runs depends on ALL four of its environments, so their roundings must be identical or an environment's
  // result would depend on its neighbours (measured in r02: 1e-13 after two substeps).  Contraction is therefore off inside
  // them and every fused multiply-add is written out.
  // Rows 0..3 are the connect (equality) rows in EVERY environment.  Reduced form: no clamp, and no cost-increase revert -- for
  // an unclamped row d = -res / A exactly minimises its own quadratic, the change is -res^2 / (2 A) <= 0 with
  // A = J M^-1 J' + R > 0, so mj_solPGS's `if (change > 1e-10) revert` can never fire there.  `force -= res * ARinv` as written
  // in mj_solPGS (product rounded, then subtracted).
  auto eq_step = [&](auto kk) {
#pragma clang fp contract(off)
    constexpr int K = decltype(kk)::value;
    const double d = -(res * AinvE);                        // live on lane K; the broadcasts below read only that lane
    const double chg = d * __builtin_fma(hAdiag, d, res);
    if constexpr (K == 0) acc = row_bcast<K>(chg); else fmac_bcast<K>(acc, chg, 1.0);   // step 0 opens every sweep
    f += (l == K) ? d : 0.0;
    fmac_bcast<K>(res, d, Ac[K]);
  };
  // one joint-limit row at row K (single rows beyond row 3 are limits)
  auto single_step = [&](auto kk) {
#pragma clang fp contract(off)
    constexpr int K = decltype(kk)::value;
    const double cand = fmax(__builtin_fma(-res, Ainv, f), 0.0);
    double d = cand - f;
    double chg = d * __builtin_fma(hAdiag, d, res);
    const bool keep = (chg <= thrL) & (l == K);
    d = keep ? d : 0.0;
    chg = keep ? chg : 0.0;
    fmac_bcast<K>(acc, chg, 1.0);
    f += d;
    fmac_bcast<K>(res, d, Ac[K]);
  };
  // one elliptic contact pair at rows (K, K+1), K even; branch-free so that steps can be scheduled across each other
  auto pair_step = [&](auto kk) {
#pragma clang fp contract(off)
    constexpr int K = decltype(kk)::value;
    const double rt = row_bcast<K + 1>(res), ot = row_bcast<K + 1>(f);   // the tangent row's values (live on lane K)
    const double rn = res, on = f;
    const double Ann = Adiag, Att = Apart;
    // normal-only update (taken when the normal force is ~0)
    double fn_n = fmax(__builtin_fma(-rn, Ainv, on), 0.0);
    // ray update
    double denom = __builtin_fma(ot, __builtin_fma(Att, ot, Ant * on), on * __builtin_fma(Ant, ot, Ann * on));
    double x = -__builtin_fma(ot, rt, on * rn) * fast_rcp(denom);
    x = fmax(x, -1.0);
    x = denom >= MINVAL ? x : 0.0;
    const bool use_n = on < MINVAL;
    double fn = use_n ? fn_n : __builtin_fma(x, on, on);
    double ft = use_n ? 0.0 : __builtin_fma(x, ot, ot);
    // friction on one dimension: unconstrained minimiser unless it leaves the cone
    double bc = __builtin_fma(Ant, fn - on, __builtin_fma(-Att, ot, rt));
    double x0 = -bc * AttInv;
    double v1 = x0 * (1.0 / mu);
    double val = __builtin_fma(v1, v1, -(fn * fn));
    const bool on_cone = (val >= 1e-10) & (val * Att * (mu * mu) >= 2e-10 * (v1 * v1));
    double ftc = on_cone ? __builtin_copysign(mu * fn, x0) : x0;
    ft = fn >= MINVAL ? ftc : ft;
    double dn = fn - on, dt = ft - ot;
    // 1/2 d'A d + d'res, grouped so that only two operations wait for the tangent step
    double chg = __builtin_fma(dt, __builtin_fma(hApart, dt, __builtin_fma(Ant, dn, rt)), dn * __builtin_fma(hAdiag, dn, rn));
    const bool keep = (chg <= thrN) & (l == K);
    dn = keep ? dn : 0.0; dt = keep ? dt : 0.0;
    chg = keep ? chg : 0.0;
    fmac_bcast<K>(acc, chg, 1.0);
    const double Dt = row_bcast<K>(dt);   // evaluated by every lane, BEFORE the selection (a DPP read needs its source lane active)
    f += (l == K + 1) ? Dt : dn;
    fmac_bcast<K>(res, dt, Ac[K + 1]);
    fmac_bcast<K>(res, dn, Ac[K]);
  };
  // common configuration (robot on its feet): no active joint limit and at most 4 contacts in every environment of the
  // wave => rows are exactly 4 connect rows + pairs at rows 4,6,8,10: straight-line sweep without per-step branches
  const bool simple = __ballot(go && (nlim != 0 || ncon > 4)) == 0;
  const bool pair4 = (anyP >> 4) & 1u, pair6 = (anyP >> 6) & 1u, pair8 = (anyP >> 8) & 1u, pair10 = (anyP >> 10) & 1u;
  int niter = 0;
  for (int iter = 0; iter < CP_ITERATIONS; iter++) {
    if (__ballot(sweeping) == 0) break;
    acc = 0.0;
    if (simple) {
      eq_step(IC<0>{}); eq_step(IC<1>{}); eq_step(IC<2>{}); eq_step(IC<3>{});
      if (pair4) pair_step(IC<4>{});
      if (pair6) pair_step(IC<6>{});
      if (pair8) pair_step(IC<8>{});
      if (pair10) pair_step(IC<10>{});
    } else {
      static_for<0, MAXR>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        if ((anyS >> K) & 1u) { if constexpr (K < 4) eq_step(kk); else single_step(kk); }
        if constexpr ((K & 1) == 0 && K + 1 < MAXR) {
          if ((anyP >> K) & 1u) pair_step(kk);
        }
      });
    }
    const double improvement = -acc;
    if (sweeping) {
      niter = iter + 1;
      if (improvement * scale < CP_TOLERANCE) { sweeping = false; AinvE = 0.0; thrL = thrN = -__builtin_inf(); }
    }
  }
  PHASE_MARK(*pc, 7);
  out.niter = niter;
  // ---- g = tau + J' f on the dof lanes, both accelerations, integration
  double gg = tau;
  {
    const int kL = c.kL, kR = c.kR;
    double g0 = 0.0, g1 = 0.0;
    static_for<0, MAXR>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      const int k = sm.rowleg[K] == 0 ? kL : kR;
      double js = sm.rowJ[K][k < 0 ? 0 : k];
      double t = (k < 0 ? 0.0 : js) * row_bcast<K>(f);
      if constexpr (K & 1) g1 += t; else g0 += t;
    });
    gg += g0 + g1;
  }
  double qacc = 0.0, qacch = 0.0;
  // qacc = M^-1 g, with this lane's row of M^-1 re-read from LDS (rowJ, overlaid by mass_rows' exchange buffers, is dead here)
  const int mrow = (c.dvalid ? c.d : 0) * NV;
  {
    double a0 = 0.0, a1 = 0.0;
    static_for<0, NV>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      const double gc = row_bcast<C>(gg);
      const double mi = sm.minv[mrow + C];
      if constexpr (C & 1) a1 += mi * gc; else a0 += mi * gc;
    });
    qacc = a0 + a1;
  }
  // Implicit joint damping of mj_Euler: qacch = (M + h B)^-1 g.  Up to r02_f this was a second mass-matrix pass and a second
  // Gauss-Jordan inversion per substep (~1000 instructions).  With E = M^-1 h B the same vector is (I + E)^-1 qacc, and E is a
  // contraction whatever the pose: its eigenvalues are those of h B^1/2 M^-1 B^1/2, bounded by h B_d / (armature_d + joint
  // inertia), 0.0393 for this model (knee spring dofs 6 and 11; tests/test_implicit_damping_bound.py samples poses through the oracle).
  // So the fixed-point iteration x <- qacc - E x from x = qacc converges to the solution with error 0.0393^n:
  // IMPLICIT_DAMPING_SWEEPS = 12 leaves 1e-17, below the rounding of any direct solve.  One iteration = ten v_fmac_f64_dpp
  // (the base dofs 0..2 are undamped, their columns of E are zero).
  {
    double me[10];
    static_for<0, 10>([&](auto jj) { constexpr int J = decltype(jj)::value; me[J] = sm.minv[mrow + 3 + J] * (H * cp_dof_damping[3 + J]); });
    double x = qacc;
#pragma unroll
    for (int it = 0; it < IMPLICIT_DAMPING_SWEEPS; it++) {
      double s0 = qacc, s1 = 0.0;
      damping_matvec(s0, s1, x, me);
      x = s0 + s1;
    }
    qacch = x;
  }
  PHASE_MARK(*pc, 8);
  lds_sync();
  if (c.dvalid && go) {
    sm.ws[c.d] = qacc;
    if (integrate) {
      double vn = sm.v[c.d] + H * qacch;
      sm.v[c.d] = vn;
      sm.q[c.d] = sm.q[c.d] + H * vn;
    }
  }
  lds_sync();
}

// ---------------------------------------------------------------- fused Env.step, 4 envs per wave
// MODE: 0 PD, 1 torque, 2 motor commands from the state record (the controller kernel of cassie_ctrl_g16.hip wrote them: StepOsc /
// StepJacobian = controller launch + this kernel with n_sub = 1).  pending[env] = substeps this kernel did NOT do (0 normally).
// Two roles: the first tier (p.pending == null: every environment, all n_sub substeps) and the hand-over tier behind the
// two-lanes-per-environment kernel (p.pending = that kernel's per-environment count of substeps left): an environment with a
// count c > 0 joins at substep n_sub - c -- so that all environments of a wave reach the end-of-step section together --
// and a wave whose four counts are zero returns at once.
template <int MODE, bool HF = false>
__global__ void __launch_bounds__(64, 2) env_step_g16_kernel(VecParams p, int* pending) {
  __shared__ EnvLds sm4[4];
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  const int env = blockIdx.x * 4 + g;
  bool valid = env < p.n_envs;
  int first = 0;   // substep at which this environment joins
  if (p.pending) {
    const int left = valid ? pending_count(p.pending[env], p.pending_pick) : 0;
    if (__ballot(left > 0) == 0) {
      if (valid && l == 0) pending[env] = 0;
      return;
    }
    if (valid && left == 0 && l == 0) pending[env] = 0;
    valid = valid && left > 0;
    first = p.n_sub - left;
  }
  EnvLds& sm = sm4[g];
  const size_t e = valid ? (size_t)env : 0;
  double* st = p.state + e * ENV_STRIDE;
  // The output section and the write-back recompute the environment index and the record pointer instead of keeping them
  // alive across the whole kernel: the allocator spilled exactly these kernel-lifetime values at the prologue (r02: the last
  // 32 B/lane of scratch, i.e. the last HBM traffic of this kernel that was not algorithmic).
  auto env_again = [&]() -> size_t {
    const int ev = (int)blockIdx.x * 4 + (opaque((int)threadIdx.x) >> 4);
    return ev < p.n_envs ? (size_t)ev : 0;
  };
  PhaseClock pc;   // profiling builds only (-DCASSIE_PHASE_TIMING, tools/phase_profile.py physics)
  pc.start();
  LaneConst c;
  load_lane_const(c, l);  // roles are per 16-lane row
  c.grp = 0; c.dvalid = l < NV;
  // ---- state load (strided inside the 704-byte record; the four records of a wave are adjacent)
  if (l < NV) {
    sm.q[l] = st[ES_Q + l]; sm.v[l] = st[ES_V + l]; sm.ws[l] = st[ES_WS + l];
    sm.qst[l] = st[ES_QSTATE + l];   // ES_KQ / ES_KV are not read: the first setState of this step overwrites them (set_state below)
  }
  if (l < NU) sm.ctrl[l] = st[ES_CTRL + l];
  if (l == 0) sm.tim[0] = st[ES_TIME];
  sm.actl[l] = (MODE != 2 && p.actions && c.act >= 0 && c.dvalid) ? p.actions[e * p.adim + c.act] : 0.0;
  lds_sync();
  bool live = valid;
  bool set_state = false;  // this environment did a setState in this launch: kq2 / kv2 are defined and go back to the record
  int pend = 0, niter_sum = (p.pending && valid) ? (int)st[ES_NITER] : 0;
  sm.ctl[l] = 0.0;
  G16Out so; so.niter = 0; so.overflow = false;
  const bool fix_kin = (p.flags & FLAG_FIX_STALE_KIN) != 0;
  // One loop, one copy of the substep code: passes 0..n_sub-1 are the physics substeps; the end-of-step section computes
  // observation / reward / termination; if any environment of the wave terminated, one more pass (mj_forward only, on the
  // reset pose, for those environments) produces the reset observation.
  bool do_reset = false, reset_pass = false;
  int sub = 0;
  if (p.pending) {   // start at the first substep any environment of the wave takes part in
    int fmin = valid ? first : p.n_sub;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) fmin = min(fmin, __shfl_xor(fmin, off));
    sub = fmin;
  }
  while (true) {
    const bool joined = sub >= first;
    const int dd = c.d < NV ? c.d : 0;
    const double q_d = sm.q[dd], v_d = sm.v[dd];
    double cnew;
    if (reset_pass || MODE == 2) cnew = c.act >= 0 ? sm.ctrl[c.act] : 0.0;  // Cassie2d::Reset: mj_forward with the stale ctrl
    else { const double act_l = sm.actl[l]; cnew = MODE == 0 ? 10.0 * (act_l - q_d) + 5.0 * (0.0 - v_d) : act_l; }
    substep<EnvLds, HF>(sm, c, l, g, cnew, reset_pass ? do_reset : (live && joined), !reset_pass, so, &p.hf, &pc);  // reset pose on the flat floor: 12 active rows
    if (!reset_pass) {
      if (live && joined && so.overflow) { live = false; pend = p.n_sub - sub; }  // hand the rest of this env to the clean-up pass
      if (live && joined) { sm.kq2[l] = q_d; sm.kv2[l] = v_d; sm.ctl[l] = cnew; niter_sum += so.niter; if (l == 0) sm.tim[0] += 0.0005; set_state = true; }  // setState of this substep
      sub++;
      if (sub < p.n_sub && __ballot(live) != 0) continue;
      if (live && c.dvalid && c.act >= 0) sm.ctrl[c.act] = sm.ctl[l];
      lds_sync();
    }
    if (!p.obs) break;
    // ---- end-of-step section: operational-space state from the kinematics of the last setState (quirks Q1/Q2)
    if (l < NV) { sm.kq[l] = sm.kq2[l]; sm.kv[l] = sm.kv2[l]; }
    lds_sync();
    opstate18(sm, c, l, fix_kin, sm.s18);
    double oa = sm.s18[l + 1 < 18 ? l + 1 : 17];      // obs[l] = s18[l+1]
    const double bodyx = sm.s18[0];
    if (l == 5 || l == 11) oa -= bodyx;
    double ob = l == 0 ? sm.s18[17] : 0.0;            // obs[16 + l]
    lds_sync();
    if (reset_pass) {
      if (do_reset) { const size_t e2 = env_again(); p.obs[e2 * 26 + l] = oa; if (l < 10) p.obs[e2 * 26 + 16 + l] = ob; }  // Cassie2dEnv.reset: 17 op-space values
      break;
    }
    double obs_a = oa, obs_b = ob, reward = 0.0;
    int done = 0;
    {
      const bool fixq = (p.flags & FLAG_FIX_STALE_QSTATE) != 0;
      const double qv = fixq ? sm.q[l < NV ? l : 0] : sm.qst[l];
      env_outputs_row(p, l, p.actions ? p.actions + env_again() * p.adim : nullptr, p.adim, sm.tim[0], bodyx, qv, obs_a, obs_b, reward, done);
    }
    // failure guard (see env_step_kernel): non-finite / diverged state => forced termination, reset clears every NaN carrier
    const bool badl = l < NV && !(in_range(sm.q[l < NV ? l : 0]) && in_range(sm.v[l < NV ? l : 0]));
    const bool bad = live && ((((unsigned)(__ballot(badl) >> (16 * g))) & 0xFFFFu) != 0 || !in_range(reward));
    if (bad) {
      obs_a = 0.0; obs_b = 0.0; reward = 0.0; done = 1;
      if (l == 0 && p.stats) atomicAdd(p.stats + STAT_NONFINITE, 1ull);
      if (p.auto_reset) {
        sm.ws[l] = 0.0; sm.kq2[l] = l < NV ? cp_env_qinit[l] : 0.0; sm.kv2[l] = 0.0; set_state = true;
        if (l < NU) sm.ctrl[l] = 0.0;
      }
    }
    {
      const size_t e2 = env_again();
      if (live && p.terminal_obs) { p.terminal_obs[e2 * 26 + l] = obs_a; if (l < 10) p.terminal_obs[e2 * 26 + 16 + l] = obs_b; }
      if (live) {
        p.obs[e2 * 26 + l] = obs_a;
        if (l < 10) p.obs[e2 * 26 + 16 + l] = obs_b;
        if (l == 0) { p.reward[e2] = reward; p.done[e2] = (uint8_t)done; }
      }
    }
    do_reset = live && done && p.auto_reset;
    if (__ballot(do_reset) == 0) break;
    // Cassie2dEnv.reset for the terminated environments: qinit, mj_forward with the stale ctrl, no setState
    if (do_reset && l < NV) { sm.q[l] = cp_env_qinit[l]; sm.v[l] = 0.0; sm.qst[l] = cp_env_qinit[l]; }
    if (do_reset && l == 0) { sm.tim[0] = 0.0; p.state[env_again() * ENV_STRIDE + ES_QPWSET] = 0.0; }  // new episode: cold start of the OSC QP too
    lds_sync();
    reset_pass = true;
  }
  // ---- state write-back
  if (valid) {
    const size_t e2 = env_again();
    double* const st2 = p.state + e2 * ENV_STRIDE;
    if (l < NV) {
      st2[ES_Q + l] = sm.q[l]; st2[ES_V + l] = sm.v[l]; st2[ES_WS + l] = sm.ws[l];
      if (set_state) { st2[ES_KQ + l] = sm.kq2[l]; st2[ES_KV + l] = sm.kv2[l]; }
      st2[ES_QSTATE + l] = sm.qst[l];
    }
    if (l < NU) st2[ES_CTRL + l] = sm.ctrl[l];
    if (l == 0) {
      st2[ES_TIME] = sm.tim[0]; st2[ES_NITER] = (double)niter_sum; pending[e2] = pend;
      if (pend > 0 && p.stats && !p.pending) atomicAdd(p.stats + STAT_CLEANUP_SUBSTEPS, (unsigned long long)pend);   // counted once, by the first tier
    }
  }
  PHASE_MARK(pc, 0);
  pc.flush(p.phase, (int)threadIdx.x);
}

// ---------------------------------------------------------------- routing of handed-down environments
// For every environment the two-lanes-per-environment kernel left pending: would THIS kernel's substep find more than MAXR rows at
// the substep the environment is stopped at (same kinematics, same activity tests, same row count as `substep` above)?  Then
// pending[env] gets PENDING_DEEP and the wave-per-environment kernel takes it directly, at the same time as this kernel works on
// the others (launch_physics_tiers, cassie_cabi.hip).  Only a routing hint -- an environment that grows past MAXR rows later in the
// step is passed on as before.  A separate tiny kernel because the two-lanes-per-environment kernel is not to be touched for this:
// counting there cost 0.8-2.6 % of the headline launch in three formulations (register allocation of its solver loop).
#ifdef CASSIE_TU_G16   // one definition: this file is also included by the terrain translation unit
__global__ void __launch_bounds__(64, 2) classify_pending_kernel(VecParams p, int* pending) {
  __shared__ EnvLds sm4[4];
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  const int env = blockIdx.x * 4 + g;
  const bool valid = env < p.n_envs;
  const int left = valid ? (pending[env] & PENDING_COUNT) : 0;
  const unsigned long long some = __ballot(left > 0 && l == 0);
  // how many left the first tier, for the host's choice of schedule: an ESTIMATE from every 64th workgroup, summed in DEVICE memory
  // (agent-scope atomics) and handed to the host by ONE plain store of the launch's total -- the last sampling workgroup to arrive writes
  // the pinned word (r06; until r05 every sampling workgroup did a system-scope atomic add on the pinned word itself, which needs PCIe
  // atomics: where the platform lacks them the hint stayed 0 and the segmented schedule was never chosen, silently).
  if (lane == 0 && p.pend_hint && p.pend_count && (blockIdx.x & 63) == 0) {
    // relaxed atomics (an acquire / release at agent scope writes back and invalidates the XCD's whole L2: measured, 8 -> 40 us for this kernel);
    // the ticket is taken only after the sum's atomic has RETURNED (its result feeds the ticket's operand), so the last arrival reads a complete sum
    unsigned seen = 0u;
    if (some) seen = __hip_atomic_fetch_add(p.pend_count, 64u * (unsigned)__popcll(some), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned samplers = (gridDim.x + 63u) >> 6;
    if (__hip_atomic_fetch_add(p.pend_count + 1, 1u + (seen & 0x80000000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == samplers - 1u) {
      const unsigned total = __hip_atomic_exchange(p.pend_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.pend_count + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // an Env.step in segments classifies once per segment under ONE serial: the word carries the step's running sum (kept on the device:
      // [2 + slot] the sum, [66 + slot] the serial it belongs to; only this one thread of this one launch touches them)
      const unsigned slot = (unsigned)p.serial & 63u;
      if (p.pend_count[66 + slot] != (unsigned)p.serial) { p.pend_count[66 + slot] = (unsigned)p.serial; p.pend_count[2 + slot] = 0u; }
      const unsigned sum = p.pend_count[2 + slot] + total;
      p.pend_count[2 + slot] = sum;
      *(volatile unsigned*)(p.pend_hint + slot) = sum;   // plain store to host memory
    }
  }
  if (some == 0) return;
  EnvLds& sm = sm4[g];
  const double* st = p.state + (valid ? (size_t)env : 0) * ENV_STRIDE;
  LaneConst c;
  load_lane_const(c, l);
  c.grp = 0; c.dvalid = l < NV;
  if (l < NV) { sm.q[l] = st[ES_Q + l]; sm.v[l] = 0.0; }
  lds_sync();
  planar_fk<0>(sm, sm.q, sm.v, c, l);
  const double basez = sm.q[1] - cp_qpos0[1] + cp_link_off[0][0][1];
  bool lim_act = false;
  if (l < 8) {
    const int dof = cp_slot_dof[SLOT_LIM + l];
    const double qd = sm.q[dof];
    lim_act = (qd - cp_jnt_range[dof][0] < 0) || (cp_jnt_range[dof][1] - qd < 0);
  }
  auto sphere_active = [&](int sph) {
    double cx, cz;
    link_point(sm, cp_sph_link[sph], cp_sph_d[sph][0], cp_sph_d[sph][1], cx, cz);
    return basez + cz - cp_sph_r[sph] < 0;
  };
  const bool con_act0 = sphere_active(l);
  const bool con_act1 = (l == 0) ? sphere_active(16) : false;
  const unsigned long long bl = __ballot(lim_act), b0 = __ballot(con_act0), b1 = __ballot(con_act1);
  const int nlim = __popc((unsigned)(bl >> (16 * g)) & 0xFFu);
  const int ncon = __popc(((unsigned)(b0 >> (16 * g)) & 0xFFFFu) | ((((unsigned)(b1 >> (16 * g))) & 1u) << 16));
  const int cbase = (nlim >= 1 && nlim <= 4 && ncon <= 4) ? 8 : ((4 + nlim + 1) & ~1);
  if (valid && left > 0 && l == 0 && cbase + 2 * ncon > MAXR) {
    pending[env] = left | PENDING_DEEP;
    if (p.deep_hint) *p.deep_hint = p.serial;
  }
}
#endif

}  // namespace g16
}  // namespace cassie
#endif
