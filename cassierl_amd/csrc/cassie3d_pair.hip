// cassie3d_pair.hip -- batched Cassie3d physics with TWO ENVIRONMENTS PER WAVEFRONT (one per 32-lane half): the <= 30-row fast kernel
// of BASELINE.json configs[4].  Included by tu_3d.hip after cassie3d_kernels.hip, whose kinematics / mass-matrix phase it shares
// (`kin_mass3`, called with the lane's index INSIDE its half and the half's own LDS block).
//
// Why: in the one-environment-per-wavefront kernel every phase uses at most 32 lanes (15 links, 20 dofs, <= 32 rows) and the
// contact step of the solver -- a 3x3 block solve on wave-uniform values, ~130 instructions, most of the kernel -- runs
// redundantly on all 64 lanes for ONE environment (r03_j PMC: 255 k VALU instructions per wavefront per step, one per 7.5 cycles per
// SIMD).  Here the same instruction stream serves two environments.
//
// What makes that work:
//   * half-wide broadcasts: `hbc(x, K)` = lane K of the caller's own half (two v_readlane per word + a select; K may be a
//     run-time wave-uniform value, so the rolled loops of the solver stay rolled);
//   * per-environment control values (row counts, activity masks, kinds) are per-lane integers that agree inside a half; loops
//     run to the larger of the two counts with per-lane predicates, so the wavefront never diverges around a cross-lane read;
//   * ALIGNED ROWS: 6 connect rows, then the active limits PADDED to a multiple of three, then 3 rows per contact -- every
//     contact block of every environment starts at a row K = 0 mod 3, so the solver walks TRIPLES: when both environments have
//     the same kind of triple (always, for robots on their feet; nearly always otherwise) one pass serves both.  Pad rows are
//     inactive rows: zero force, zero columns of A -- an environment's arithmetic is what it is without them (they add exact
//     zeros), so results do not depend on the neighbour in the other half;
//   * an environment that needs more than MAXR_PAIR rows is left untouched from that substep on and handed to the general kernel
//     through `pending_out`, as before; its neighbour carries on.
#ifndef CASSIE3D_PAIR_HIP_
#define CASSIE3D_PAIR_HIP_

namespace cassie3d {

constexpr int MAXR_PAIR = 30;   // rows per environment (10 triples) inside the 32 lanes of a half

// lane K (0..31, wave-uniform, static or dynamic) of the caller's own half
__device__ __forceinline__ double hbc(double x, int K, bool upper) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int lo0 = __builtin_amdgcn_readlane(lo, K), hi0 = __builtin_amdgcn_readlane(hi, K);
  const int lo1 = __builtin_amdgcn_readlane(lo, K + 32), hi1 = __builtin_amdgcn_readlane(hi, K + 32);
  return __hiloint2double(upper ? hi1 : hi0, upper ? lo1 : lo0);
}
__device__ __forceinline__ int hbci(int x, int K, bool upper) {
  const int a = __builtin_amdgcn_readlane(x, K), b = __builtin_amdgcn_readlane(x, K + 32);
  return upper ? b : a;
}
// sum over the 32 lanes of a half (every lane of the half gets it)
__device__ __forceinline__ double hsum(double x) {
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) x += __shfl_xor(x, off);
  return x;
}
// the larger of a per-environment integer over the two halves (wave-uniform)
__device__ __forceinline__ int pair_max(int x) {
  const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 32);
  return a > b ? a : b;
}

struct Out3p { int niter, nefc; bool overflow; };   // per lane, equal inside a half

// In-register Gauss-Jordan inverse, row d of each half's matrix on lane d < NV of that half
__device__ __forceinline__ void gauss_jordan20_pair(double (&Mr)[NV], int hl, bool upper) {
  static_for<0, NV>([&](auto kk) {
    constexpr int K = decltype(kk)::value;
    const double piv = hbc(Mr[K], K, upper);
    const double inv = cassie::fast_rcp(piv);
    const bool isk = hl == K;
    const double t = isk ? 1.0 - inv : Mr[K] * inv;
    static_for<0, NV>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      if constexpr (C != K) {
        const double pk = hbc(Mr[C], K, upper);
        Mr[C] = __builtin_fma(-t, pk, Mr[C]);
      }
    });
    Mr[K] = isk ? inv : -t;
  });
}

// ---------------------------------------------------------------- one mj_forward (+ Euler step) of the two environments of a wave
// `live`: this half's environment takes part (valid, not handed over).  Nothing of a non-live environment's state is written.
__device__ void substep3_pair(Smem3<32>& sm, int lane, double ctrl_l, bool integrate, bool live, Out3p& out) {
  constexpr int MR = 32;
  const bool upper = lane >= 32;
  const int hl = lane & 31;
  double bias;
  kin_mass3(sm, hl, bias, nullptr);
  const int d = hl < NV ? hl : 0;
  const bool dvalid = hl < NV;
  const double damping = c3_dof_damping[d];
  double Mr[NV];
  static_for<0, NV>([&](auto jj) { constexpr int J = decltype(jj)::value; Mr[J] = dvalid ? sm.minv[d][J] : 0.0; });
  if (!live) {  // keep the arithmetic of an idle half finite: identity instead of whatever its LDS block holds
    static_for<0, NV>([&](auto jj) { constexpr int J = decltype(jj)::value; Mr[J] = (dvalid && J == d) ? 1.0 : 0.0; });
  }
  gauss_jordan20_pair(Mr, hl, upper);
  // ================= smooth acceleration
  double tau;
  {
    const int a = c3_dof_act[d];
    double u = 0.0;
    if (a >= 0) { u = ctrl_l; const double lo = c3_act_ctrlrange[a][0], hi = c3_act_ctrlrange[a][1]; u = u < lo ? lo : (u > hi ? hi : u); u *= c3_act_gear[a]; }
    tau = (dvalid && live) ? -damping * sm.v[d] - bias + u : 0.0;
  }
  double qs = 0.0;
  static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; qs += Mr[C] * hbc(tau, C, upper); });
  lds_sync();
  if (dvalid) {
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; sm.minv[d][C] = Mr[C]; });
    sm.qs[d] = qs;
  }
  // ================= collision: sphere s on lane s of the half (capsule ends are spheres for a plane, mjc_PlaneCapsule)
  bool con_act = false;
  if (hl < NSPH) {
    const int l = c3_sph_link[hl];
    double r[3], hw[3];
    matvec3(sm.xmat[l], c3_sph_pos[hl], r);
    const double cx = sm.xpos[l][0] + r[0], cy = sm.xpos[l][1] + r[1], cz = sm.xpos[l][2] + r[2];
    const double dist = cz - c3_sph_radius[hl];
    con_act = live && dist < 0;
    sm.sphc[hl][0] = cx; sm.sphc[hl][1] = cy; sm.sphc[hl][2] = cz - c3_sph_radius[hl] - 0.5 * dist;  // contact point
    sm.sphdist[hl] = dist;
    // mju_makeFrame with normal +z: first tangent = hint minus its normal part (spheres: world y), second = n x t1
    matvec3(sm.xmat[l], c3_sph_hint[hl], hw);
    const bool has_hint = c3_sph_hint[hl][0] != 0.0 || c3_sph_hint[hl][1] != 0.0 || c3_sph_hint[hl][2] != 0.0;
    double tx = has_hint ? hw[0] : 0.0, ty = has_hint ? hw[1] : 1.0;
    const double n = sqrt(tx * tx + ty * ty);
    if (n < MINVAL) { tx = 1.0; ty = 0.0; } else { tx /= n; ty /= n; }
    sm.spht1[hl][0] = tx; sm.spht1[hl][1] = ty;
  }
  bool lim_act = false;
  double lim_dist = 0.0, lim_sgn = 0.0;
  if (hl < NLIM) {
    const int dof = c3_lim_dof[hl];
    const double qd = sm.q[c3_dof_qadr[dof]];
    const double dlo = qd - c3_lim_range[hl][0], dhi = c3_lim_range[hl][1] - qd;
    if (dlo < 0) { lim_act = true; lim_dist = dlo; lim_sgn = 1.0; }
    else if (dhi < 0) { lim_act = true; lim_dist = dhi; lim_sgn = -1.0; }
    lim_act = lim_act && live;
  }
  const unsigned long long bc_ = __ballot(con_act), bl_ = __ballot(lim_act);
  const unsigned con_mask = (unsigned)(upper ? (bc_ >> 32) : bc_), lim_mask = (unsigned)(upper ? (bl_ >> 32) : bl_);
  const int ncon = __popc(con_mask), nlim = __popc(lim_mask);
  // row layout: 6 connect rows | limits, padded to a multiple of 3 | 3 rows per contact
  const int cstart = 3 * NEQ + 3 * ((nlim + 2) / 3);
  const int nrows = cstart + 3 * ncon;            // padded row count (a multiple of 3)
  out.nefc = 3 * NEQ + nlim + 3 * ncon;           // what MuJoCo would count
  out.overflow = live && nrows > MAXR_PAIR;
  const bool go = live && !out.overflow;          // this half's environment is solved in this substep
  lds_sync();
  // ================= the row owned by this lane: up to two (link, point, sign) point-Jacobian terms along `dir`
  int kind = K_NONE, cbase = hl;
  double pos = 0.0, invw = 0.0;
  const double* solref = c3_contact_solref;
  const double* solimp = c3_contact_solimp;
  int mask1 = 0, mask2 = 0, limdof = -1;
  double p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0}, dir[3] = {0, 0, 0};
  if (go && hl < 3 * NEQ) {
    kind = K_EQ;
    const int e = hl / 3, comp = hl % 3;
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
  } else if (go && hl < 3 * NEQ + nlim) {
    kind = K_LIM;
    cbase = nth_set(lim_mask, hl - 3 * NEQ);  // the lane (of this half) that tested this limit (shuffle source below)
    limdof = c3_lim_dof[cbase];
    invw = c3_dof_invweight[limdof];
    solref = c3_limit_solref; solimp = c3_limit_solimp;
  } else if (go && hl >= cstart && hl < nrows) {
    const int k = (hl - cstart) / 3, comp = (hl - cstart) % 3;
    kind = comp == 0 ? K_CN : K_CT;
    cbase = hl - comp;
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
    const int src = (kind == K_LIM ? cbase : 0) + (lane & 32);
    const double ld = __shfl(lim_dist, src), ls = __shfl(lim_sgn, src);
    if (kind == K_LIM) { pos = ld; lim_s = ls; cbase = hl; }
  }
  const bool active = kind != K_NONE;
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
    val = active ? val : 0.0;   // pad rows, rows of an idle half: exact zeros (their masks are 0 anyway; this also kills a NaN of an idle half)
    sm.rowJ[hl][j] = val;
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
  R = __shfl(R, cbase + (lane & 32));  // friction rows share the normal row's regulariser (impratio 1, isotropic friction)
  const double b = active ? bq - aref : 0.0;
  const double jar = active ? jw - aref : 0.0;
  // X = J M^-1 (M^-1 is symmetric: its row j is read as a contiguous broadcast inside the half)
  double X[NV];
  static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; X[C] = 0.0; });
#pragma unroll 1
  for (int j = 0; j < NV; j++) {
    const double Jj = sm.rowJ[hl][j];
    static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; X[C] += sm.minv[j][C] * Jj; });
  }
  if (!active) { static_for<0, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; X[C] = 0.0; }); }
  lds_sync();
  const int nrows_w = pair_max(go ? nrows : 0);   // wave-uniform loop bound
  double Adiag = 1.0;
  for (int c = 0; c < nrows_w; c++) {
    double a = 0.0;
    static_for<0, NV>([&](auto jj) { constexpr int Jx = decltype(jj)::value; a += X[Jx] * sm.rowJ[c < 32 ? c : 0][Jx]; });
    if (c == hl && active) { a += R; Adiag = a; }
    if (c <= hl) sm.A[Smem3<32>::tri(c) + hl] = (active && go && c < nrows) ? a : 0.0;   // upper triangle; pad rows / an idle half: exact zeros
  }
  lds_sync();
  const double Ainv = 1.0 / Adiag;
  // ================= warm start (mj_constraintUpdate on qacc_warmstart), kept only if its dual cost beats zero force
  const double mu = MU;
  double f = 0.0;
  {
    const double D = 1.0 / R;
    const int hb = lane & 32;
    const double jn = __shfl(jar, hb + cbase), j1 = __shfl(jar, hb + (cbase + 1 < 32 ? cbase + 1 : 31)), j2 = __shfl(jar, hb + (cbase + 2 < 32 ? cbase + 2 : 31));
    if (kind == K_EQ) f = -D * jar;
    else if (kind == K_LIM) f = jar < 0 ? -D * jar : 0.0;
    else if (kind == K_CN || kind == K_CT) {
      const int comp = hl - cbase;
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
  const int trl = Smem3<32>::tri(hl);
  for (int c = 0; c < nrows_w; c++) res += sm.a_at(c, hl, trl) * hbc(f, c < 32 ? c : 0, upper);
  {
    const double cost = hsum(active ? f * (0.5 * res + b) : 0.0);
    if (cost > 0) { f = 0.0; res = 0.0; }
  }
  res += b;
  // ================= PGS (mj_solPGS, elliptic cones), triple by triple
  const double scale = 1.0 / (MEANINERTIA * NV);
  int niter = 0;
  const bool isLim = kind == K_LIM;
  const double hAdiag = 0.5 * Adiag;
  bool sweeping = go && nrows > 0;     // this half's environment still iterates (equal inside a half)
  for (int iter = 0; iter < ITERATIONS; iter++) {
    if (__ballot(sweeping) == 0) break;
    double improvement = 0.0;  // contact rows: uniform inside a half; single rows: accumulated on the owner lane, reduced once per sweep
    double acc = 0.0;
    for (int K = 0; K < nrows_w; K += 3) {
      const int kindK = hbci(kind, K, upper);
      const bool tripC = sweeping && kindK == K_CN;                     // this half: a contact block at rows K..K+2
      const bool tripS = sweeping && kindK != K_CN && K < nrows;        // this half: three single rows (connect / limit / pad)
      if (__ballot(tripS) != 0) {
#pragma unroll 1
        for (int j = 0; j < 3; j++) {
          // every lane evaluates the single-row update of ITS OWN row from its own registers; only lane K+j's result is used
          double nf = f - res * Ainv;
          nf = isLim ? fmax(nf, 0.0) : nf;
          double dOwn = nf - f;
          const double chg = dOwn * (hAdiag * dOwn + res);
          const bool keep = tripS && active && chg <= 1e-10;
          dOwn = keep ? dOwn : 0.0;
          acc += (keep && hl == K + j) ? chg : 0.0;
          const double dK = hbc(dOwn, K + j, upper);
          res += sm.a_at(K + j, hl, trl) * dK;
          if (hl == K + j) f += dK;
        }
      }
      if (__ballot(tripC) != 0) {  // contact: rows K (normal), K+1, K+2 (tangents)
        const double o0 = hbc(f, K, upper), o1 = hbc(f, K + 1, upper), o2 = hbc(f, K + 2, upper);
        const double r0 = hbc(res, K, upper), r1 = hbc(res, K + 1, upper), r2 = hbc(res, K + 2, upper);
        // symmetric 3x3 diagonal block of A (the mirrored entries agree to rounding; one of each pair is read)
        const int t0 = Smem3<32>::tri(K), t1 = Smem3<32>::tri(K + 1), t2 = Smem3<32>::tri(K + 2);
        const double A00 = sm.A[t0 + K], A01 = sm.A[t0 + K + 1], A02 = sm.A[t0 + K + 2];
        const double A11 = sm.A[t1 + K + 1], A12 = sm.A[t1 + K + 2], A22 = sm.A[t2 + K + 2];
        // normal-only update (taken when the normal force is ~0)
        const double fn_n = fmax(o0 - r0 * hbc(Ainv, K, upper), 0.0);
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
          bool newton = tripC && !degenerate && val0 >= 1e-10 && f0 >= MINVAL;   // sliding contact: this half keeps iterating
          if (__ballot(newton) != 0) {
            double val = val0, P11 = Q22 * di0, P22 = Q11 * di0, P12 = -Q12 * di0;
            for (int it = 0; it < 20; it++) {
              // mju_QCQP2's loop with its exits as a per-half flag: a half that has left keeps its values
              const double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
              const double delta = -val * fast_rcp(deriv);
              if (delta < 1e-10) newton = false;
              if (newton) la += delta;
              if (it == 19) newton = false;  // iteration budget of mju_QCQP2: the last multiplier is kept, v is not recomputed
              const double det = (Q11 + la) * (Q22 + la) - Q12 * Q12;
              if (newton && det < 1e-10) { degenerate = true; newton = false; }
              if (newton) {
                const double di = fast_rcp(det);
                P11 = (Q22 + la) * di; P22 = (Q11 + la) * di; P12 = -Q12 * di;
                v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
                val = v1 * v1 + v2 * v2 - f0 * f0;
                if (val < 1e-10) newton = false;
              }
              if (__ballot(newton) == 0) break;
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
        const bool keep = tripC && chg <= 1e-10;   // (false for a half that is not at a contact block: NaN-safe, chg may be anything there)
        d0 = keep ? d0 : 0.0; d1 = keep ? d1 : 0.0; d2 = keep ? d2 : 0.0;
        improvement -= keep ? chg : 0.0;
        res += sm.a_at(K, hl, trl) * d0 + sm.a_at(K + 1, hl, trl) * d1 + sm.a_at(K + 2, hl, trl) * d2;
        if (hl == K) f += d0;
        if (hl == K + 1) f += d1;
        if (hl == K + 2) f += d2;
      }
    }
    improvement -= hsum(acc);
    if (sweeping) {
      niter = iter + 1;
      if (improvement * scale < TOLERANCE) sweeping = false;
    }
  }
  out.niter = (go && nrows > 0) ? niter : 0;
  // ================= total force g = tau + J' f;  qacc = M^-1 g;  Euler with implicit joint damping: (M + h B)^-1 g
  double g = dvalid ? tau : 0.0;
  for (int r = 0; r < nrows_w; r++) {
    const double fr = hbc(f, r < 32 ? r : 0, upper);
    g += dvalid ? sm.rowJ[r < 32 ? r : 0][d] * fr : 0.0;
  }
  double qacc = 0.0, qacch = 0.0;
  {
    double a0 = 0.0, a1 = 0.0;
    static_for<0, NV>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      const double t = sm.minv[d][C] * hbc(g, C, upper);
      if constexpr (C & 1) a1 += t; else a0 += t;
    });
    qacc = a0 + a1;
  }
  // implicit joint damping by the fixed-point iteration of the one-environment kernel (cassie3d_kernels.hip)
  {
    double me[NV - 6];
    static_for<6, NV>([&](auto cc) { constexpr int C = decltype(cc)::value; me[C - 6] = sm.minv[d][C] * (H * c3_dof_damping[C]); });
    double x = qacc;
#pragma unroll
    for (int it = 0; it < IMPLICIT_DAMPING_SWEEPS3; it++) {
      double s0 = qacc, s1 = 0.0;
      static_for<6, NV>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        const double xc = hbc(x, C, upper);
        if constexpr (C & 1) s1 = __builtin_fma(-me[C - 6], xc, s1); else s0 = __builtin_fma(-me[C - 6], xc, s0);
      });
      x = s0 + s1;
    }
    qacch = x;
  }
  lds_sync();
  if (dvalid && go) {
    sm.ws[d] = qacc;
    if (integrate) sm.v[d] = sm.v[d] + H * qacch;
  }
  lds_sync();
  if (integrate && go) {
    if (hl < 3) sm.q[hl] += H * sm.v[hl];
    if (hl >= 6 && hl < NV) sm.q[hl + 1] += H * sm.v[hl];
    if (hl == 3) {
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
  }
  lds_sync();
}

// ---------------------------------------------------------------- n_sub torque-mode substeps, two environments per wavefront
// First pass of Cassie3dVecStep: every environment that needs at most MAXR_PAIR (padded) rows; the others are handed to
// env_step3d_kernel<MAXR, 1> through `pending_out` (substeps left, state saved at that point).
__global__ void __launch_bounds__(64, 2) env_step3d_pair_kernel(Params3 p) {
  __shared__ Smem3<32> sm2[2];
  const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
  const int env = blockIdx.x * 2 + h;
  const bool valid = env < p.n_envs;
  Smem3<32>& sm = sm2[h];
  const size_t e = valid ? (size_t)env : 0;
  double* st = p.state + e * ENV3_STRIDE;
  if (hl < NQ) sm.q[hl] = st[E3_Q + hl];
  if (hl < NV) { sm.v[hl] = st[E3_V + hl]; sm.ws[hl] = st[E3_WS + hl]; }
  double time = st[E3_TIME];
  const int a = hl < NV ? c3_dof_act[hl] : -1;
  double ctrl_l = 0.0;
  if (a >= 0) ctrl_l = p.actions ? p.actions[e * NU + a] : st[E3_CTRL + a];
  lds_sync();
  Out3p out; out.niter = 0; out.nefc = 0; out.overflow = false;
  int niter_sum = 0, left = 0;
  bool live = valid;
  for (int sub = 0; sub < p.n_sub; sub++) {
    substep3_pair(sm, lane, ctrl_l, p.integrate != 0, live, out);
    if (live && out.overflow) { live = false; left = p.n_sub - sub; }   // detected before anything of this substep was written
    if (live) { niter_sum += out.niter; if (p.integrate) time += H; }
    if (__ballot(live) == 0) break;
  }
  if (valid) {
    if (hl < NQ) st[E3_Q + hl] = sm.q[hl];
    if (hl < NV) { st[E3_V + hl] = sm.v[hl]; st[E3_WS + hl] = sm.ws[hl]; }
    if (a >= 0) st[E3_CTRL + a] = ctrl_l;
    if (hl == 0) {
      st[E3_TIME] = time; st[E3_NITER] = (double)niter_sum;
      if (left == 0) st[E3_NEFC] = (double)out.nefc;
      if (p.pending_out) p.pending_out[env] = left;
    }
  }
}

}  // namespace cassie3d
#endif
