/* cassie_oracle.c -- CPU ORACLE (test infrastructure only; see cassie_oracle.h for the
 * "parity unpinned" statement and the list of reference call sites this restates).
 *
 * Part 1: rigid-body kinematics/dynamics in 3-D world coordinates (what RBDL's
 *         UpdateKinematics / CRBA / NonlinearEffects / CalcPoint* and MuJoCo's
 *         mj_kinematics / mj_crb / mj_rne compute).
 * Part 2: MuJoCo-semantics forward dynamics + PGS/elliptic solve + Euler step
 *         (mj_step as configured by model/cassie2d_stiff.xml:5,16,178-189).
 * Part 3: DynamicState + the four Cassie2d::Step* variants and the two state getters.
 * Part 4: OSC QP (OSC_RBDL.cpp:114-291) solved to KKT convergence.
 * Part 5: environment layer (rllab/envs/cassie2d.py, cassie_stand2d.py).
 */
#include "cassie_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* -DORC_CASSIE3D builds the same Part 1/2 pipeline for model/cassie3d_stiff.xml (floating base: 3 world-frame
 * translations + unit quaternion with body-frame angular velocity, 14 hinges).  Parts 3-5
 * (DynamicState, controllers, environments) exist for Cassie2d only, as in the reference. */
#ifdef ORC_CASSIE3D
#include "cassie3d_model.h"
#define NQ CM_NQ
#define QADR(j) (cm_jnt_qadr[j])
#define QPOS0 cm_qpos0
#else
#include "cassie2d_model.h"
#define NQ CM_NV
#define QADR(j) (j)
#define QPOS0 cm_jnt_ref
#endif

#define NV CM_NV
#define NB CM_NBODY
#define NU CM_NU
#define NEQ CM_NEQ
#define MINVAL 1e-15

enum { CT_EQUALITY = 0, CT_LIMIT = 1, CT_CONTACT = 2 };

/* ------------------------------------------------------------------ small linear algebra */
static void v3set(double* a, double x, double y, double z) { a[0] = x; a[1] = y; a[2] = z; }
static void v3cpy(double* a, const double* b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
static void v3add(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static void v3sub(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void v3addscl(double* r, const double* a, const double* b, double s) {
  r[0] = a[0] + s * b[0]; r[1] = a[1] + s * b[1]; r[2] = a[2] + s * b[2];
}
static double v3dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void v3cross(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double v3normalize(double* a) {
  double n = sqrt(v3dot(a, a));
  if (n < MINVAL) { v3set(a, 1, 0, 0); return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
static void m3mulv(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void m3tmulv(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  double y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  double z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void m3mul(double* r, const double* a, const double* b) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof t);
}
static void m3rot(double* r, const double* axis, double ang) { /* Rodrigues */
  double c = cos(ang), s = sin(ang), t = 1 - c, x = axis[0], y = axis[1], z = axis[2];
  r[0] = c + t * x * x;     r[1] = t * x * y - s * z; r[2] = t * x * z + s * y;
  r[3] = t * x * y + s * z; r[4] = c + t * y * y;     r[5] = t * y * z - s * x;
  r[6] = t * x * z - s * y; r[7] = t * y * z + s * x; r[8] = c + t * z * z;
}

/* dense Cholesky A = L L^T (lower), n <= NV;  returns 0 on success */
static int chol_factor(int n, const double* A, double* L) {
  memset(L, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
    if (d <= 0) return 1;
    d = sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  return 0;
}
static void chol_solve(int n, const double* L, const double* b, double* x) {
  double y[64];
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
    y[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

/* ------------------------------------------------------------------ data */
typedef struct {
  double xpos[NB][3], xmat[NB][9], xipos[NB][3];
  double anchor[NV][3], axis[NV][3];
  double w[NB][3], vo[NB][3], al[NB][3], ao[NB][3]; /* angular vel, origin vel, velocity-product ang/lin accel */
  int has_vel;
} Kin;

typedef struct {
  double dist, pos[3], frame[9];
  int body, geom;
} Contact;

typedef struct { /* DynamicMatrices (DynamicState.h:14-21), stiff config == spring config for this model */
  double M[NV * NV], bias[NV], Bt[NV * NU], Jc[12 * NV], Jeq[6 * NV], JeqdotQdot[6];
} DynState;

struct Oracle {
  /* test knobs */
  double gravity_z, damping[NV];
  int contact_enabled;
  /* height-field terrain (rllab/envs/terrain_random.py): heights in metres, row r <-> y, column c <-> x; NULL = flat */
  const double* hf;
  int hf_nrow, hf_ncol;
  double hf_sx, hf_sy;
  int row_cap; /* > 0: at most this many constraint rows; the last contacts (MuJoCo order) that do not fit are dropped (Cassie3d kernel's cap) */
  int assume; /* ORC_ASSUME_* bits: the places where closed MuJoCo Pro 1.50 may differ from the published 2.x pipeline */
  /* constants derived at create (what MuJoCo's compiler / LoadModel derive) */
  double eq_anchor2[2][NEQ][3];
  double dof_invweight0[NV], body_invweight0[NB], meaninertia;
  unsigned char affects[NB][NV]; /* dof k moves body b */
  /* mjData */
  double qpos[NQ], qvel[NV], qacc[NV], qacc_ws[NV], ctrl[NU];
  Kin kin;
  double M[NV * NV], L[NV * NV], bias[NV], passive[NV], actuator[NV], qfrc_smooth[NV], qacc_smooth[NV];
  int ncon;
  Contact con[ORC_MAXCON];
  int nefc;
  int efc_type[ORC_MAXEFC], efc_id[ORC_MAXEFC];
  double efc_J[ORC_MAXEFC * NV], efc_pos[ORC_MAXEFC], efc_margin[ORC_MAXEFC], efc_vel[ORC_MAXEFC];
  double efc_diagApprox[ORC_MAXEFC], efc_R[ORC_MAXEFC], efc_D[ORC_MAXEFC], efc_aref[ORC_MAXEFC];
  double efc_b[ORC_MAXEFC], efc_force[ORC_MAXEFC];
  double AR[ORC_MAXEFC * ORC_MAXEFC];
  int solver_niter;
  /* DynamicModel state_ (last setState) and DynamicState */
  double kin_qpos[NQ], kin_qvel[NV];
  DynState ds;
  /* last OSC QP */
  double qp_x[39], qp_kkt[4];
  double qp_prev_x[39];
  int qp_have_prev;
};

/* ------------------------------------------------------------------ Part 1: kinematics & dynamics terms */
static void kinematics(const Oracle* o, int sem, const double* q, const double* v, Kin* k) {
  (void)o;
  memset(k, 0, sizeof *k);
  k->has_vel = v != NULL;
  for (int i = 0; i < 3; i++) k->xmat[0][4 * i] = 1.0;
  for (int b = 1; b < NB; b++) {
    int p = cm_body_parent[b];
    double pos[3], mat[9], r[3], w[3], al[3], vo[3], ao[3], t[3], t2[3];
    m3mulv(r, k->xmat[p], cm_body_pos[b]);
    v3add(pos, k->xpos[p], r);
    m3mul(mat, k->xmat[p], &cm_body_rot[sem][b][0][0]);
    v3cpy(w, k->w[p]); v3cpy(al, k->al[p]);
    v3cross(t, w, r); v3add(vo, k->vo[p], t);
    v3cross(t2, w, t); v3cross(t, al, r);
    v3add(ao, k->ao[p], t); v3add(ao, ao, t2);
    for (int j = 0; j < NV; j++) {
      if (cm_jnt_body[j] != b) continue;
      if (cm_jnt_type[j] == 2) { /* rotational half of a free joint: quaternion at q[QADR], body-frame angular velocity */
        if (cm_jnt_axis[j][0] != 1.0) continue; /* the y and z dofs are handled together with the x dof */
        const double* qq = q + QADR(j);
        double n = sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
        double qw = qq[0] / n, qx = qq[1] / n, qy = qq[2] / n, qz = qq[3] / n;
        double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                       2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                       2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
        double m2[9], wl[3] = {v ? v[j] : 0.0, v ? v[j + 1] : 0.0, v ? v[j + 2] : 0.0}, ww[3];
        m3mul(m2, mat, R); memcpy(mat, m2, sizeof m2);
        for (int c = 0; c < 3; c++) {
          v3set(k->axis[j + c], mat[c], mat[3 + c], mat[6 + c]);
          v3cpy(k->anchor[j + c], pos);
        }
        m3mulv(ww, mat, wl);                 /* the three dofs act together: w += R omega, and the velocity-product */
        v3cross(t, w, ww); v3add(al, al, t); /* angular acceleration is w_parent x (R omega) (mj_comVel, ball/free) */
        v3add(w, w, ww);
        continue;
      }
      double ax[3], qd = v ? v[j] : 0.0, d = q[QADR(j)] - cm_jnt_ref[j];
      m3mulv(ax, mat, cm_jnt_axis[j]);
      v3cpy(k->axis[j], ax);
      if (cm_jnt_type[j] == 0) { /* slide */
        double disp[3] = {ax[0] * d, ax[1] * d, ax[2] * d}, axqd[3] = {ax[0] * qd, ax[1] * qd, ax[2] * qd};
        v3cross(t, w, disp);                /* w x disp */
        v3cross(t2, w, t);                  /* w x (w x disp) */
        v3add(vo, vo, t); v3add(vo, vo, axqd);
        v3add(ao, ao, t2);
        v3cross(t, al, disp); v3add(ao, ao, t);
        v3cross(t, w, axqd); v3addscl(ao, ao, t, 2.0);
        v3add(pos, pos, disp);
        v3cpy(k->anchor[j], pos);
      } else { /* hinge about the body origin (every joint pos is 0 0 0 in this model) */
        double R[9], axqd[3] = {ax[0] * qd, ax[1] * qd, ax[2] * qd};
        v3cpy(k->anchor[j], pos);
        m3rot(R, ax, d);
        m3mul(mat, R, mat);
        v3cross(t, w, axqd); v3add(al, al, t);
        v3add(w, w, axqd);
      }
    }
    v3cpy(k->xpos[b], pos); memcpy(k->xmat[b], mat, sizeof mat);
    v3cpy(k->w[b], w); v3cpy(k->al[b], al); v3cpy(k->vo[b], vo); v3cpy(k->ao[b], ao);
    m3mulv(r, mat, cm_body_ipos[b]);
    v3add(k->xipos[b], pos, r);
  }
}

/* translational / rotational Jacobian of a world point rigidly attached to body b */
static void jac_point(const Oracle* o, const Kin* k, int b, const double* p, double* Jv /*3xNV*/, double* Jw /*3xNV or NULL*/) {
  memset(Jv, 0, sizeof(double) * 3 * NV);
  if (Jw) memset(Jw, 0, sizeof(double) * 3 * NV);
  for (int j = 0; j < NV; j++) {
    if (!o->affects[b][j]) continue;
    if (cm_jnt_type[j] == 0) {
      for (int i = 0; i < 3; i++) Jv[i * NV + j] = k->axis[j][i];
    } else {
      double r[3], c[3];
      v3sub(r, p, k->anchor[j]);
      v3cross(c, k->axis[j], r);
      for (int i = 0; i < 3; i++) Jv[i * NV + j] = c[i];
      if (Jw) for (int i = 0; i < 3; i++) Jw[i * NV + j] = k->axis[j][i];
    }
  }
}

static void point_vel(const Kin* k, int b, const double* p, double* out) {
  double r[3], t[3];
  v3sub(r, p, k->xpos[b]);
  v3cross(t, k->w[b], r);
  v3add(out, k->vo[b], t);
}
/* velocity-product acceleration (qacc = 0, no gravity): RBDL CalcPointAcceleration(..., QDDot=0) */
static void point_acc(const Kin* k, int b, const double* p, double* out) {
  double r[3], t[3], t2[3];
  v3sub(r, p, k->xpos[b]);
  v3cross(t, k->w[b], r); v3cross(t2, k->w[b], t);
  v3cross(t, k->al[b], r);
  v3add(out, k->ao[b], t); v3add(out, out, t2);
}

static void world_inertia(const Kin* k, int b, double* Iw) {
  double t[9], Rt[9];
  const double* R = k->xmat[b];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
  m3mul(t, R, &cm_body_inertia[b][0][0]);
  m3mul(Iw, t, Rt);
}

/* M = sum_b m Jv'Jv + Jw' I Jw + armature   (== CRBA + rotor inertia, DynamicModel.cpp:267-272; == mj_crb) */
static void mass_matrix(const Oracle* o, const Kin* k, double* M) {
  memset(M, 0, sizeof(double) * NV * NV);
  for (int j = 0; j < NV; j++) M[j * NV + j] = cm_dof_armature[j];
  for (int b = 1; b < NB; b++) {
    double Jv[3 * NV], Jw[3 * NV], Iw[9], IJ[3 * NV];
    jac_point(o, k, b, k->xipos[b], Jv, Jw);
    world_inertia(k, b, Iw);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < NV; j++) IJ[i * NV + j] = Iw[3 * i] * Jw[j] + Iw[3 * i + 1] * Jw[NV + j] + Iw[3 * i + 2] * Jw[2 * NV + j];
    for (int i = 0; i < NV; i++)
      for (int j = 0; j < NV; j++) {
        double s = 0;
        for (int c = 0; c < 3; c++) s += cm_body_mass[b] * Jv[c * NV + i] * Jv[c * NV + j] + Jw[c * NV + i] * IJ[c * NV + j];
        M[i * NV + j] += s;
      }
  }
}

/* bias = C(q,v) + g(q)  (== RBDL NonlinearEffects, DynamicModel.cpp:320-323; == mj_rne with qacc=0) */
static void bias_force(const Oracle* o, const Kin* k, double gz, double* out) {
  memset(out, 0, sizeof(double) * NV);
  for (int b = 1; b < NB; b++) {
    double Jv[3 * NV], Jw[3 * NV], Iw[9], ac[3], F[3], N[3], Iwv[3], t[3];
    jac_point(o, k, b, k->xipos[b], Jv, Jw);
    world_inertia(k, b, Iw);
    point_acc(k, b, k->xipos[b], ac);
    ac[2] -= gz; /* inertial force against gravity: m (a - g) */
    for (int i = 0; i < 3; i++) F[i] = cm_body_mass[b] * ac[i];
    m3mulv(N, Iw, k->al[b]);
    m3mulv(Iwv, Iw, k->w[b]);
    v3cross(t, k->w[b], Iwv);
    v3add(N, N, t);
    for (int j = 0; j < NV; j++)
      for (int c = 0; c < 3; c++) out[j] += Jv[c * NV + j] * F[c] + Jw[c * NV + j] * N[c];
  }
}

/* ------------------------------------------------------------------ Part 2: MuJoCo forward dynamics */
static void site_world(const Kin* k, int s, double* p) {
  double r[3];
  m3mulv(r, k->xmat[cm_site_body[s]], cm_site_pos[s]);
  v3add(p, k->xpos[cm_site_body[s]], r);
}

static void make_frame(double* f) { /* mju_makeFrame: f[0..2] normal given, f[3..5] optional hint */
  v3normalize(f);
  if (sqrt(v3dot(f + 3, f + 3)) < 0.5) {
    v3set(f + 3, 0, 0, 0);
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double d = v3dot(f, f + 3);
  v3addscl(f + 3, f + 3, f, -d);
  v3normalize(f + 3);
  v3cross(f + 6, f, f + 3);
}

static int plane_sphere(Contact* c, const double* center, double radius) {
  /* floor: plane through the origin with normal +z (cassie2d_stiff.xml:53) */
  double dist = center[2] - radius;
  if (dist >= 0) return 0; /* margin = 0: only penetrating contacts are instantiated */
  c->dist = dist;
  v3set(c->pos, center[0], center[1], center[2] - radius - 0.5 * dist);
  memset(c->frame, 0, sizeof c->frame);
  c->frame[2] = 1.0;
  return 1;
}

/* Sphere against the terrain of a <geom type="hfield"> (terrain_random.py:38-76 adds one over the floor plane).
 * RESTATEMENT, not MuJoCo's algorithm: MuJoCo collides the sphere with the triangular prisms of the grid cells under it
 * through its general convex solver (mjc_ConvexHField), which has no closed form.  What is kept of it: the triangulation (every
 * cell split along its (c,r)-(c+1,r+1) diagonal) and the closest-feature semantics (face, edge or vertex, with that feature's
 * normal).  What is dropped: the mechanism lives in the sagittal plane, so the surface is met through its SECTION at the sphere's
 * own y -- a polyline in x-z whose vertices are the crossings of the cell edges (x = c dx) and of the cell diagonals
 * (x = (c + fy) dx) -- and the y-slope of the relief is ignored.  The sphere centre is compared with every section segment of
 * the cells that overlap [x - radius, x + radius]; the closest point gives distance and normal (r04; r03 only knew the extended
 * line under the centre).  MuJoCo may return one contact per prism; this keeps the closest feature only.
 * Outside the field's extent the floor plane z = 0 is the ground.  One contact per sphere, contact point half-way into the
 * penetration as for plane-sphere. */
static int hfield_sphere(const Oracle* o, Contact* c, const double* center, double radius) {
  const int nr = o->hf_nrow, nc = o->hf_ncol;
  const double dx = 2.0 * o->hf_sx / (nc - 1), dy = 2.0 * o->hf_sy / (nr - 1);
  const double gx = (center[0] + o->hf_sx) / dx, gy = (center[1] + o->hf_sy) / dy;
  if (!(gx >= 0.0 && gx <= (double)(nc - 1) && gy >= 0.0 && gy <= (double)(nr - 1))) return plane_sphere(c, center, radius);
  int ci = (int)gx, ri = (int)gy;
  if (ci > nc - 2) ci = nc - 2;
  if (ri > nr - 2) ri = nr - 2;
  const double fx = gx - ci, fy = gy - ri;
  /* cells whose x-extent overlaps [x - radius, x + radius] */
  int c0 = (int)floor(gx - radius / dx), c1 = (int)floor(gx + radius / dx);
  if (c0 < 0) c0 = 0;
  if (c1 > nc - 2) c1 = nc - 2;
  /* closest point of the section polyline (world x): per cell the segments E(c)-Dg(c) and Dg(c)-E(c+1) */
  double best = 1e300, bq[2] = {0, 0};
  for (int col = c0; col <= c1; col++) {
    double px[3], pz[3];
    const double h00 = o->hf[ri * nc + col], h01 = o->hf[(ri + 1) * nc + col];
    const double h10 = o->hf[ri * nc + col + 1], h11 = o->hf[(ri + 1) * nc + col + 1];
    px[0] = -o->hf_sx + col * dx;        pz[0] = (1.0 - fy) * h00 + fy * h01;
    px[1] = -o->hf_sx + (col + fy) * dx; pz[1] = (1.0 - fy) * h00 + fy * h11;
    px[2] = -o->hf_sx + (col + 1) * dx;  pz[2] = (1.0 - fy) * h10 + fy * h11;
    for (int s = 0; s < 2; s++) {
      const double ux = px[s + 1] - px[s], uz = pz[s + 1] - pz[s];
      const double len2 = ux * ux + uz * uz;
      double tt = len2 > 0 ? ((center[0] - px[s]) * ux + (center[2] - pz[s]) * uz) / len2 : 0.0;
      if (tt < 0) tt = 0;
      if (tt > 1) tt = 1;
      const double qx = px[s] + tt * ux, qz = pz[s] + tt * uz;
      const double dd = hypot(center[0] - qx, center[2] - qz);
      if (dd < best) { best = dd; bq[0] = qx; bq[1] = qz; }
    }
  }
  /* the face under the centre (triangle of cell ci chosen by the diagonal): side of the surface, fallback normal */
  const double z00 = o->hf[ri * nc + ci], z10 = o->hf[ri * nc + ci + 1], z01 = o->hf[(ri + 1) * nc + ci], z11 = o->hf[(ri + 1) * nc + ci + 1];
  double a, b;
  if (fy <= fx) { a = (z10 - z00) / dx; b = (z11 - z10) / dy; }
  else { a = (z11 - z01) / dx; b = (z01 - z00) / dy; }
  const double zs = z00 + a * (fx * dx) + b * (fy * dy);
  const double fnz = 1.0 / sqrt(1.0 + a * a), fnx = -a * fnz;
  const double above_face = (center[2] - zs) * fnz;
  double nx, nz, dist;
  if (above_face > 0 && best > 1e-12 && center[2] > bq[1]) {
    nx = (center[0] - bq[0]) / best; nz = (center[2] - bq[1]) / best; dist = best - radius;
  } else { /* centre at or under the surface: the face under it (not reached by the 20 mm spheres of this model) */
    nx = fnx; nz = fnz; dist = above_face - radius;
  }
  if (dist >= 0) return 0;
  const double back = radius + 0.5 * dist;
  c->dist = dist;
  v3set(c->pos, center[0] - nx * back, center[1], center[2] - nz * back);
  memset(c->frame, 0, sizeof c->frame);
  c->frame[0] = nx; c->frame[2] = nz;
  return 1;
}
static int ground_sphere(const Oracle* o, Contact* c, const double* center, double radius) {
  return o->hf ? hfield_sphere(o, c, center, radius) : plane_sphere(c, center, radius);
}

static void collide(Oracle* o) {
  const Kin* k = &o->kin;
  o->ncon = 0;
  if (!o->contact_enabled) return;
  for (int g = 0; g < CM_NGEOM; g++) {
    int b = cm_geom_body[g];
    double c[3], r[3];
    m3mulv(r, k->xmat[b], cm_geom_pos[g]);
    v3add(c, k->xpos[b], r);
    if (cm_geom_type[g] == 0) {
      Contact* cc = &o->con[o->ncon];
      if (ground_sphere(o, cc, c, cm_geom_radius[g])) { make_frame(cc->frame); cc->body = b; cc->geom = g; o->ncon++; }
    } else { /* mjc_PlaneCapsule: sphere tests at +axis end, then -axis end; frame y hint = capsule axis */
      double ax[3], e[3];
      m3mulv(ax, k->xmat[b], cm_geom_axis[g]);
      for (int s = 0; s < 2; s++) {
        Contact* cc = &o->con[o->ncon];
        v3addscl(e, c, ax, (s == 0 ? 1.0 : -1.0) * cm_geom_halflen[g]);
        if (ground_sphere(o, cc, e, cm_geom_radius[g])) {
          v3cpy(cc->frame + 3, ax);
          make_frame(cc->frame);
          cc->body = b; cc->geom = g; o->ncon++;
        }
      }
    }
  }
}

static int add_row(Oracle* o, int type, int id, const double* J, double pos, double margin, double diag) {
  int i = o->nefc++;
  o->efc_type[i] = type; o->efc_id[i] = id;
  memcpy(&o->efc_J[i * NV], J, sizeof(double) * NV);
  o->efc_pos[i] = pos; o->efc_margin[i] = margin; o->efc_diagApprox[i] = diag;
  return i;
}

static void make_constraints(Oracle* o) {
  const Kin* k = &o->kin;
  o->nefc = 0;
  /* equality: connect (mj_instantiateEquality, mjEQ_CONNECT) */
  for (int e = 0; e < NEQ; e++) {
    int b1 = cm_eq_body1[e], b2 = cm_eq_body2[e];
    double p1[3], p2[3], r[3], J1[3 * NV], J2[3 * NV];
    m3mulv(r, k->xmat[b1], cm_eq_anchor1[e]); v3add(p1, k->xpos[b1], r);
    m3mulv(r, k->xmat[b2], o->eq_anchor2[0][e]); v3add(p2, k->xpos[b2], r);
    jac_point(o, k, b1, p1, J1, NULL);
    jac_point(o, k, b2, p2, J2, NULL);
    for (int c = 0; c < 3; c++) {
      double J[NV];
      for (int j = 0; j < NV; j++) J[j] = J1[c * NV + j] - J2[c * NV + j];
      add_row(o, CT_EQUALITY, e, J, p1[c] - p2[c], 0.0, o->body_invweight0[b1] + o->body_invweight0[b2]);
    }
  }
  /* joint limits (mj_instantiateLimit), margin 0 */
  for (int j = 0; j < NV; j++) {
    if (!cm_jnt_limited[j]) continue;
    for (int side = 0; side < 2; side++) {
      double dist = side == 0 ? o->qpos[QADR(j)] - cm_jnt_range[j][0] : cm_jnt_range[j][1] - o->qpos[QADR(j)];
      if (dist < 0) {
        double J[NV] = {0};
        J[j] = side == 0 ? 1.0 : -1.0;
        add_row(o, CT_LIMIT, j, J, dist, 0.0, o->dof_invweight0[j]);
      }
    }
  }
  if (o->row_cap > 0 && o->nefc + 3 * o->ncon > o->row_cap) o->ncon = (o->row_cap - o->nefc) / 3; /* test switch, see row_cap */
  /* contacts, elliptic cone, condim 3 (mj_instantiateContact) */
  for (int c = 0; c < o->ncon; c++) {
    const Contact* cc = &o->con[c];
    double Jp[3 * NV];
    jac_point(o, k, cc->body, cc->pos, Jp, NULL); /* body1 is the world: J = J(body2) - 0 */
    for (int r = 0; r < 3; r++) {
      double J[NV];
      for (int j = 0; j < NV; j++)
        J[j] = cc->frame[3 * r] * Jp[j] + cc->frame[3 * r + 1] * Jp[NV + j] + cc->frame[3 * r + 2] * Jp[2 * NV + j];
      add_row(o, CT_CONTACT, c, J, r == 0 ? cc->dist : 0.0, 0.0, o->body_invweight0[cc->body]);
    }
  }
}

static double impedance(const double* solimp, double pos, double margin, int smoothstep) {
  if (solimp[0] == solimp[1] || solimp[2] <= MINVAL) return 0.5 * (solimp[0] + solimp[1]);
  double x = fabs((pos - margin) / solimp[2]);
  if (x >= 1) return solimp[1];
  if (x <= 0) return solimp[0];
  double y = x <= 0.5 ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x); /* MuJoCo 2.x defaults: midpoint 0.5, power 2 */
  if (smoothstep) y = x * x * (3 - 2 * x);                      /* ORC_ASSUME_IMP_SMOOTHSTEP: cubic sigmoid candidate */
  return solimp[0] + y * (solimp[1] - solimp[0]);
}

static void make_impedance(Oracle* o) {
  for (int i = 0; i < o->nefc; i++) {
    const double *solref, *solimp;
    switch (o->efc_type[i]) {
      case CT_EQUALITY: solref = cm_eq_solref[o->efc_id[i]]; solimp = cm_eq_solimp[o->efc_id[i]]; break;
      case CT_LIMIT: solref = cm_limit_solref; solimp = cm_limit_solimp; break;
      default: solref = cm_contact_solref; solimp = cm_contact_solimp; break;
    }
    double tc = solref[0], dr = solref[1], dmax = solimp[1];
    if (tc < 2 * CM_TIMESTEP) tc = 2 * CM_TIMESTEP; /* refsafe */
    double kk = 1.0 / (dmax * dmax * tc * tc * dr * dr), bb = 2.0 / (dmax * tc);
    double ipos = o->efc_pos[i];
    if ((o->assume & ORC_ASSUME_CONNECT_NORM_IMP) && o->efc_type[i] == CT_EQUALITY) {
      /* candidate: one impedance per connect constraint, evaluated at the NORM of its 3-vector violation */
      int i0 = i;
      while (i0 > 0 && o->efc_type[i0 - 1] == CT_EQUALITY && o->efc_id[i0 - 1] == o->efc_id[i]) i0--;
      ipos = sqrt(o->efc_pos[i0] * o->efc_pos[i0] + o->efc_pos[i0 + 1] * o->efc_pos[i0 + 1] + o->efc_pos[i0 + 2] * o->efc_pos[i0 + 2]);
    }
    double imp = impedance(solimp, ipos, o->efc_margin[i], o->assume & ORC_ASSUME_IMP_SMOOTHSTEP);
    double R = (1 - imp) / imp * o->efc_diagApprox[i];
    o->efc_R[i] = R > MINVAL ? R : MINVAL;
    o->efc_aref[i] = -bb * o->efc_vel[i] - kk * imp * (o->efc_pos[i] - o->efc_margin[i]);
  }
  /* elliptic contacts: friction rows share the normal row's regulariser (impratio 1, isotropic mu) */
  for (int i = 0; i < o->nefc; i++)
    if (o->efc_type[i] == CT_CONTACT) {
      o->efc_R[i + 1] = o->efc_R[i] / CM_IMPRATIO;
      o->efc_R[i + 2] = o->efc_R[i + 1] * cm_contact_friction[0] * cm_contact_friction[0] /
                        (cm_contact_friction[0] * cm_contact_friction[0]);
      i += 2;
    }
  for (int i = 0; i < o->nefc; i++) o->efc_D[i] = 1.0 / o->efc_R[i];
}

/* mju_QCQP2: min 0.5 x'Ax + b'x  s.t. sum (x_i/d_i)^2 <= r^2 */
static int qcqp2(double* res, const double* Ain, const double* bin, const double* d, double r) {
  double b1 = bin[0] * d[0], b2 = bin[1] * d[1];
  double A11 = Ain[0] * d[0] * d[0], A22 = Ain[3] * d[1] * d[1], A12 = Ain[1] * d[0] * d[1];
  double la = 0, v1 = 0, v2 = 0;
  for (int iter = 0; iter < 20; iter++) {
    double det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10) { res[0] = 0; res[1] = 0; return 0; }
    double detinv = 1 / det, P11 = (A22 + la) * detinv, P22 = (A11 + la) * detinv, P12 = -A12 * detinv;
    v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
    double val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10) break;
    double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * d[0]; res[1] = v2 * d[1];
  return la != 0;
}

static double cost_change(const double* A, double* force, const double* old, const double* res, int dim) {
  double delta[3], change = 0;
  for (int i = 0; i < dim; i++) delta[i] = force[i] - old[i];
  for (int i = 0; i < dim; i++) {
    double s = 0;
    for (int j = 0; j < dim; j++) s += A[i * dim + j] * delta[j];
    change += 0.5 * delta[i] * s + delta[i] * res[i];
  }
  if (change > 1e-10) { memcpy(force, old, sizeof(double) * dim); change = 0; }
  return change;
}

static void solve_pgs(Oracle* o) { /* mj_solPGS, elliptic cones */
  int n = o->nefc;
  double* f = o->efc_force;
  const double* AR = o->AR;
  double scale = 1.0 / (o->meaninertia * (NV > 1 ? NV : 1));
  const double mu = cm_contact_friction[0]; /* regularised mu == friction[0] (impratio 1) */
  const double fric[2] = {cm_contact_friction[0], cm_contact_friction[0]};
  o->solver_niter = 0;
  for (int iter = 0; iter < CM_ITERATIONS; iter++) {
    double improvement = 0;
    for (int i = 0; i < n; i++) {
      int dim = o->efc_type[i] == CT_CONTACT ? 3 : 1;
      double res[3], old[3], Athis[9];
      for (int j = 0; j < dim; j++) {
        double s = o->efc_b[i + j];
        for (int c = 0; c < n; c++) s += AR[(i + j) * n + c] * f[c];
        res[j] = s;
      }
      for (int j = 0; j < dim; j++) old[j] = f[i + j];
      for (int j = 0; j < dim; j++) for (int c = 0; c < dim; c++) Athis[j * dim + c] = AR[(i + j) * n + i + c];
      if (dim == 1) {
        f[i] -= res[0] / AR[i * n + i];
        if (o->efc_type[i] != CT_EQUALITY && f[i] < 0) f[i] = 0;
      } else {
        if (f[i] < MINVAL) { /* normal update */
          f[i] -= res[0] / AR[i * n + i];
          if (f[i] < 0) f[i] = 0;
          f[i + 1] = f[i + 2] = 0;
        } else { /* ray update */
          double v[3] = {f[i], f[i + 1], f[i + 2]}, v1[3], denom = 0;
          for (int j = 0; j < 3; j++) { v1[j] = Athis[3 * j] * v[0] + Athis[3 * j + 1] * v[1] + Athis[3 * j + 2] * v[2]; denom += v[j] * v1[j]; }
          if (denom >= MINVAL) {
            double x = -(v[0] * res[0] + v[1] * res[1] + v[2] * res[2]) / denom;
            if (f[i] + x * v[0] < 0) x = -f[i] / v[0];
            for (int j = 0; j < 3; j++) f[i + j] += x * v[j];
          }
        }
        if (f[i] >= MINVAL) { /* friction: QCQP on the cone boundary given the normal force */
          double Ac[4], bc[2], vv[2];
          for (int j = 0; j < 2; j++) {
            for (int c = 0; c < 2; c++) Ac[2 * j + c] = Athis[(j + 1) * 3 + (c + 1)];
            bc[j] = res[j + 1];
            for (int c = 0; c < 2; c++) bc[j] -= Ac[2 * j + c] * old[1 + c];
            bc[j] += Athis[(j + 1) * 3] * (f[i] - old[0]);
          }
          int active = qcqp2(vv, Ac, bc, fric, f[i]);
          if (active) {
            double s = vv[0] * vv[0] / (fric[0] * fric[0]) + vv[1] * vv[1] / (fric[1] * fric[1]);
            s = sqrt(f[i] * f[i] / (s > MINVAL ? s : MINVAL));
            vv[0] *= s; vv[1] *= s;
          }
          f[i + 1] = vv[0]; f[i + 2] = vv[1];
        }
        (void)mu;
      }
      improvement -= cost_change(Athis, f + i, old, res, dim);
      i += dim - 1;
    }
    o->solver_niter = iter + 1;
    if (improvement * scale < CM_TOLERANCE) break;
  }
}

/* mj_constraintUpdate (forces from a primal acceleration), used for the warm start */
static void constraint_update(Oracle* o, const double* jar) {
  const double mu = cm_contact_friction[0];
  double* f = o->efc_force;
  for (int i = 0; i < o->nefc; i++) {
    if (o->efc_type[i] == CT_EQUALITY) f[i] = -o->efc_D[i] * jar[i];
    else if (o->efc_type[i] == CT_LIMIT) f[i] = jar[i] < 0 ? -o->efc_D[i] * jar[i] : 0.0;
    else {
      double U[3] = {jar[i] * mu, jar[i + 1] * cm_contact_friction[0], jar[i + 2] * cm_contact_friction[0]};
      double N = U[0], T = sqrt(U[1] * U[1] + U[2] * U[2]);
      if (N >= mu * T || (T <= 0 && N >= 0)) { f[i] = f[i + 1] = f[i + 2] = 0; }
      else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { for (int j = 0; j < 3; j++) f[i + j] = -o->efc_D[i + j] * jar[i + j]; }
      else {
        double Dm = o->efc_D[i] / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
        f[i] = -Dm * NmT * mu;
        for (int j = 1; j < 3; j++) f[i + j] = -f[i] / T * U[j] * cm_contact_friction[0];
      }
      i += 2;
    }
  }
}

static void minv_mul(const Oracle* o, const double* b, double* x) { chol_solve(NV, o->L, b, x); }

void orc_forward(Oracle* o) {
  kinematics(o, 0, o->qpos, o->qvel, &o->kin);
  mass_matrix(o, &o->kin, o->M);
  if (chol_factor(NV, o->M, o->L)) { fprintf(stderr, "oracle: mass matrix not SPD\n"); abort(); }
  bias_force(o, &o->kin, o->gravity_z, o->bias);
  for (int j = 0; j < NV; j++) { o->passive[j] = -o->damping[j] * o->qvel[j]; o->actuator[j] = 0; }
  for (int a = 0; a < NU; a++) { /* mj_fwdActuation: clamp ctrl, gain 1, joint transmission with gear */
    double c = o->ctrl[a];
    if (c < cm_act_ctrlrange[a][0]) c = cm_act_ctrlrange[a][0];
    if (c > cm_act_ctrlrange[a][1]) c = cm_act_ctrlrange[a][1];
    o->actuator[cm_act_dof[a]] += cm_act_gear[a] * c;
  }
  for (int j = 0; j < NV; j++) o->qfrc_smooth[j] = o->passive[j] - o->bias[j] + o->actuator[j];
  minv_mul(o, o->qfrc_smooth, o->qacc_smooth);
  collide(o);
  make_constraints(o);
  int n = o->nefc;
  if (n == 0) {
    memcpy(o->qacc, o->qacc_smooth, sizeof o->qacc);
    if (!(o->assume & ORC_ASSUME_WS_STEP_ONLY)) memcpy(o->qacc_ws, o->qacc, sizeof o->qacc);
    o->solver_niter = 0;
    return;
  }
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int j = 0; j < NV; j++) s += o->efc_J[i * NV + j] * o->qvel[j];
    o->efc_vel[i] = s;
  }
  make_impedance(o);
  /* AR = J M^-1 J' + diag(R) */
  static __thread double MinvJT[ORC_MAXEFC * NV];
  for (int i = 0; i < n; i++) minv_mul(o, &o->efc_J[i * NV], &MinvJT[i * NV]);
  for (int i = 0; i < n; i++)
    for (int c = 0; c < n; c++) {
      double s = 0;
      for (int j = 0; j < NV; j++) s += o->efc_J[i * NV + j] * MinvJT[c * NV + j];
      o->AR[i * n + c] = s + (i == c ? o->efc_R[i] : 0.0);
    }
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int j = 0; j < NV; j++) s += o->efc_J[i * NV + j] * o->qacc_smooth[j];
    o->efc_b[i] = s - o->efc_aref[i];
  }
  /* warm start from qacc_warmstart (mj_fwdConstraint); PGS keeps it only if its dual cost beats zero force */
  {
    double jar[ORC_MAXEFC];
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int j = 0; j < NV; j++) s += o->efc_J[i * NV + j] * o->qacc_ws[j];
      jar[i] = s - o->efc_aref[i];
    }
    constraint_update(o, jar);
    double cost = 0;
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int c = 0; c < n; c++) s += o->AR[i * n + c] * o->efc_force[c];
      cost += o->efc_force[i] * (0.5 * s + o->efc_b[i]);
    }
    if (cost > 0) memset(o->efc_force, 0, sizeof(double) * n);
  }
  solve_pgs(o);
  for (int j = 0; j < NV; j++) {
    double s = o->qacc_smooth[j];
    for (int i = 0; i < n; i++) s += MinvJT[i * NV + j] * o->efc_force[i];
    o->qacc[j] = s;
  }
  if (!(o->assume & ORC_ASSUME_WS_STEP_ONLY)) memcpy(o->qacc_ws, o->qacc, sizeof o->qacc);
}

static void euler(Oracle* o) { /* mj_Euler: joint damping integrated implicitly, then semi-implicit positions */
  const double h = CM_TIMESTEP;
  double qfrc[NV], MhB[NV * NV], Lh[NV * NV], qacc[NV];
  /* total force = M qacc */
  for (int i = 0; i < NV; i++) {
    double s = 0;
    for (int j = 0; j < NV; j++) s += o->M[i * NV + j] * o->qacc[j];
    qfrc[i] = s;
  }
  int damped = 0;
  for (int j = 0; j < NV; j++) damped |= o->damping[j] > 0;
  if (damped) {
    memcpy(MhB, o->M, sizeof MhB);
    for (int j = 0; j < NV; j++) MhB[j * NV + j] += h * o->damping[j];
    chol_factor(NV, MhB, Lh);
    chol_solve(NV, Lh, qfrc, qacc);
  } else {
    memcpy(qacc, o->qacc, sizeof qacc);
  }
  for (int j = 0; j < NV; j++) o->qvel[j] += h * qacc[j];
  for (int j = 0; j < NV; j++) {
    if (cm_jnt_type[j] != 2) { o->qpos[QADR(j)] += h * o->qvel[j]; continue; }
    if (cm_jnt_axis[j][0] != 1.0) continue;
    /* mju_quatIntegrate: quat <- normalize(quat) * axisangle(omega_body, h |omega|) */
    double* qq = o->qpos + QADR(j);
    double wx = o->qvel[j], wy = o->qvel[j + 1], wz = o->qvel[j + 2], wn = sqrt(wx * wx + wy * wy + wz * wz);
    double ax[3] = {1, 0, 0}, ang = 0;
    if (wn >= MINVAL) { ax[0] = wx / wn; ax[1] = wy / wn; ax[2] = wz / wn; ang = h * wn; }
    double sh = sin(0.5 * ang), r[4] = {cos(0.5 * ang), ax[0] * sh, ax[1] * sh, ax[2] * sh};
    double n = sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    double a[4] = {qq[0] / n, qq[1] / n, qq[2] / n, qq[3] / n};
    qq[0] = a[0] * r[0] - a[1] * r[1] - a[2] * r[2] - a[3] * r[3];
    qq[1] = a[0] * r[1] + a[1] * r[0] + a[2] * r[3] - a[3] * r[2];
    qq[2] = a[0] * r[2] - a[1] * r[3] + a[2] * r[0] + a[3] * r[1];
    qq[3] = a[0] * r[3] + a[1] * r[2] - a[2] * r[1] + a[3] * r[0];
  }
}

static void mj_step(Oracle* o) {
  orc_forward(o);
  memcpy(o->qacc_ws, o->qacc, sizeof o->qacc); /* a step always leaves its qacc as the next warm start */
  euler(o);
}

/* ------------------------------------------------------------------ Part 3: DynamicModel / DynamicState / Cassie2d */
static void set_state(Oracle* o) { /* DynamicModel::setState (DynamicModel.cpp:237-242) */
  memcpy(o->kin_qpos, o->qpos, sizeof o->qpos);
  memcpy(o->kin_qvel, o->qvel, sizeof o->qvel);
}

#ifndef ORC_CASSIE3D
static const int contact_site_ids[4] = {2, 3, 4, 5}; /* Cassie2d.cpp:34 */
static const int target_site_ids[5] = {1, 2, 3, 4, 5}; /* Cassie2d.cpp:35-36 */

static void update_dynamic_state(Oracle* o, Kin* k) { /* DynamicState::UpdateDynamicState (DynamicState.cpp:45-91) */
  DynState* d = &o->ds;
  kinematics(o, 1, o->kin_qpos, o->kin_qvel, k);
  mass_matrix(o, k, d->M);
  bias_force(o, k, CM_GRAVITY_Z, d->bias);
  for (int j = 0; j < NV; j++) d->bias[j] += cm_dof_damping[j] * o->kin_qvel[j]; /* bias -= passive */
  memset(d->Bt, 0, sizeof d->Bt);
  for (int a = 0; a < NU; a++) d->Bt[cm_act_dof[a] * NU + a] = cm_act_gear[a];
  for (int c = 0; c < 4; c++) {
    double p[3];
    site_world(k, contact_site_ids[c], p);
    jac_point(o, k, cm_site_body[contact_site_ids[c]], p, &d->Jc[3 * c * NV], NULL);
  }
  for (int e = 0; e < NEQ; e++) {
    int b1 = cm_eq_body1[e], b2 = cm_eq_body2[e];
    double p1[3], p2[3], r[3], J1[3 * NV], J2[3 * NV], a1[3], a2[3];
    m3mulv(r, k->xmat[b1], cm_eq_anchor1[e]); v3add(p1, k->xpos[b1], r);
    m3mulv(r, k->xmat[b2], o->eq_anchor2[1][e]); v3add(p2, k->xpos[b2], r);
    jac_point(o, k, b1, p1, J1, NULL);
    jac_point(o, k, b2, p2, J2, NULL);
    for (int i = 0; i < 3 * NV; i++) d->Jeq[3 * e * NV + i] = J1[i] - J2[i];
    point_acc(k, b1, p1, a1); point_acc(k, b2, p2, a2);
    for (int c = 0; c < 3; c++) d->JeqdotQdot[3 * e + c] = a1[c] - a2[c];
  }
}

#endif /* !ORC_CASSIE3D */

void orc_step_torque(Oracle* o, const double* torques) {
  set_state(o); /* UpdateDynamicState result is unused in this mode */
  memcpy(o->ctrl, torques, sizeof o->ctrl);
  mj_step(o);
}

#ifndef ORC_CASSIE3D
void orc_step_pd(Oracle* o, const double* angles) {
  static const int joints[NU] = {3, 4, 6, 8, 9, 11};
  set_state(o);
  for (int i = 0; i < NU; i++)
    o->ctrl[i] = 10.0 * (angles[i] - o->qpos[joints[i]]) + 5.0 * (0.0 - o->qvel[joints[i]]);
  mj_step(o);
}

#include "cassie_oracle_ctrl.inc"
#endif /* !ORC_CASSIE3D */

void orc_get_state(const Oracle* o, double* qpos, double* qvel) {
  memcpy(qpos, o->qpos, sizeof o->qpos);
  memcpy(qvel, o->qvel, sizeof o->qvel);
}

#ifndef ORC_CASSIE3D
void orc_get_opstate(const Oracle* o, int flags, double* s) {
  /* GetOperationalSpaceState (Cassie2d.cpp:218-237): kinematics of the LAST setState (quirk Q1/Q2),
   * pitch and pitch rate from the current mj_data (Q1), element [2] of the foot vectors never written (Q4). */
  Kin k;
  const double* q = (flags & ORC_FIX_STALE_KIN) ? o->qpos : o->kin_qpos;
  const double* v = (flags & ORC_FIX_STALE_KIN) ? o->qvel : o->kin_qvel;
  double x[15], xd[15];
  kinematics(o, 1, q, v, &k);
  for (int i = 0; i < 5; i++) {
    int sid = target_site_ids[i];
    site_world(&k, sid, &x[3 * i]);
    point_vel(&k, cm_site_body[sid], &x[3 * i], &xd[3 * i]);
  }
  memset(s, 0, 18 * sizeof(double));
  for (int i = 0; i < 2; i++) {
    s[i] = x[i * 2];
    s[3 + i] = xd[i * 2];
    s[6 + i] = (x[i * 2 + 3] + x[i * 2 + 6]) / 2.0;
    s[9 + i] = (xd[i * 2 + 3] + xd[i * 2 + 6]) / 2.0;
    s[12 + i] = (x[i * 2 + 9] + x[i * 2 + 12]) / 2.0;
    s[15 + i] = (xd[i * 2 + 9] + xd[i * 2 + 12]) / 2.0;
  }
  s[2] = o->qpos[2];
  s[5] = o->qvel[2];
}

#endif /* !ORC_CASSIE3D */

/* ------------------------------------------------------------------ lifecycle */
static void derive_constants(Oracle* o) {
  /* what MuJoCo's compiler (set0) and DynamicModel::LoadModel derive at load time */
  for (int b = 0; b < NB; b++)
    for (int j = 0; j < NV; j++) {
      int a = b, hit = 0;
      while (a > 0) { if (a == cm_jnt_body[j]) hit = 1; a = cm_body_parent[a]; }
      o->affects[b][j] = (unsigned char)hit;
    }
  for (int sem = 0; sem < 2; sem++) {
    Kin k;
    kinematics(o, sem, QPOS0, NULL, &k);
    for (int e = 0; e < NEQ; e++) {
      double r[3], pw[3];
      m3mulv(r, k.xmat[cm_eq_body1[e]], cm_eq_anchor1[e]);
      v3add(pw, k.xpos[cm_eq_body1[e]], r);
      v3sub(r, pw, k.xpos[cm_eq_body2[e]]);
      m3tmulv(o->eq_anchor2[sem][e], k.xmat[cm_eq_body2[e]], r);
    }
    if (sem == 0) {
      double M0[NV * NV], L0[NV * NV], e[NV], col[NV], tr = 0;
      mass_matrix(o, &k, M0);
      chol_factor(NV, M0, L0);
      for (int j = 0; j < NV; j++) tr += M0[j * NV + j];
      o->meaninertia = tr / NV;
      for (int j = 0; j < NV; j++) {
        memset(e, 0, sizeof e); e[j] = 1;
        chol_solve(NV, L0, e, col);
        o->dof_invweight0[j] = col[j];
      }
      for (int j = 0; j + 2 < NV; j++) /* free joint: translational and rotational triples are averaged (mj_setConst) */
        if (cm_jnt_type[j] == 2 && cm_jnt_axis[j][0] == 1.0) {
          double tr3 = (o->dof_invweight0[j - 3] + o->dof_invweight0[j - 2] + o->dof_invweight0[j - 1]) / 3.0;
          double ro3 = (o->dof_invweight0[j] + o->dof_invweight0[j + 1] + o->dof_invweight0[j + 2]) / 3.0;
          for (int c = 0; c < 3; c++) { o->dof_invweight0[j - 3 + c] = tr3; o->dof_invweight0[j + c] = ro3; }
        }
      o->body_invweight0[0] = 0;
      for (int b = 1; b < NB; b++) {
        double Jv[3 * NV], x[NV], s = 0;
        jac_point(o, &k, b, k.xipos[b], Jv, NULL);
        for (int c = 0; c < 3; c++) {
          chol_solve(NV, L0, &Jv[c * NV], x);
          for (int j = 0; j < NV; j++) s += Jv[c * NV + j] * x[j];
        }
        o->body_invweight0[b] = s / 3.0;
      }
    }
  }
}

#ifdef ORC_CASSIE3D
#define qpos_init cm_qpos_init /* standing pose: the sagittal angles of Cassie2d.cpp:56-58 on an upright floating base */
#else
static const double qpos_init[NV] = {0.0, 0.939, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407,
                                     0.68111815, -1.40730353, 1.62972043, -1.77611107, -0.61968402}; /* Cassie2d.cpp:56-58 */
#endif

Oracle* orc_create(void) {
  Oracle* o = (Oracle*)calloc(1, sizeof *o);
  o->gravity_z = CM_GRAVITY_Z;
  memcpy(o->damping, cm_dof_damping, sizeof o->damping);
  o->contact_enabled = 1;
  derive_constants(o);
  memcpy(o->qpos, qpos_init, sizeof o->qpos);
  orc_forward(o);  /* mj_forward (Cassie2d.cpp:62) */
  set_state(o);    /* dyn_model_.setState (Cassie2d.cpp:64) */
  return o;
}
void orc_free(Oracle* o) { free(o); }

void orc_reset(Oracle* o, const double* qpos, const double* qvel) {
  memcpy(o->qpos, qpos, sizeof o->qpos);
  memcpy(o->qvel, qvel, sizeof o->qvel);
  orc_forward(o); /* NB: no setState here (quirk Q2) */
}

/* ------------------------------------------------------------------ test hooks */
void orc_set_state_raw(Oracle* o, const double* qpos, const double* qvel, const double* ws) {
  memcpy(o->qpos, qpos, sizeof o->qpos);
  memcpy(o->qvel, qvel, sizeof o->qvel);
  if (ws) memcpy(o->qacc_ws, ws, sizeof o->qacc_ws);
  set_state(o);
}
void orc_set_gravity(Oracle* o, double gz) { o->gravity_z = gz; }
void orc_set_damping_scale(Oracle* o, double s) { for (int j = 0; j < NV; j++) o->damping[j] = s * cm_dof_damping[j]; }
void orc_set_contact_enabled(Oracle* o, int e) { o->contact_enabled = e; }
void orc_set_assumptions(Oracle* o, int mask) { o->assume = mask; }
void orc_set_row_cap(Oracle* o, int cap) { o->row_cap = cap; }
void orc_get_contacts(const Oracle* o, double* dist, double* pos, double* frame) {
  for (int i = 0; i < o->ncon; i++) {
    if (dist) dist[i] = o->con[i].dist;
    if (pos) memcpy(pos + 3 * i, o->con[i].pos, sizeof(double) * 3);
    if (frame) memcpy(frame + 9 * i, o->con[i].frame, sizeof(double) * 9);
  }
}
void orc_set_hfield(Oracle* o, const double* heights_m, int nrow, int ncol, double size_x, double size_y) {
  o->hf = (heights_m && nrow >= 2 && ncol >= 2) ? heights_m : NULL; /* caller keeps the array alive */
  o->hf_nrow = nrow; o->hf_ncol = ncol; o->hf_sx = size_x; o->hf_sy = size_y;
}
void orc_get_efc_extra(const Oracle* o, double* R, double* vel, double* diagApprox, double* b) {
  if (R) memcpy(R, o->efc_R, sizeof(double) * o->nefc);
  if (vel) memcpy(vel, o->efc_vel, sizeof(double) * o->nefc);
  if (diagApprox) memcpy(diagApprox, o->efc_diagApprox, sizeof(double) * o->nefc);
  if (b) memcpy(b, o->efc_b, sizeof(double) * o->nefc);
}
int orc_nefc(const Oracle* o) { return o->nefc; }
int orc_ncon(const Oracle* o) { return o->ncon; }
int orc_solver_niter(const Oracle* o) { return o->solver_niter; }
void orc_get_mass_matrix(const Oracle* o, int sem, const double* qpos, double* M) {
  Kin k; kinematics(o, sem, qpos, NULL, &k); mass_matrix(o, &k, M);
}
void orc_get_bias(const Oracle* o, int sem, const double* qpos, const double* qvel, double* bias) {
  Kin k; kinematics(o, sem, qpos, qvel, &k); bias_force(o, &k, o->gravity_z, bias);
}
/* n independent mechanisms, n_sub torque-mode steps each, OpenMP over mechanisms (CPU baseline of the batched kernels) */
void orc_batch_step_torque(Oracle** os, int n, const double* torques /*[n][NU]*/, int n_sub, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
  for (int i = 0; i < n; i++)
    for (int s = 0; s < n_sub; s++) orc_step_torque(os[i], torques + (size_t)i * NU);
}
void orc_get_qacc(const Oracle* o, double* qacc) { memcpy(qacc, o->qacc, sizeof o->qacc); }
void orc_get_warmstart(const Oracle* o, double* w) { memcpy(w, o->qacc_ws, sizeof o->qacc_ws); }
void orc_get_ctrl(const Oracle* o, double* c) { memcpy(c, o->ctrl, sizeof o->ctrl); }
void orc_set_ctrl(Oracle* o, const double* c) { memcpy(o->ctrl, c, sizeof o->ctrl); } /* then orc_forward: mj_forward with these controls */
void orc_get_efc(const Oracle* o, double* J, double* force, double* pos, double* aref, int* type) {
  if (J) memcpy(J, o->efc_J, sizeof(double) * o->nefc * NV);
  if (force) memcpy(force, o->efc_force, sizeof(double) * o->nefc);
  if (pos) memcpy(pos, o->efc_pos, sizeof(double) * o->nefc);
  if (aref) memcpy(aref, o->efc_aref, sizeof(double) * o->nefc);
  if (type) memcpy(type, o->efc_type, sizeof(int) * o->nefc);
}
double orc_energy(const Oracle* o, double* kinetic, double* potential) {
  Kin k; double M[NV * NV], ke = 0, pe = 0;
  kinematics(o, 0, o->qpos, o->qvel, &k);
  mass_matrix(o, &k, M);
  for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) ke += 0.5 * o->qvel[i] * M[i * NV + j] * o->qvel[j];
  for (int b = 1; b < NB; b++) pe += -cm_body_mass[b] * o->gravity_z * k.xipos[b][2];
  if (kinetic) *kinetic = ke;
  if (potential) *potential = pe;
  return ke + pe;
}
void orc_site_pos(const Oracle* o, int sem, const double* qpos, int site, double* p) {
  Kin k; kinematics(o, sem, qpos, NULL, &k); site_world(&k, site, p);
}
void orc_get_model_consts(const Oracle* o, int sem, double* a2, double* dw, double* bw, double* mi) {
  if (a2) memcpy(a2, o->eq_anchor2[sem], sizeof o->eq_anchor2[sem]);
  if (dw) memcpy(dw, o->dof_invweight0, sizeof o->dof_invweight0);
  if (bw) memcpy(bw, o->body_invweight0, sizeof o->body_invweight0);
  if (mi) *mi = o->meaninertia;
}
#ifndef ORC_CASSIE3D
void orc_get_dynamic_state(Oracle* o, double* M, double* bias, double* Bt, double* Jc, double* Jeq, double* Jd) {
  Kin k; update_dynamic_state(o, &k);
  if (M) memcpy(M, o->ds.M, sizeof o->ds.M);
  if (bias) memcpy(bias, o->ds.bias, sizeof o->ds.bias);
  if (Bt) memcpy(Bt, o->ds.Bt, sizeof o->ds.Bt);
  if (Jc) memcpy(Jc, o->ds.Jc, sizeof o->ds.Jc);
  if (Jeq) memcpy(Jeq, o->ds.Jeq, sizeof o->ds.Jeq);
  if (Jd) memcpy(Jd, o->ds.JeqdotQdot, sizeof o->ds.JeqdotQdot);
}
void orc_get_osc_qp(const Oracle* o, double* x39, double* kkt4) {
  if (x39) memcpy(x39, o->qp_x, sizeof o->qp_x);
  if (kkt4) memcpy(kkt4, o->qp_kkt, sizeof o->qp_kkt);
}

#include "cassie_oracle_env.inc"
#else
int orc_dims(int* nq, int* nv, int* nu) { *nq = NQ; *nv = NV; *nu = NU; return NB; }
#endif /* !ORC_CASSIE3D */
