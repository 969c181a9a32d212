// cassie_terrain.h -- sphere against the height-field terrain (the <hfield> + <geom type='hfield'> pair that
// rllab/envs/terrain_random.py:38-76 writes into the MJCF), shared by every kernel family's height-field instantiation and by the
// CPU instantiation of the two-lanes-per-environment core (oracle/leg_host/).
//
// The mechanism lives in the sagittal plane, so the terrain is met through its CROSS-SECTION at the sphere's own y.  MuJoCo
// triangulates every grid cell along its (c, r)-(c+1, r+1) diagonal; at a fixed y inside grid row r (fraction fy) the section of
// that surface is a POLYLINE with two vertices per cell:
//     E(c)  at x = c dx        height h[r][c] + fy (h[r+1][c]   - h[r][c])     (crossing of the cell edge)
//     Dg(c) at x = (c + fy) dx height h[r][c] + fy (h[r+1][c+1] - h[r][c])     (crossing of the diagonal)
// r04: the sphere is tested against the CLOSEST FEATURE of that polyline within its reach -- every segment of the cells that overlap
// [x - radius, x + radius], each as a face (projection inside the segment) or through an end vertex -- with the feature's own normal:
// the face normal, or the direction from the vertex to the centre at a convex corner.  (r03 tested the extended line of the segment
// under the centre only: 1.37 mm early contact with the wrong normal at a ridge, tests/test_oracle_terrain.py.)  What remains a
// restatement is the restriction to the section: the y-slope of the relief, which a planar mechanism cannot feel, is ignored,
// MuJoCo's general convex solver on the prisms (mjc_ConvexHField) is replaced by this closed form, and a sphere makes ONE contact
// (its closest feature) where MuJoCo may return one per prism.  Outside the field: the floor plane z = 0.
// Contact frame = (normal (nx, nz), tangent (nz, -nx)); for a height field nz > 0.
#ifndef CASSIE_TERRAIN_H_
#define CASSIE_TERRAIN_H_
#include <math.h>
#include <stddef.h>

#include "cassie_vec_layout.h"

#ifndef CASSIE_TERRAIN_FN
#ifdef __HIPCC__
#define CASSIE_TERRAIN_FN __device__ __forceinline__
#else
#define CASSIE_TERRAIN_FN inline
#endif
#endif

namespace cassie {

// squared distance from (cx, cz) to the segment (ax, az)-(bx, bz) and the closest point on it
CASSIE_TERRAIN_FN void terrain_seg(double cx, double cz, double ax, double az, double bx, double bz, double& d2, double& qx, double& qz) {
  const double ux = bx - ax, uz = bz - az;
  const double l2 = ux * ux + uz * uz;
  double t = l2 > 0.0 ? ((cx - ax) * ux + (cz - az) * uz) / l2 : 0.0;
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  qx = ax + t * ux; qz = az + t * uz;
  d2 = (cx - qx) * (cx - qx) + (cz - qz) * (cz - qz);
}

CASSIE_TERRAIN_FN void terrain_sphere(const Terrain& t, double wx, double wy, double wz, double radius, double& dist, double& nx, double& nz) {
  const int nr = t.nrow, nc = t.ncol;
  const double dx = 2.0 * t.sx / (nc - 1), dy = 2.0 * t.sy / (nr - 1);
  const double gx = (wx + t.sx) / dx, gy = (wy + t.sy) / dy;
  nx = 0.0; nz = 1.0; dist = wz - radius;
  if (!(gx >= 0.0 && gx <= (double)(nc - 1) && gy >= 0.0 && gy <= (double)(nr - 1))) return;
  if (dist > t.hmax) { dist = dist - t.hmax; return; }   // clear of the highest point of the field (most spheres of a standing robot): a lower bound, > 0
  int ci = (int)gx, ri = (int)gy;
  ci = ci > nc - 2 ? nc - 2 : ci;
  ri = ri > nr - 2 ? nr - 2 : ri;
  const double fx = gx - ci, fy = gy - ri;
  const double* r0 = t.h + (size_t)ri * nc;
  const double* r1 = r0 + nc;
  // the face under the centre (lower triangle of the cell: Dg(ci)-E(ci+1), upper: E(ci)-Dg(ci)): which side of the surface the
  // centre is on, and the normal to fall back on
  const double z00 = r0[ci], z10 = r0[ci + 1], z01 = r1[ci], z11 = r1[ci + 1];
  const double a = (fy <= fx ? z10 - z00 : z11 - z01) / dx;                      // slope of that segment
  const double fnz = 1.0 / sqrt(1.0 + a * a), fnx = -a * fnz;
  const double zd = z00 + fy * (z11 - z00);                                      // Dg(ci), at x = fy dx
  const double fdist = (wz - (zd + a * ((fx - fy) * dx))) * fnz;
  // closest point of the section inside the sphere's reach (x relative to the left edge of cell ci)
  const double reach = radius / dx;
  int c0 = (int)(gx - reach), c1 = (int)(gx + reach);
  c0 = gx - reach < 0.0 ? 0 : c0;
  c1 = c1 > nc - 2 ? nc - 2 : c1;
  const double cx = fx * dx;
  double d2 = 1e300, qx = 0.0, qz = 0.0;
  for (int c = c0; c <= c1; c++) {
    const double xe = (double)(c - ci) * dx;
    const double h00 = r0[c], h10 = r0[c + 1];
    const double ze = h00 + fy * (r1[c] - h00), zg = h00 + fy * (r1[c + 1] - h00), zn = h10 + fy * (r1[c + 1] - h10);
    double e2, ex, ez;
    terrain_seg(cx, wz, xe, ze, xe + fy * dx, zg, e2, ex, ez);
    if (e2 < d2) { d2 = e2; qx = ex; qz = ez; }
    terrain_seg(cx, wz, xe + fy * dx, zg, xe + dx, zn, e2, ex, ez);
    if (e2 < d2) { d2 = e2; qx = ex; qz = ez; }
  }
  const double d = sqrt(d2);
  // centre above the surface: distance to the closest feature along (centre - closest point); at or below it (never in practice:
  // the radii are 20 mm and more, penetrations fractions of a millimetre): the face under the centre
  const bool above = fdist > 0.0 && d > 1e-12 && wz - qz > 0.0;
  nx = above ? (cx - qx) / d : fnx;
  nz = above ? (wz - qz) / d : fnz;
  dist = (above ? d : fdist) - radius;
}

}  // namespace cassie
#endif
