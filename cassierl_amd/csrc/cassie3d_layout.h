// cassie3d_layout.h -- HBM record, debug record and kernel parameter block of the batched Cassie3d physics
// (shared by cassie3d_kernels.hip and the host side; not a public header).
#ifndef CASSIE3D_LAYOUT_H_
#define CASSIE3D_LAYOUT_H_

namespace cassie3d {

constexpr int MAXR = 64;        // constraint rows of the general kernel
constexpr int MAXR_FAST = 32;   // ... of the high-occupancy kernel (8 KB instead of 32 KB for A): 6 connect rows + 8 contacts, say
constexpr double MINVAL = 1e-15;
// HBM record of one environment (doubles)
constexpr int ENV3_STRIDE = 80;
enum { E3_Q = 0, E3_V = 21, E3_WS = 41, E3_CTRL = 61, E3_TIME = 71, E3_NITER = 72, E3_NEFC = 73, E3_OVF = 74 };
enum { K_NONE = 0, K_EQ = 1, K_LIM = 2, K_CN = 3, K_CT = 4 };
// debug record (tests only)
enum { D3_M = 0, D3_BIAS = 400, D3_QS = 420, D3_NEFC = 440, D3_QACC = 441, D3_F = 461, D3_AREF = 525, D3_J = 589, D3_STRIDE = 589 + 64 * 20 };

struct Params3 {
  double* state;           // [n][ENV3_STRIDE]
  const double* actions;   // [n][10] motor commands (pre-clamp), device
  double* debug;           // [n][D3_STRIDE] or null
  const int* pending_in;   // [n] substeps to do per env (second pass) or null: n_sub for everyone
  int* pending_out;        // [n] substeps NOT done because the env needed more rows than this kernel has, or null: flag E3_OVF
  unsigned long long* stats;  // [S3_N] event counters of the handle (rare-path atomics only)
  int n_envs, n_sub, integrate;
  // the lane-per-leg kernel on a SEGMENT of the step's substeps (cassie_cabi.hip: launch3d): gone[env] != 0 = the environment left that
  // kernel in an earlier segment (not touched; seg_first: written for everyone); pending_out counts to the END of the step (seg_later
  // = substeps of the segments behind this one).  gone == null: the whole step in one launch.
  int* gone;
  int seg_first, seg_later;
};
// S3_GENERAL_SUBSTEPS: env-substeps the 32-row kernel handed to the 64-row kernel; S3_CAPPED_SUBSTEPS: env-substeps in which
// the 64-row kernel had to leave contacts out (more than 64 constraint rows)
// S3_LEG_HANDOVER_SUBSTEPS: env-substeps the lane-per-leg kernel handed to the wavefront-per-environment kernels (row capacity)
enum { S3_GENERAL_SUBSTEPS = 0, S3_CAPPED_SUBSTEPS = 1, S3_LEG_HANDOVER_SUBSTEPS = 2, S3_N = 4 };

}  // namespace cassie3d
#endif
