/* cassie3d_vec.h -- batched Cassie3d physics (model/cassie3d_stiff.xml) on MI355X, C ABI of libcassie2d.so.
 *
 * BASELINE.json configs[4] / SURVEY.md section 8 row N3.  The reference has NO Cassie3d interface to mirror: it ships the
 * MJCF (model/cassie3d_stiff.xml:60-193, actuators :184-195) and vestigial hooks only (src/xml_parser.h:321-323,
 * include/RobotInterface.h:54, src/DynamicModel.cpp:250-265).  These entry points therefore follow the conventions of
 * include/cassie_vec.h (the batched form of src/Cassie2d/Cassie2d.cpp:15-27): opaque handle, plain pointers and sizes, int
 * error codes (CASSIE_OK ... from cassie_vec.h), device pointers unless the name ends in Host.
 *
 * What one call computes: n_sub MuJoCo steps (mj_step: PGS / elliptic cones / Euler with implicit joint damping, options of
 * cassie3d_stiff.xml:5) of every environment with the given motor commands held constant -- the 3-D counterpart of
 * Cassie2d::Step (src/Cassie2d/Cassie2d.cpp:86-94).  nq = 21 (world position, unit quaternion w x y z, 14 hinge angles in
 * MJCF order), nv = 20 (world linear velocity, body-frame angular velocity, hinge rates), nu = 10 (MJCF actuator order).
 *
 * No CPU path: Cassie3dVecCreate fails with CASSIE_ENODEVICE when no HIP device is present.
 */
#ifndef CASSIE3D_VEC_H_
#define CASSIE3D_VEC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CASSIE3D_NQ 21
#define CASSIE3D_NV 20
#define CASSIE3D_NU 10
/* one environment record in HBM: qpos[21] qvel[20] qacc_warmstart[20] ctrl[10] time niter nefc overflow pad */
#define CASSIE3D_STATE_STRIDE 80
#define CASSIE3D_OFF_QPOS 0
#define CASSIE3D_OFF_QVEL 21
#define CASSIE3D_OFF_WARMSTART 41
#define CASSIE3D_OFF_CTRL 61
#define CASSIE3D_OFF_TIME 71
#define CASSIE3D_OFF_NITER 72
#define CASSIE3D_OFF_NEFC 73
#define CASSIE3D_OFF_OVERFLOW 74 /* number of substeps of this environment in which the 64-row cap left contacts out (see below) */

typedef struct Cassie3dVec Cassie3dVec;

/* Environment switch read at create time (A/B and cross-check, not part of the contract): CASSIE3D_PAIR=1 runs the first pass with
 * two environments per wavefront (csrc/cassie3d_pair.hip) instead of one; same results to rounding, measured slower. */
int Cassie3dVecCreate(Cassie3dVec** out, int n_envs, int device);
void Cassie3dVecFree(Cassie3dVec* h);
const char* Cassie3dVecLastError(const Cassie3dVec* h);
int Cassie3dVecSetStream(Cassie3dVec* h, void* hip_stream);
int Cassie3dVecSynchronize(Cassie3dVec* h);
/* standing pose (qpos_dev == NULL) or the given [n][21] / [n][20] device arrays; then mj_forward (Cassie2d::Reset, :78-82) */
int Cassie3dVecReset(Cassie3dVec* h, const double* qpos_dev, const double* qvel_dev);
/* n_sub mj_steps with torques_dev [n][10] (pre-clamp motor commands, as Cassie2d::Step takes them) */
int Cassie3dVecStep(Cassie3dVec* h, const double* torques_dev, int n_sub);
double* Cassie3dVecStatePtr(Cassie3dVec* h); /* device [n][CASSIE3D_STATE_STRIDE] */
/* Event counters since create / the last reset of the counters (synchronises the stream):
 *   out4[0] env-substeps requested
 *   out4[1] env-substeps the 32-row kernel handed to the 64-row kernel
 *   out4[2] env-substeps in which more than 64 constraint rows were active (the model's maximum is 69: 6 connect + 12 limit rows
 *           + 17 contacts x 3) and the LAST contacts in MuJoCo order were left out for that substep.  The environment keeps
 *           stepping (no frozen environment); record slot CASSIE3D_OFF_OVERFLOW counts the same per environment.
 *   out4[3] env-substeps the lane-per-leg kernel (first tier) handed to the wavefront-per-environment kernels (row capacity of its per-lane LDS slots) */
int Cassie3dVecGetCounters(Cassie3dVec* h, uint64_t* out4);
int Cassie3dVecResetCounters(Cassie3dVec* h);
/* host-pointer conveniences (tests, small batches) */
int Cassie3dVecStepHost(Cassie3dVec* h, const double* torques, int n_sub);
int Cassie3dVecGetStateHost(Cassie3dVec* h, double* state /*[n][80]*/);
int Cassie3dVecSetStateHost(Cassie3dVec* h, const double* state /*[n][80]*/);
int Cassie3dVecDebugForwardHost(Cassie3dVec* h, const double* torques, double* dbg /*[n][CASSIE3D_DEBUG_STRIDE]*/);
#define CASSIE3D_DEBUG_STRIDE 1869
/* average time of `steps` back-to-back Cassie3dVecStep calls, HIP events on the handle's stream */
int Cassie3dVecTimeSteps(Cassie3dVec* h, const double* torques_dev, int n_sub, int steps, float* avg_ms);

#ifdef __cplusplus
}
#endif
#endif
