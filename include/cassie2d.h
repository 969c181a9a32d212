/* cassie2d.h -- legacy C-ABI of libcassie2d.so, the drop-in boundary of the reference.
 *
 * These are exactly the ten C-linkage symbols the reference exports from
 * src/Cassie2d/Cassie2d.cpp:15-27 and binds through ctypes in
 * rllab/envs/cassie2d.py:22-50 (and cassie_stand2d.py:20-47), with the six POD structs of
 * src/Cassie2d/RobotInterface.h:14-50 (mirrored by rllab/envs/cassie2d_structs.py:5-51).
 * An unchanged cassie2d.py that finds this library at ../../bin/libcassie2d.so runs on the
 * MI355X path: every call is a batch-of-one launch of the same HIP kernels the batched
 * API (cassie_vec.h) uses.  There is no CPU implementation behind these symbols: if no
 * HIP device is available Cassie2dInit prints the reason and aborts (the reference also
 * exits the process on a failed init, Cassie2d.cpp:49-52).
 */
#ifndef CASSIE2D_H_
#define CASSIE2D_H_

#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RobotInterface.h:14-16 */
typedef struct { double torques[6]; } ControllerTorque;
/* RobotInterface.h:18-21 */
typedef struct { double left_force[3]; double right_force[3]; } ControllerForce;
/* RobotInterface.h:23-28 */
typedef struct { double body_xdd[2]; double left_xdd[2]; double right_xdd[2]; double pitch_add; } ControllerOsc;
/* RobotInterface.h:30-32 */
typedef struct { double angles[6]; } ControllerPd;
/* RobotInterface.h:34-41 */
typedef struct {
  double base_pos[3]; double base_vel[3];
  double left_pos[5]; double left_vel[5];
  double right_pos[5]; double right_vel[5];
} StateGeneral;
/* RobotInterface.h:43-50 */
typedef struct {
  double body_x[3]; double body_xd[3];
  double left_x[3]; double left_xd[3];
  double right_x[3]; double right_xd[3];
} StateOperationalSpace;

typedef struct Cassie2d Cassie2d;

Cassie2d* Cassie2dInit(void);                                              /* Cassie2d.cpp:17 */
void Reset(Cassie2d* cassie, StateGeneral* state);                         /* Cassie2d.cpp:18 */
void StepOsc(Cassie2d* cassie, ControllerOsc* action);                     /* Cassie2d.cpp:19 */
void StepTorque(Cassie2d* cassie, ControllerTorque* action);               /* Cassie2d.cpp:20 */
void StepJacobian(Cassie2d* cassie, ControllerForce* action);              /* Cassie2d.cpp:21 */
void StepPd(Cassie2d* cassie, ControllerPd* action);                       /* Cassie2d.cpp:22 */
void GetGeneralState(Cassie2d* cassie, StateGeneral* state);               /* Cassie2d.cpp:23 */
void GetOperationalSpaceState(Cassie2d* cassie, StateOperationalSpace* state); /* Cassie2d.cpp:24 */
void Display(Cassie2d* cassie, bool display);                              /* Cassie2d.cpp:25 -- recorded, no window */
void Render(Cassie2d* cassie);                                             /* Cassie2d.cpp:26 -- no-op (GUI out of scope) */

#ifdef __cplusplus
}
#endif
#endif
