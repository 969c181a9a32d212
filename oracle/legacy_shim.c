/* legacy_shim.c -- ORACLE-backed libcassie2d.so exporting the reference's ten C symbols
 * (src/Cassie2d/Cassie2d.cpp:15-27).  TEST INFRASTRUCTURE ONLY: it exists so that the
 * reference's own rllab/envs/cassie2d.py can be run (in the container that has
 * /root/reference, with stub rllab modules) to record golden (action -> obs, reward, done)
 * streams: reference Python arithmetic on oracle physics.  Never shipped, never loaded by
 * the cassierl_amd package. */
#include <stdbool.h>
#include <stdlib.h>
#include "cassie_oracle.h"

typedef struct { double torques[6]; } ControllerTorque;
typedef struct { double left_force[3], right_force[3]; } ControllerForce;
typedef struct { double body_xdd[2], left_xdd[2], right_xdd[2], pitch_add; } ControllerOsc;
typedef struct { double angles[6]; } ControllerPd;
typedef struct { double base_pos[3], base_vel[3], left_pos[5], left_vel[5], right_pos[5], right_vel[5]; } StateGeneral;
typedef struct { double body_x[3], body_xd[3], left_x[3], left_xd[3], right_x[3], right_xd[3]; } StateOperationalSpace;

void* Cassie2dInit(void) { return orc_create(); }
void Reset(void* c, StateGeneral* s) {
  double q[13], v[13];
  for (int i = 0; i < 3; i++) { q[i] = s->base_pos[i]; v[i] = s->base_vel[i]; }
  for (int i = 0; i < 5; i++) { q[3 + i] = s->left_pos[i]; v[3 + i] = s->left_vel[i]; q[8 + i] = s->right_pos[i]; v[8 + i] = s->right_vel[i]; }
  orc_reset((Oracle*)c, q, v);
}
void StepOsc(void* c, ControllerOsc* a) { orc_step_osc((Oracle*)c, a->body_xdd); }
void StepTorque(void* c, ControllerTorque* a) { orc_step_torque((Oracle*)c, a->torques); }
void StepJacobian(void* c, ControllerForce* a) { orc_step_jacobian((Oracle*)c, a->left_force); }
void StepPd(void* c, ControllerPd* a) { orc_step_pd((Oracle*)c, a->angles); }
void GetGeneralState(void* c, StateGeneral* s) {
  double q[13], v[13];
  orc_get_state((Oracle*)c, q, v);
  for (int i = 0; i < 3; i++) { s->base_pos[i] = q[i]; s->base_vel[i] = v[i]; }
  for (int i = 0; i < 5; i++) { s->left_pos[i] = q[3 + i]; s->left_vel[i] = v[3 + i]; s->right_pos[i] = q[8 + i]; s->right_vel[i] = v[8 + i]; }
}
void GetOperationalSpaceState(void* c, StateOperationalSpace* s) {
  double x[18];
  orc_get_opstate((Oracle*)c, 0, x);
  for (int i = 0; i < 2; i++) {
    s->body_x[i] = x[i]; s->body_xd[i] = x[3 + i]; s->left_x[i] = x[6 + i]; s->left_xd[i] = x[9 + i];
    s->right_x[i] = x[12 + i]; s->right_xd[i] = x[15 + i];
  }
  s->body_x[2] = x[2]; s->body_xd[2] = x[5];
}
void Display(void* c, bool d) { (void)c; (void)d; }
void Render(void* c) { (void)c; }
