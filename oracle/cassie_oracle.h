/* cassie_oracle.h -- CPU ORACLE for the Cassie2d hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this library.  The product (libcassie2d.so built from cassierl_amd/csrc) never
 * links it and has no CPU fallback.
 *
 * PARITY UNPINNED: the arithmetic of the reference's physics step lives in MuJoCo Pro
 * 1.50 (closed binary, not under /root/reference) and RBDL / qpOASES 3.2.1 / Eigen
 * (un-vendored).  The reference has no tests or golden vectors for this path
 * (SURVEY.md section 4).  This file restates the published algorithms (MuJoCo
 * "Computation" chapter; RBDL's CRBA/RNEA/point-Jacobian definitions) for exactly
 * the feature subset cassie2d_stiff.xml uses, anchored on the reference's call sites:
 *   src/Cassie2d/Cassie2d.cpp:78-237, src/DynamicModel.cpp:237-367,
 *   src/DynamicState.cpp:45-91, src/OSC_RBDL.cpp:29-291, src/HelperFunctions.h:8-29,
 *   rllab/envs/cassie2d.py:78-225,263-331, rllab/envs/cassie_stand2d.py:86-137.
 * It is pinned only by (a) model known-answer values derived from the XML,
 * (b) golden vectors generated from the two importable reference Python modules,
 * (c) physics invariants (tests/test_oracle_physics.py).
 *
 * The same Part 1/2 source compiled with -DORC_CASSIE3D (liboracle3d.so, oracle/cassie3d_model.h) is the oracle of the Cassie3d
 * kernels: floating base (3 world translations + unit quaternion, body-frame angular velocity, mju_quatIntegrate) + 14 hinges
 * of model/cassie3d_stiff.xml.  The reference has no Cassie3d step at all (MJCF only), so that build is PARITY UNPINNED too; it is
 * pinned by numpy known-answer values of the MJCF (tests/golden/model3d_kat.json) and mechanics invariants (tests/test_oracle3d.py).
 * Only orc_create/free/reset/step_torque/forward/get_* and the test hooks exist in that build (qpos has 21 entries, qvel 20, ctrl 10).
 *
 * Deliberately written as a GENERAL 3-D articulated-body pipeline (22 bodies,
 * 3 constraint rows per contact / connect) so that it shares no formulation with the
 * planar HIP kernels it checks.
 */
#ifndef CASSIE_ORACLE_H_
#define CASSIE_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NV 13
#define ORC_NU 6
#define ORC_MAXCON 17
#define ORC_MAXEFC 72

typedef struct Oracle Oracle;

/* quirk flags; 0 = bit-faithful to the reference */
#define ORC_FIX_STALE_KIN 1    /* Q1/Q2: op-space state from CURRENT kinematics */
#define ORC_FIX_STALE_QSTATE 2 /* Q3: reward joint term from the current joint state */

Oracle* orc_create(void);                 /* Cassie2d::Cassie2d  (Cassie2d.cpp:29-72)  */
void orc_free(Oracle* o);
void orc_reset(Oracle* o, const double* qpos, const double* qvel); /* Cassie2d::Reset (:78-82) */
void orc_step_torque(Oracle* o, const double* torques);            /* Cassie2d::Step (:86-94) */
void orc_step_pd(Oracle* o, const double* angles);                 /* Cassie2d::StepPd (:96-117) */
void orc_step_jacobian(Oracle* o, const double* force6);           /* Cassie2d::StepJacobian (:119-177) */
void orc_step_osc(Oracle* o, const double* osc7);                  /* Cassie2d::StepOsc (:179-209) */
void orc_get_state(const Oracle* o, double* qpos, double* qvel);   /* GetGeneralState (:213-216) */
void orc_get_opstate(const Oracle* o, int flags, double* x18);     /* GetOperationalSpaceState (:218-237); layout of
                                                                      cassie2d_structs.py operational_state_to_array */

/* mj_forward / mj_step building blocks, exposed for the physics-invariant tests */
void orc_forward(Oracle* o);
void orc_set_state_raw(Oracle* o, const double* qpos, const double* qvel, const double* qacc_warmstart);
void orc_set_gravity(Oracle* o, double gz);
void orc_set_damping_scale(Oracle* o, double s);
void orc_set_contact_enabled(Oracle* o, int enabled);
/* Where closed MuJoCo Pro 1.50 (the reference's physics, src/Makefile:5,20) might differ from the published 2.x pipeline
 * this oracle restates.  0 = the 2.x semantics (default, what the HIP kernels implement); each bit
 * switches ONE candidate 1.50 behaviour on so that its effect on a trajectory can be measured (tests/test_oracle_mj_kats.py). */
#define ORC_ASSUME_IMP_SMOOTHSTEP 1   /* impedance sigmoid y = x^2 (3 - 2x) instead of the piecewise quadratic (midpoint .5, power 2) */
#define ORC_ASSUME_CONNECT_NORM_IMP 2 /* connect rows: one impedance from the norm of the 3-vector violation, not per row */
#define ORC_ASSUME_WS_STEP_ONLY 4     /* qacc_warmstart written by mj_step only, not by a bare mj_forward (Reset) */
void orc_set_assumptions(Oracle* o, int mask);
/* cap > 0: keep at most `cap` constraint rows by dropping the last contacts (mirrors the 64-row cap of the Cassie3d kernel) */
void orc_set_row_cap(Oracle* o, int cap);
/* terrain (N4): heights in metres, [nrow][ncol] row-major, row r at y = -size_y + r*2*size_y/(nrow-1), column c likewise in x;
 * the array is NOT copied.  NULL restores the flat floor. */
void orc_set_hfield(Oracle* o, const double* heights_m, int nrow, int ncol, double size_x, double size_y);
void orc_get_contacts(const Oracle* o, double* dist, double* pos3, double* frame9);
void orc_get_efc_extra(const Oracle* o, double* R, double* vel, double* diagApprox, double* b);
int orc_nefc(const Oracle* o);
int orc_ncon(const Oracle* o);
int orc_solver_niter(const Oracle* o);
void orc_get_mass_matrix(const Oracle* o, int sem, const double* qpos, double* M);
void orc_get_bias(const Oracle* o, int sem, const double* qpos, const double* qvel, double* bias);
void orc_batch_step_torque(Oracle** os, int n, const double* torques, int n_sub, int nthreads);
void orc_get_qacc(const Oracle* o, double* qacc);
void orc_get_warmstart(const Oracle* o, double* qacc_ws);
void orc_get_efc(const Oracle* o, double* J /*[nefc*13]*/, double* force, double* pos, double* aref, int* type);
void orc_get_ctrl(const Oracle* o, double* ctrl6);
void orc_set_ctrl(Oracle* o, const double* ctrl);
double orc_energy(const Oracle* o, double* kinetic, double* potential);
void orc_site_pos(const Oracle* o, int sem, const double* qpos, int site, double* p3);
void orc_get_model_consts(const Oracle* o, int sem, double* eq_anchor2 /*[2*3]*/, double* dof_invweight0 /*[13]*/,
                          double* body_invweight0_tran /*[22]*/, double* meaninertia);
/* DynamicState::UpdateDynamicState outputs (RBDL semantics) at the last setState */
void orc_get_dynamic_state(Oracle* o, double* M, double* bias, double* Bt, double* Jc, double* Jeq, double* JeqdotQdot);
/* last OSC QP (39 variables) and its KKT residual, for tests */
void orc_get_osc_qp(const Oracle* o, double* x39, double* kkt4);

/* ---- environment layer: restatement of rllab/envs/cassie2d.py (mode 0) and cassie_stand2d.py (mode 1) */
typedef struct OracleEnv OracleEnv;
#define ORC_CTRL_PD 0
#define ORC_CTRL_TORQUE 1
#define ORC_CTRL_OSC 2
OracleEnv* orc_env_create(int env_kind /*0 walk,1 stand*/, int control_mode, int flags,
                          const double* traj_qpos /*[n*13] or NULL*/, const double* traj_time, int traj_n);
void orc_env_free(OracleEnv* e);
Oracle* orc_env_oracle(OracleEnv* e);
void orc_env_reset(OracleEnv* e, double* obs26);                     /* cassie2d.py:78-95 */
void orc_env_step(OracleEnv* e, const double* action, int n_sub, double* obs26, double* reward, int* done); /* :97-225 */
double orc_env_time(const OracleEnv* e);

/* batch of independent envs stepped with OpenMP (CPU baseline for bench.py) */
void orc_envs_step(OracleEnv** envs, int n, const double* actions, int adim, int n_sub, int auto_reset,
                   double* obs, double* rew, unsigned char* done, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
