/*
 * mrs_oracle.h -- CPU restatement of the mrsgym step()/reset() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / timed CPU baseline.
 *
 * Parity status:
 *   - controller, mixer, force assembly, adjacency, history: PINNED against
 *     golden vectors produced by importing the reference's own Python
 *     (tools/gen_golden.py -> tests/golden/).
 *   - rigid-body integration + ground contact (pybullet, unpinned version,
 *     not present in the reference tree nor in this image): PARITY UNPINNED.
 *     Restated from the published bullet3 algorithm (btMultiBody ABA for a
 *     free base with zero-mass fixed links, stepPositionsMultiDof); every
 *     Bullet-side constant is a runtime field of OrcParams.
 *
 * Citations are file:line into the reference tree (mrsgym/...).
 */
#ifndef MRS_ORACLE_H
#define MRS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ACTION_TYPE strings of the reference that resolve to a method
 * (Quadcopter.py:26-65); "set_force" (README.md:68) has no implementation. */
enum {
    ORC_ACT_NONE = 0,        /* step(None): MRS.py:243-253 skips set_actions */
    ORC_ACT_SET_SPEEDS = 1,  /* Quadcopter.py:38-45   adim 4 */
    ORC_ACT_SET_CONTROL = 2, /* Quadcopter.py:26-34   adim 4 */
    ORC_ACT_TARGET_ACCEL = 3,/* Quadcopter.py:48-50   adim 3 */
    ORC_ACT_TARGET_VEL = 4,  /* Quadcopter.py:53-55   adim 3 */
    ORC_ACT_TARGET_POS = 5,  /* Quadcopter.py:58-60   adim 3 */
    ORC_ACT_TARGET_ORI = 6   /* Quadcopter.py:63-65   adim 3 */
};

typedef struct OrcParams {
    /* cf2x.urdf:5,11,12 via Quadcopter.read_attributes (Quadcopter.py:119-150) */
    double mass, arm, kf, km, thrust2weight;
    double ixx_file, iyy_file, izz_file; /* URDF file inertia: used by set_control only */
    double gnd_eff_coeff, prop_radius, drag_xy, drag_z, dw1, dw2, dw3;
    double prop_x[4], prop_y[4], prop_z[4]; /* cf2x.urdf:42,54,66,78 prop link COMs */
    double coll_radius, coll_half_len;      /* cf2x.urdf:34 */
    /* BulletSim.py:13-14 (world) and DefaultSim (BulletSim.py:52-57, frozen in the controller) */
    double gravity, dt, ctrl_gravity, ctrl_dt;
    /* [BULLET-KNOWLEDGE] dynamics constants, parity unpinned, re-pinnable */
    double inertia[3];   /* diag inertia Bullet recomputes from the collision shape */
    double lin_damp, ang_damp, max_coord_vel;
    int use_gyro;
    /* ground (plane.urdf:24 box 30x30x1 at origin => top z = +0.5) + contact model */
    double ground_z, friction, erp, contact_threshold;
    int solver_iters;
    int enable_contact;
    int pair_contact;    /* quad-quad contact as a sphere separation constraint (mrs_oracle.c:pair_contact) */
    int rest_shortcut;   /* flat bodies whose contact rows have a closed-form fixed point get it (contact_flat_closed_form); 0: sweeps only */
} OrcParams;

/* Per-agent controller memory (QuadControl.py lazily-created attributes). */
typedef struct OrcPid {
    double integral_pos_e[3]; /* QuadControl.py:41-44 */
    double d_vel_e[3];        /* :57,:62 */
    double integral_vel_e[3]; /* :60-61,:66 */
    double integral_ori_e[3]; /* :105,:108-110 */
    float last_vel_e[3];      /* :56,:64  (NaN = not yet created) */
    float last_target_vel[3]; /* :58-59,:65 (NaN = not yet created) */
} OrcPid;

void orc_params_default(OrcParams *p);
/* derived values of Quadcopter.calculate_parameters (Quadcopter.py:153-168):
 * out[0..6] = GravityForce, HoverRPM, MaxRPM, MaxThrust, MaxXYTorque, MaxZTorque, GroundEffectHClip */
void orc_derived(const OrcParams *p, double out[7]);
void orc_pid_init(OrcPid *pid, int n);

/* rotation helpers (scipy Rotation conventions used by Object.py:51-56,90-97) */
void orc_euler_to_quat(const double e[3], double q[4]);          /* from_euler('xyz') -> as_quat, xyzw */
void orc_quat_to_euler(const double q[4], double e[3]);          /* as_euler('xyz') */
void orc_quat_to_matrix(const double q[4], double R[9]);         /* row-major as_matrix */
void orc_matrix_to_euler_nearest(const double M[9], double e[3]);/* from_matrix(M).as_euler('xyz'), scipy 1.15 */

/* state read-back with the reference's fp32 truncation (Object.py:78-97) */
void orc_observe(const double pos[3], const double quat[4], const double vel[3], const double angvel[3],
                 float opos[3], float oeuler[3], float ovel[3], float oangvel[3], float omat[9]);

void orc_observe_batch(int n, const double *pos, const double *quat, const double *vel, const double *angvel,
                       float *opos, float *oeuler, float *ovel, float *oangvel);

/* controller entry points; all take already-observed fp32 values and fp32 actions,
 * return rpm[4] in double (QuadControl.py:35-127). */
void orc_pos_control(const OrcParams *p, OrcPid *s, const float pos[3], const float vel[3], const float ori[3],
                     const float angvel[3], const float target_pos[3], double rpm[4]);
void orc_vel_control(const OrcParams *p, OrcPid *s, const float vel[3], const float ori[3],
                     const float angvel[3], const float target_vel[3], double rpm[4]);
void orc_accel_control(const OrcParams *p, OrcPid *s, const double target_accel_in[3], const float ori[3],
                       const float angvel[3], double rpm[4]);
void orc_attitude_control(const OrcParams *p, OrcPid *s, const double target_ori[3], const float ori[3],
                          const float angvel[3], const double target_accel[3], double rpm[4]);
/* Quadcopter.nnlsRPM (Quadcopter.py:172-208); returns LH iteration count */
int orc_nnls_rpm(const OrcParams *p, double thrust, double tx, double ty, double tz, double rpm[4]);
/* Quadcopter.set_control glue in its fp32 arithmetic (Quadcopter.py:26-34) */
int orc_set_control(const OrcParams *p, const float control[4], double rpm[4]);

/* MRS.calc_A (MRS.py:117-124): pos fp32 [N][3] -> A fp32 [N][N] of 0/1 */
void orc_adjacency(int n, const float *pos, double comm_range, float *A);

void orc_adjacency_batch(int E, int n, const float *pos, double comm_range, float *A);

/* Reynolds flocking expert (examples/simulating_data/helper/Reynolds.py:80-110, Reynolds_Node.py:26-38; D >= 6,
 * K = 1): x_prev fp32 [E][N][D] = the PREVIOUS step's states, actions fp32 [E][N][3] */
void orc_reynolds(int E, int n, int D, const float *x_prev, float *actions);

/* MRS.generate_start_pos (MRS.py:127-154) on the sample stream laid out per round: cand fp32 [n_rounds][N][3], pos fp32 [N][3]
 * out; returns the rounds consumed or -1 (layout still colliding after n_rounds).  Pinned by tests/golden/F5b. */
int orc_spawn_from(int N, int n_rounds, const float *cand, double agent_radius, float *pos);

/* One env step for E envs x N agents (MRS.py:240-257 without the callbacks).
 * Arrays are [E][N][k] row-major.  actions may be NULL (ORC_ACT_NONE).
 * speeds_out (optional) receives the rotor speeds used this step, [E][N][4] double;
 * wrench_out (optional) the external body-frame wrench (force, torque) handed to the integrator, [E][N][6].
 * nthreads: OpenMP threads over envs (1 = scalar port). */
void orc_step(const OrcParams *p, int E, int N, double *pos, double *quat, double *vel, double *angvel,
              OrcPid *pid, const float *actions, int action_type, int adim, double *speeds_out, double *wrench_out,
              int nthreads);

/* CPU baseline of the whole per-step hot path (step + cat(pos,vel) slice + adjacency), env-parallel. */
void orc_step_full(const OrcParams *p, int E, int N, double *pos, double *quat, double *vel, double *angvel,
                   OrcPid *pid, const float *actions, int action_type, int adim, double comm_range,
                   float *obs, float *A, int nthreads);

/* Pieces of orc_step, exposed so the reference's own Python can be driven on top of
 * them by tools/gen_golden.py (fake-bullet harness) and for unit tests. */
void orc_integrate(const OrcParams *p, double pos[3], double quat[4], double vel[3], double angvel[3],
                   const double force_body[3], const double torque_body[3]);

/* test hook: the ground-contact rows of ONE body swept n_sweeps times, no closed forms, no early exits (v, w in/out: the
 * unconstrained velocities -> the constrained ones) */
void orc_contact_rows(const OrcParams *p, const double pos[3], const double quat[4], double v[3], double w[3], int n_sweeps);
void orc_contact_solve(const OrcParams *p, const double pos[3], const double quat[4], double v[3], double w[3]);

/* ---- geometry sensors (mrs_sensors.c; Object.py:100-174 against the analytic scene; PARITY UNPINNED, see there) ---- */
/* Object.raycast (Object.py:150-174) for agent `self` of ONE env: hit_obj -1 none / 0..N-1 quadcopter / N ground */
void orc_raycast(const OrcParams *p, int N, const double *pos, const double *quat, int self, const float *offset,
                 const float *dirs, int n_rays, int body, float range, int *hit_obj, float *pos_world, float *pos_body, float *dist);
/* Object.get_dist (Object.py:119-133) of agent `self` against quadcopters 0..N-1 and the ground (index N) */
void orc_closest(const OrcParams *p, int N, const double *pos, const double *quat, int self, double *dist, double *pself, double *pother);
/* all rows at once: D[N][N+1] */
void orc_proximity(const OrcParams *p, int N, const double *pos, const double *quat, double *D);

#ifdef __cplusplus
}
#endif
#endif
