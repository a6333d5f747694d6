/*
 * mrs_hip.h -- C-ABI of the MI355X-native mrsgym step()/reset() hot path.
 *
 * The reference (Acciorocketships/mrs-gym) has no FFI of its own: its lower boundary is the
 * set of in-process pybullet C-API calls issued per agent per step.  This library replaces that
 * boundary; each entry point names the reference interface (file:line under mrsgym/) it stands for.
 *
 *   plain pointers and sizes only; every buffer is DEVICE memory owned by the caller
 *   (PyTorch allocates it); the library borrows it for the duration of the call, launches
 *   asynchronously on the hipStream_t handed in as `void* stream`, never allocates or frees
 *   user-visible memory, never throws.  Return value: 0 = ok, >0 = hipError_t, <0 = MRS_E_*.
 *   A handle is bound to one device and is not thread-safe; distinct handles may be driven
 *   from distinct threads/processes (one per GPU).
 *
 * Layout (agent-major SoA; a = env*N + agent, T = E*N): plane c of a k-component quantity
 * lives at ptr[c*T + a], so that a wavefront's 64 lanes (64 consecutive quadcopters) read
 * 64 consecutive words.
 */
#ifndef MRS_HIP_H
#define MRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRS_ABI_VERSION 5

/* error codes (negative) */
#define MRS_E_ARG (-1)        /* bad argument (NULL, size, unsupported N) */
#define MRS_E_ACTION_TYPE (-2)/* unknown ACTION_TYPE: the reference raises AttributeError (Environment.py:92) */
#define MRS_E_NO_DEVICE (-3)

/* ACTION_TYPE (Quadcopter.py:26-65). README.md:68 "set_force" has no implementation upstream. */
#define MRS_ACT_NONE 0         /* step(None): MRS.py:243-253 */
#define MRS_ACT_SET_SPEEDS 1   /* Quadcopter.set_speeds       :38-45  adim 4 */
#define MRS_ACT_SET_CONTROL 2  /* Quadcopter.set_control      :26-34  adim 4 */
#define MRS_ACT_TARGET_ACCEL 3 /* Quadcopter.set_target_accel :48-50  adim 3 */
#define MRS_ACT_TARGET_VEL 4   /* Quadcopter.set_target_vel   :53-55  adim 3 */
#define MRS_ACT_TARGET_POS 5   /* Quadcopter.set_target_pos   :58-60  adim 3 */
#define MRS_ACT_TARGET_ORI 6   /* Quadcopter.set_target_ori   :63-65  adim 3 */

/* observation fields a fused state_fn may concatenate (Object.get_pos/get_vel/get_ori/get_angvel,
 * Object.py:78-97; all float32 as the reference's getters return them) */
#define MRS_OBS_POS 0     /* 3 */
#define MRS_OBS_VEL 1     /* 3 */
#define MRS_OBS_EULER 2   /* 3  get_ori(): extrinsic 'xyz' roll,pitch,yaw */
#define MRS_OBS_ANGVEL 3  /* 3  world frame */
#define MRS_OBS_QUAT 4    /* 4  xyzw */
#define MRS_OBS_MAX_FIELDS 8

/* orientation encodings accepted by mrs_set_state (Object.set_state, Object.py:51-56) */
#define MRS_ORI_EULER 0
#define MRS_ORI_QUAT 1
#define MRS_ORI_MATRIX 2

/* status bits written to MrsBuffers.status[env] */
#define MRS_STATUS_NAN_ACTION 1u /* MRS.py:247-248: the env's step was skipped, state untouched */
#define MRS_STATUS_SPAWN_FAIL 2u /* rejection sampling hit its iteration bound (MRS.py:137-153 would spin forever) */
#define MRS_STATUS_SPAWN_MORE 4u /* mrs_spawn_from ran out of candidate rounds: draw more and call again with resume = 1 */

/* Scene + model constants.  Filled by mrs_params_default from the values the
 * reference parses out of cf2x.urdf / plane.urdf (Quadcopter.read_attributes, Quadcopter.py:119-150)
 * and BulletSim.py:13-14; the [BULLET-KNOWLEDGE] block is re-pinnable at run time. */
typedef struct MrsParams {
    double mass, arm, kf, km, thrust2weight;
    double ixx_file, iyy_file, izz_file;
    double gnd_eff_coeff, prop_radius, drag_xy, drag_z, dw1, dw2, dw3;
    double prop_x[4], prop_y[4], prop_z[4];
    double coll_radius, coll_half_len;
    double gravity, dt;             /* world (BulletSim kwargs GRAVITY, DT) */
    double ctrl_gravity, ctrl_dt;   /* controller's frozen DefaultSim (Quadcopter.py:18, QuadControl.py:10) */
    /* [BULLET-KNOWLEDGE] */
    double inertia[3];
    double lin_damp, ang_damp, max_coord_vel;
    int32_t use_gyro;
    int32_t enable_contact;
    double ground_z, friction, erp, contact_threshold;
    /* at most this many sequential-impulse sweeps over the ground-contact rows per body and step (default 10; they stop earlier on
     * convergence or stagnation).  Accuracy against a converged solve: tests/golden/F6c + tests/test_oracle_golden.py (99 % of the
     * body-steps of tumbling bodies within 9e-4 m/s at 10, 4e-3 at 8, 2e-2 at 6); pybullet's own default is 50. */
    int32_t solver_iters;
    /* 1: the attitude controller rebuilds its rotation matrix from the FLOAT32-rounded Euler angles exactly as
     * from_euler(get_ori()) does (Object.py:97 -> QuadControl.py:99); 0 (default): from the unrounded angles of the same
     * float32 quaternion read-back -- they differ by <= 2^-24 relative per angle, the reference's own read-back noise
     * (docs/experiments.md section 4, deviation 7); saves ~130 float64 instructions per agent-step. */
    int32_t round_euler_readback;
    /* 1 (default): quad-quad contact -- every quadcopter a sphere of coll_radius; a pair within contact_threshold gets,
     * per body, half of the normal velocity change that closes the gap this step / pushes the overlap out with erp (one
     * pass, no friction, no torque; the build's own model).  Needs enable_contact.  The step finds the
     * envs that have such a pair from flags its adjacency pass left for the positions it wrote: a caller that writes
     * MrsBuffers.pos itself rather than through mrs_set_state* / mrs_spawn* calls mrs_observe or mrs_adjacency afterwards
     * (either refreshes the flags), or the first step after the write may miss a new contact. */
    int32_t pair_contact;
    /* 1 (default): a body lying FLAT on the ground -- |R20|, |R21| < 1e-6: all four rim points active with one common gap -- whose
     * rows have a closed-form fixed point is finished in its own lane (contact_at_rest) instead of being listed for the
     * sequential-impulse solve: lifting (no rim point's right-hand side is positive: no impulse at all, exact) or sticking (the
     * contact can hold it: v = (0, 0, u), w = 0; a conservative yaw-free feasibility test of the normal and friction impulses).
     * A body at rest is the special case w = 0, v_xy = 0 (round 3).  Within 3e-6 m/s of what 400 float64 sweeps converge to on
     * 70 000 captured contact problems.  0: every body near the ground goes through the sweeps, as in the
     * oracle (bench.py's `literal` leg).  ABI 5 (was a reserved word). */
    int32_t rest_shortcut;
} MrsParams;

/* Device buffers of one swarm shard (all borrowed).  Optional members may be NULL. */
typedef struct MrsBuffers {
    double *pos;      /* [3][T]  world position                                    */
    double *quat;     /* [4][T]  body->world orientation, xyzw                     */
    double *vel;      /* [3][T]  world linear velocity                             */
    double *angvel;   /* [3][T]  world angular velocity                            */
    float *pid;       /* [5][T][4] controller memory (QuadControl.py:41-110): five planes of 16-byte records (16-byte
                         aligned; one load and one store instruction per record and step), float32:
                         0: integral_pos_e xyz, pad      1: d_vel_e xyz, integral_vel_e x
                         2: integral_vel_e yz, last_vel_e xy      3: last_vel_e z, last_target_vel xyz
                         4: integral_ori_e xyz, pad      (last_*: NaN = attribute not created yet).
                         The arithmetic is float64 in registers; only the step-to-step carry is float32
                         (measured effect on 1000-step trajectories: < 1e-6, tests/test_gpu_parity.py). */
    float *obs;       /* (E,N,D) newest observation slice, row-major, or NULL      */
    uint64_t *adj;    /* (E,N,W) bit-packed newest adjacency rows, W = ceil(N/64), or NULL */
    float *rpm;       /* [4][T]  rotor speeds used by the last step (optional)     */
    uint32_t *status; /* [E]     MRS_STATUS_* bits, OR-ed in (optional)            */
    float *adj_dense; /* (E,N,N) newest adjacency as the float32 0/1 matrices MRS.calc_A returns (MRS.py:117-124), or NULL.
                         Needs adj as well.  mrs_step / mrs_adjacency write it from the kernel that builds the rows where an
                         env is whole wavefronts (N = 64, 128, 192, 256) and the pointer is 16-byte aligned -- 4 N^2 bytes per
                         env on top of the step's own traffic instead of a second kernel that reads the packed rows back --
                         and through mrs_adjacency_expand behind the step otherwise (ABI 4) */
} MrsBuffers;

typedef struct MrsHandle MrsHandle;

int mrs_abi_version(void);
const char *mrs_last_error(void);

/* cf2x + plane constants, Bullet defaults. */
int mrs_params_default(MrsParams *out);
/* Quadcopter.calculate_parameters (Quadcopter.py:153-168):
 * GravityForce, HoverRPM, MaxRPM, MaxThrust, MaxXYTorque, MaxZTorque, GroundEffectHClip */
int mrs_params_derived(const MrsParams *p, double out[7]);

/* BulletSim.__init__/setup + env_generator('simple') (BulletSim.py:11-35, EnvCreator.py:7-13):
 * binds a handle to `device` for E envs of N quadcopters over a ground box. */
int mrs_create(const MrsParams *params, int n_envs, int n_agents, int device, MrsHandle **out);
void mrs_destroy(MrsHandle *h);
int mrs_set_params(MrsHandle *h, const MrsParams *params);

/* bytes/words the caller must allocate for MrsBuffers members */
int mrs_adj_words(int n_agents);               /* W */
int mrs_obs_dim(const int32_t *fields, int n_fields); /* D, or <0 */

/* QuadControl lazily-created attributes back to "not created" (a fresh Quadcopter, Quadcopter.py:14-19).
 * env_mask: NULL = all envs, else E bytes (device), non-zero = apply. */
int mrs_pid_reset(MrsHandle *h, const MrsBuffers *b, const uint8_t *env_mask, void *stream);

/* Environment.set_state -> Object.set_state (Environment.py:97-103, Object.py:42-65).
 * pos/vel/angvel: (E,N,3) float32 row-major device arrays or NULL = keep (MRS.set semantics);
 * ori: (E,N,3) euler 'xyz' | (E,N,4) quat xyzw | (E,N,9) matrix, by ori_kind, or NULL = keep. */
int mrs_set_state(MrsHandle *h, const MrsBuffers *b, const float *pos, const float *ori, int ori_kind,
                  const float *vel, const float *angvel, const uint8_t *env_mask, void *stream);
/* float64 variant of the same (bit-exact state restore / sharding tests) */
int mrs_set_state_f64(MrsHandle *h, const MrsBuffers *b, const double *pos, const double *quat,
                      const double *vel, const double *angvel, const uint8_t *env_mask, void *stream);

/* MRS.step hot path (MRS.py:240-257): Environment.set_actions (controller -> rotor forces,
 * Quadcopter.dynamics ground effect / drag / downwash), BulletSim.step_sim, then the newest
 * observation slice (b->obs, fields as given) and the newest adjacency rows (b->adj, if
 * comm_range is not NaN).  actions: (E,N,adim) float32 row-major, NULL only for MRS_ACT_NONE.
 * comm_range = +inf reproduces MRS.py:118-119 (ones - eye). */
int mrs_step(MrsHandle *h, const MrsBuffers *b, const float *actions, int action_type,
             const int32_t *obs_fields, int n_obs_fields, double comm_range, void *stream);

/* n_substeps consecutive MRS.step calls from ONE host call = n_substeps kernel launches queued back to back on the stream
 * (the `n_substeps` of SURVEY.md 8b; frame-skip / on-device rollouts; an in-kernel substep loop was built and spills):
 * substep s uses the action batch at actions + s * action_stride (0: the same actions are held, as gym wrappers that
 * repeat an action do) and writes its observation slice at b->obs + s * obs_stride floats and its adjacency rows at
 * b->adj + s * adj_stride words (strides may be negative: a history ring that grows downwards).  Results are those of
 * n_substeps mrs_step calls, bit for bit; the launches are queued back to back from this one call.
 * Host callbacks (reward_fn, done_fn, ...) cannot run between substeps: that is the caller's contract. */
int mrs_step_n(MrsHandle *h, const MrsBuffers *b, const float *actions, int action_type, int n_substeps, int64_t action_stride,
               const int32_t *obs_fields, int n_obs_fields, double comm_range, int64_t obs_stride, int64_t adj_stride, void *stream);

/* Environment.get_X fast path for a state_fn that concatenates getters (Environment.py:84-87),
 * without stepping: used by reset()/set() -> calc_Xk (MRS.py:190, :203). */
int mrs_observe(MrsHandle *h, const MrsBuffers *b, const int32_t *obs_fields, int n_obs_fields, void *stream);

/* MRS.calc_A (MRS.py:117-124) without stepping; writes b->adj. */
int mrs_adjacency(MrsHandle *h, const MrsBuffers *b, double comm_range, void *stream);
/* bit-packed rows -> the float32 0/1 matrices the reference returns: packed (M,N,W) -> dense (M,N,N) */
int mrs_adjacency_expand(MrsHandle *h, const uint64_t *packed, float *dense, int n_matrices, void *stream);

/* MRS.generate_start_pos / generate_start_ori with the default spawn distribution
 * (MRS.py:69-78, :127-161): xy ~ N(0,1) pulled into the unit disc, z ~ U[1,3], greedy re-sampling
 * until all pairwise distances >= 2*agent_radius; yaw ~ U[ori_lo, ori_hi] per axis.
 * Counter-based RNG keyed by (seed, global env index) so shards reproduce the single-GPU stream.
 * Writes pos/quat, zeroes vel/angvel. */
int mrs_spawn(MrsHandle *h, const MrsBuffers *b, uint64_t seed, int64_t env_index_base, double agent_radius,
              const float ori_lo[3], const float ori_hi[3], int max_rounds, const uint8_t *env_mask, void *stream);

/* MRS.generate_start_pos (MRS.py:127-154) for a USER distribution (README.md:71-74: START_POS as a torch distribution,
 * per-agent (3,) or joint (N,3) samples): the caller draws the samples -- candidates (E, n_rounds, N, 3) float32, round
 * r = what every agent would receive if re-sampled in round r (per-agent and joint draws look the same here) -- and the
 * greedy rejection of the most-colliding agents (:137-151, torch.mode = lowest index among the most frequent) runs on
 * the device, one workgroup per env.  Writes positions only.  Envs that use up their rounds without a collision-free
 * layout are flagged MRS_STATUS_SPAWN_MORE; call again with fresh candidates, resume = 1 and those envs in env_mask. */
int mrs_spawn_from(MrsHandle *h, const MrsBuffers *b, const float *candidates, int n_rounds, int resume, double agent_radius,
                   const uint8_t *env_mask, void *stream);

/* A caller of the path, fused: the Reynolds flocking expert the reference's data generator
 * drives the env with -- examples/simulating_data/helper/Reynolds.py:80-110 (forward_batch) with the controller
 * of helper/Reynolds_Node.py:26-38, in the configuration its own caller uses (gen_data.py:33: K = 1).
 * x_prev: (E,N,D) float32 row-major, D >= 6 = cat(pos, vel, ...) of the PREVIOUS step (history slot 1 of the
 * K_HOPS ring can be passed as is); actions: (E,N,3) float32 target velocities.  forward_batch ignores the
 * adjacency it is handed (it substitutes ones - eye, Reynolds.py:83), so none is taken here. */
int mrs_reynolds(MrsHandle *h, const float *x_prev, int D, float *actions, void *stream);

/* ---- geometry sensors: Object.py:100-174 against the analytic scene of env_generator('simple')
 * (ground box of plane.urdf:24 + one collision cylinder per quadcopter, cf2x.urdf:34).  Replaces p.rayTestBatch,
 * p.getClosestPoints, p.getContactPoints, p.getOverlappingObjects.  Not part of mrs_step: call when a callback asks. */

/* Object.raycast (Object.py:150-174), for EVERY quadcopter of every env at once: each casts the same n_rays rays.
 * offset, directions: (n_rays,3) float32 device arrays (directions are scaled by `range` as :151 does, on a copy);
 * body != 0: both are body-frame vectors (:158-160).  Outputs, row-major device arrays:
 *   hit_obj   (E,N,R) int32  -1 = miss, 0..N-1 = quadcopter of the same env, N = ground   ("object", :164)
 *   pos_world (E,N,R,3)      hit position minus the rotated offset                         ("pos world", :165)
 *   pos_body  (E,N,R,3)      R^T pos_world - R^T pos                                       ("pos", :166)
 *   dist      (E,N,R)        |pos_body|                                                    ("dist", :171); misses are zeros */
int mrs_raycast(MrsHandle *h, const MrsBuffers *b, const float *offset, const float *directions, int n_rays, int body,
                float range, int32_t *hit_obj, float *pos_world, float *pos_body, float *dist, void *stream);

/* Object.get_dist (Object.py:119-133) for every ordered pair: dist (E,N,N+1) float32, column j < N = quadcopter j of
 * the same env (0 on the diagonal, 0 when the hulls overlap), column N = the ground (signed: negative when sunk in);
 * +inf where the bodies are further apart than max_dist (the reference returns an empty result there, :121-122).
 * p_self / p_other (optional, (E,N,N+1,3) float32): "closest pos self" / "closest pos other" in world coordinates.
 * collision() (:136-137), get_contact_points (:100-116) and get_closest_objects (:140-147) are thresholds on this.
 * Accuracy: float64 GJK stopped at a relative gap of 1e-9 (the outputs are float32: half an ulp is 6e-8 relative); tested against
 * the oracle's 1e-14 iteration at 2e-6 absolute. */
int mrs_proximity(MrsHandle *h, const MrsBuffers *b, double max_dist, float *dist, float *p_self, float *p_other, void *stream);

/* ---- flocking metrics of the reference's analytics: examples/simulating_data/helper/
 * MRSAnalytics.py:36-101 over M frames X (M,N,D) float32, D >= 6 = cat(pos, vel, ...) (one frame = one step of one
 * episode; Trainer.get_episodes' NaN padding propagates as in torch).  Every output is optional (NULL = skip):
 *   separation (M,N)      distance to the closest other agent, coincident agents masked (:61-72)
 *   cohesion (M), cohesion_noleader (M)   largest pairwise distance, with / without agent 0 (:82-93)
 *   dist_to_leader (M)    |mean(pos[1:]) - pos[0]| (:95-101)
 *   vel_stddev (M)        sqrt(det(sum_i (v_i - vbar)(v_i - vbar)^T)) (:44-53)
 * No handle: the call touches no env state; runs on the current device of `stream`. */
int mrs_flock_metrics(const float *X, int n_frames, int n_agents, int D, float *separation, float *cohesion,
                      float *cohesion_noleader, float *dist_to_leader, float *vel_stddev, void *stream);

#ifdef __cplusplus
}
#endif
#endif
