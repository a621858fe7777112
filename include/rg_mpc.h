/*
 * rg_mpc.h -- C-ABI of the MI355X-native batched convex-MPC gait controller.
 *
 * Drop-in boundary for ONE path of nicrusso7/robot-gym: everything
 * MPCController.get_action() does per control tick
 * (reference robot_gym/controllers/mpc/mpc_controller.py:102-106), batched over B robots.
 * The reference has no FFI for this path (it is pure Python calling the third-party
 * `mpc_controller` / `mpc_osqp` modules, mpc_controller.py:6-7); the entry points below are
 * what a maintainer's ctypes binding would load.
 *
 * Conventions
 *   - return 0 on success, a negative rg_mpc_status otherwise; nothing throws across the ABI;
 *     rg_mpc_last_error() gives the text of the last failure on a handle (or of create()).
 *   - the CALLER owns every I/O buffer (device memory, e.g. torch-ROCm tensors passed as
 *     data_ptr()); the library owns only per-robot persistent controller state.
 *   - all work is enqueued on the hipStream_t passed in (NULL = default stream); no hidden
 *     synchronisation in rg_mpc_step / rg_mpc_set_command / rg_mpc_hybrid_to_torque.
 *   - one handle per (device, stream); calls on one handle are not thread-safe.  Every call leaves the calling
 *     thread's current HIP device as it found it (a process driving several GPUs keeps one handle per device).
 *   - inputs are float32 struct-of-arrays, component-major:  x[c*B + b]  (lane = robot loads
 *     coalesce); outputs are row-major per robot:  action[b*60 + k].
 */
#ifndef RG_MPC_H
#define RG_MPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RG_MPC_ABI_VERSION 5
#define RG_MPC_MAX_HORIZON 20
#define RG_MPC_NUM_LEGS 4
#define RG_MPC_NUM_MOTORS 12
#define RG_MPC_ACTION_DIM 60 /* reference model/robots/simple_motor.py:15,130 */

typedef enum {
  RG_MPC_OK = 0,
  RG_MPC_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
  RG_MPC_ERR_HIP = -2,       /* HIP runtime error (text in last_error) */
  RG_MPC_ERR_NO_DEVICE = -3, /* no usable GPU */
  RG_MPC_ERR_ALLOC = -4
} rg_mpc_status;

/* gait_generator.LegState of the upstream library; the reference uses the enum at
 * model/robots/ghost/ctrl_constants.py:32-37 */
enum { RG_LEG_SWING = 0, RG_LEG_STANCE = 1, RG_LEG_EARLY_CONTACT = 2, RG_LEG_LOSE_CONTACT = 3 };

/* RG_SOLVER_ADMM: friction-cone ADMM only (iteration cap admm_iters); unconverged robots are counted as failures.
 * RG_SOLVER_ACTIVE_SET: exact dual active-set method only (horizon 10), inside the QP launch: the force-space body for one /
 *   two stance legs, the wrench-space exact body for three / four; the exact re-solve launch takes only a working set that
 *   outgrew its body or a degenerate stance -- and, under a contact schedule, every robot (that plan has no QP launch).
 * RG_SOLVER_AUTO: ADMM first; robots that have not converged after admm_iters iterations are
 *   re-solved exactly by the active-set kernel (second launch over a retry list).
 * RG_SOLVER_HYBRID (default): what the reference's default solver computes -- the exact minimiser (upstream qpOASES, an
 *   online active-set method; mpc_controller.py:47-56 passes no solver) -- for the robots where it is also the cheaper way:
 *   one and two stance legs (every robot of a trot, 60 variables, one or two active constraints at the optimum) are solved
 *   by a one-wave dual active-set body warm-started from the robot's working set of the previous tick (warm_start); three
 *   and four legs keep the wrench-space ADMM with the exact re-solve behind it, as under RG_SOLVER_AUTO.  At horizon 20 the
 *   same split (the exact body on 256 lanes, the solve on one wave of them; three and four legs on the schedule body with a
 *   constant schedule).  With contact_lookahead it is RG_SOLVER_AUTO. */
enum { RG_SOLVER_ADMM = 0, RG_SOLVER_ACTIVE_SET = 1, RG_SOLVER_AUTO = 2, RG_SOLVER_HYBRID = 3 };

/* Everything MPCController._setup_controller wires (mpc_controller.py:28-66) plus the
 * upstream module defaults it does not override, as explicit fields. */
typedef struct {
  int32_t abi_version;      /* RG_MPC_ABI_VERSION */
  int32_t horizon;          /* upstream _PLANNING_HORIZON_STEPS = 10; accepted: 10 and 20 (anything else is rejected by create) */
  double dt_plan;           /* upstream _PLANNING_TIMESTEP = 0.025 */
  double mass;              /* MPC_BODY_MASS      ghost/ctrl_constants.py:8 */
  double inertia[9];        /* MPC_BODY_INERTIA   ghost/ctrl_constants.py:9 (row-major) */
  double body_height;       /* MPC_BODY_HEIGHT    ghost/ctrl_constants.py:10 */
  double weights[13];       /* upstream _MPC_WEIGHTS (rpy, xyz, omega, v, g) */
  double alpha;             /* 1e-5 ; P = 2 B'WB + alpha I */
  double mu[4];             /* friction coefficient per leg (FR, FL, RR, RL); upstream's foot_friction_coeffs, 0.45 x 4 in the reference.
                               Unequal values select kernel instantiations that carry the coefficient per lane */
  double fz_max_scale;      /* 10  : fz_max = scale * m * g */
  double fz_min_scale;      /* 0.1 : fz_min = scale * m * g */
  double gravity;           /* 9.8 */
  double stance_duration[4];/* STANCE_DURATION_SECONDS ctrl_constants.py:13 */
  double duty_factor[4];    /* DUTY_FACTOR             ctrl_constants.py:28 */
  double init_phase[4];     /* INIT_PHASE_FULL_CYCLE   ctrl_constants.py:29 */
  int32_t init_state[4];    /* INIT_LEG_STATE          ctrl_constants.py:32-37 */
  double contact_phase_thresh; /* 0.1 */
  int32_t window;           /* COMVelocityEstimator window_size=20, mpc_controller.py:36 */
  int32_t kin_mode;         /* 0: foot_pos + jac supplied by caller; 1: computed from q on device */
  double foot_clearance;    /* 0.01, mpc_controller.py:45 */
  double swing_kp[3];       /* Raibert gain 0.03 */
  double max_clearance;     /* 0.1 swing apex */
  double hip[12];           /* DEFAULT_HIP_POSITIONS ghost/constants.py:31-36 */
  double motor_kp[12];      /* MOTOR_POSITION_GAINS ghost/motor_constants.py:13 */
  double motor_kd[12];      /* MOTOR_VELOCITY_GAINS :15 */
  double motor_dir[12];     /* MOTOR_DIRECTION :11 */
  double motor_off[12];     /* MOTOR_OFFSET :9 */
  /* URDF leg chains (util/pybullet_data/robots/ghost.urdf), [leg][joint][xyz] */
  double jxyz[36];
  double jrpy[36];
  double jaxis[36];
  double toe_xyz[12];
  double toe_com[12];
  double base_com[3];
  int32_t ik_iters;         /* fixed damped-Newton iteration count (8; converged to 1e-14 after 6 on swing-size moves) */
  int32_t solver;           /* RG_SOLVER_* (default RG_SOLVER_HYBRID) */
  double ik_damping;        /* lambda^2 */
  double ik_max_step;       /* rad per iteration */
  /* friction-cone ADMM */
  int32_t admm_iters;       /* ADMM iteration cap over both stages (450); exactly this many iterations when admm_tol == 0 */
  int32_t reserved0;        /* must be 0 */
  double admm_rho;          /* 1e-4 */
  double admm_relax;        /* 1.8 */
  double admm_tol;          /* stop when no z entry moved more than admm_tol * m * g over the last
                               admm_check iterations (1e-6); 0 = fixed iteration count */
  int32_t admm_check;       /* convergence check period (5) */
  int32_t contact_lookahead;/* EXTENSION (not in upstream, default 0): horizon step k >= 1 uses a per-step contact schedule
                               instead of holding the current contacts: the caller's
                               rg_mpc_state_ptrs.contact_sched when given, else the open-loop desired contact state at
                               t + k*dt_plan */
  int32_t warm_start;       /* 1: start ADMM from the robot's previous-tick (z, y) (stored as float32) while its contact set is
                               unchanged, as upstream's OSQP path does; results stay within admm_tol of the cold solve; the
                               exact body starts from the robot's previous working set (the result is the same minimiser).
                               0: cold start every tick.  (Not used by the contact-schedule body.) */
  int32_t reserved2;        /* must be 0 */
  /* second ADMM stage: robots not converged after admm_switch iterations are re-factorised with
   * admm_rho2 and continue from their iterate up to admm_iters.  admm_rho2 = 0 or admm_switch >= admm_iters: single stage. */
  double admm_rho2;         /* 5e-4 */
  int32_t admm_switch;      /* 150 */
  int32_t admm_accel;       /* 80: from this iteration on, a convergence vote may extrapolate the iterate (z, y) along its
                               dominant mode (Aitken step: consecutive vote-to-vote displacements parallel, shrinking by a
                               steady ratio r -> jump by r / (1 - r) of the last displacement).  Cuts the few crawling robots
                               per tick that bound the launch by a third to a half; a robot that has jumped has to pass
                               stricter stopping tests.  Not applied to single-leg stance.  0 = off */
  double admm_extrap;       /* 1.5: third convergence condition -- the distance still to go estimated from the shrink rate of
                               the movement per vote window, m r / (1 - r), must be below admm_extrap * admm_tol * m * g
                               (stops crawling robots from passing the "stopped moving" test early); 0 = off */
  /* thresholds of the dominant-mode extrapolation (tuned constants, like every other one an explicit field):
   * a vote jumps when consecutive window displacements d1, d0 satisfy cos^2(d1, d0) > accel_cos2 and the shrink
   * ratio r = <d1,d0>/<d0,d0> lies in (accel_rmin, accel_rmax); the rate that guards the stopping test of a robot
   * that has jumped is capped at accel_rate_cap */
  double accel_cos2;        /* 0.9 */
  double accel_rmax;        /* 0.98 */
  double accel_rmin;        /* 0.5 */
  double accel_rate_cap;    /* 0.999 */
  /* always-on audit lane: on average audit_k pseudo-randomly chosen robots per tick whose ADMM solve CONVERGED are re-solved
   * by the exact active-set bodies (on a library-owned low-priority side stream, overlapped with the following ticks; outputs
   * are never touched) and the joint torques of the two solutions are compared: rg_mpc_audit_stats.  The picks are made on
   * the first tick and then on every RG_MPC_AUDIT_PERIOD-th one, RG_MPC_AUDIT_PERIOD x audit_k of them.  0 = off, at most 16 */
  int32_t audit_k;          /* 8 */
  int32_t reserved3;        /* must be 0 */
  double audit_tol;         /* 1e-4: per-robot torque error max_j |dtau_j| / max(max_j |tau_j|, 1 N m) counted as over tolerance */
  double admm_rho34_scale;  /* first-stage rho of the wrench-space ADMM body (three and four stance legs, horizon 10) = admm_rho x this.
                               Its iteration count falls with rho at every percentile (admm_rho x 0.5: mean 54 -> ~43), where the
                               force-space body's tail grows; with the exact body taking the one- and two-leg robots
                               (RG_SOLVER_HYBRID) these robots are the launch's longest jobs */
  double admm_rho_sched_scale; /* the same for the schedule body (any contact schedule; three and four legs at horizon 20), which
                               iterates in wrench space too */
  int32_t lane_grid;        /* lanes per robot in the QP launch of the default plan at horizon 10: 1 = one 64-lane wave (8 x 8 lanes,
                               8 x 8 register tiles); 2 = 256 lanes (16 x 16 lanes, 4 x 4 tiles): a robot's dependent chain is
                               ~a third shorter, its wave-slot time larger -- for batches that leave most of the chip idle;
                               0 = chosen by rg_mpc_create from the batch (2 up to RG_MPC_WIDE_BATCH robots).  Ignored by
                               the other plans and at horizon 20 (always 256 lanes) */
  /* Recall-sensitive conventions.  The MPC arithmetic of the reference lives in an un-vendored package (motion_imitation
   * 0.0.5, reference requirements.txt:8) and is restated here from recall.  Where the recall has two
   * plausible readings, the reading is a switch: 0 = what this library and its oracle implement by default, 1 = the other
   * one, so that a maintainer who can run the upstream package (tests/golden/make_upstream_golden.py) flips a convention
   * without touching a kernel. */
  int32_t conv_alpha_doubled;   /* 0: P = 2 B'WB + alpha I ; 1: P = 2 (B'WB + alpha I), i.e. the regulariser inside the factor 2 */
  int32_t conv_feet_rotation;   /* lever arms r_i = R foot_i with yaw zeroed: 0: R = Rx(roll) Ry(pitch) ; 1: R = Ry(pitch) Rx(roll), the
                                   order the body inertia is rotated with */
  int32_t conv_com_height;      /* CoM height from the contact feet: 0: |mean z| ; 1: mean |z| */
  int32_t conv_first_latch;     /* 0: the first update after a reset does NOT latch a swing foot on a STANCE->SWING edge ; 1: it does */
  int32_t conv_window_divide;   /* velocity filter while its window fills: 0: divide by the window size ; 1: by the samples held */
  int32_t conv_friction_rows;   /* the four coefficients mu[0..3] when they DIFFER: 0: mu[l] is leg l's (upstream's name, foot_friction_coeffs) ;
                                   1: mu[t] belongs to cone row t (-fx, +fx, -fy, +fy) of EVERY block -- how upstream's UpdateConstraintsMatrix
                                   is recalled to use them.  Equal coefficients (every shipped robot: 0.45 x 4) make the two the same.
                                   1 with unequal coefficients needs solver = RG_SOLVER_ACTIVE_SET (the ADMM bodies project onto a
                                   symmetric pyramid).  (This field was reserved4, must-be-0, in earlier builds of ABI 5.) */
} rg_mpc_config;
#define RG_MPC_WIDE_BATCH 1024

/* Device pointers, float32 / int32, component-major [c][B].  Reference getters named per
 * field (model/robots/robot.py). */
typedef struct {
  const float *rpy;       /* [3][B] GetBaseRollPitchYaw            robot.py:79-86 */
  const float *rpy_rate;  /* [3][B] GetBaseRollPitchYawRate (body) robot.py:205-213 */
  const float *v_world;   /* [3][B] GetBaseVelocity                robot.py:172-178 */
  const float *quat;      /* [4][B] GetTrueBaseOrientation x,y,z,w robot.py:180-183 */
  const float *q;         /* [12][B] GetMotorAngles                robot.py:231-236 */
  const float *foot_pos;  /* [12][B] GetFootPositionsInBaseFrame   robot.py:389-397 (kin_mode 0; may be NULL in kin_mode 1) */
  const float *jac;       /* [36][B] per leg d foot_i/d joint_j, index leg*9+i*3+j:
                             columns 6+joint of calculateJacobian, controllers/mpc/kinematics.py:25-27,47-51 (kin_mode 0) */
  const int32_t *contact; /* [4][B] GetFootContacts                robot.py:215-229 */
  const float *cmd;       /* [3][B] (vx,vy,wz) AFTER the robot offsets of mpc_controller.py:90-95; NULL = use rg_mpc_set_command */
  const int32_t *contact_sched; /* [4][B] optional, contact_lookahead only: bit k (1 <= k < horizon) of word [leg][b] = that leg
                             is planned to be in contact at horizon step k (terrain / measured-contact knowledge of the caller;
                             reference sources of contact variation: model/world/terrain.py:33-93, robot.py:215-229).  Bit 0 is
                             ignored: step 0 is always the controller's own contact decision.  NULL = open-loop gait schedule */
  const double *t_robot;  /* [B] optional per-robot clock values (float64): robot b is stepped at t_robot[b] instead of the
                             scalar t of rg_mpc_step -- every reference controller reads its OWN simulation's clock
                             (controllers/controller.py:6-8, core/simulation.py:141-142), and sub-envs of a vectorised env that
                             were reset at different moments are at different clock values.  NULL = all robots at t */
} rg_mpc_state_ptrs;

typedef struct {
  float *action;           /* [B][60] hybrid command (q*,kp,qd*,kd,tau)x12 -- required */
  float *grf;              /* [B][12] first-step contact forces (optional, may be NULL) */
  float *tau_stance;       /* [B][12] J' f for all 12 joints (optional) */
  int32_t *leg_state;      /* [B][4] (optional) */
  int32_t *desired_state;  /* [B][4] (optional) */
  float *phase;            /* [B][4] normalized phase (optional) */
  float *foot_target;      /* [B][12] swing trajectory point (optional) */
  float *v_body;           /* [B][3] filtered body-frame CoM velocity (optional) */
} rg_mpc_out_ptrs;

typedef struct rg_mpc_handle rg_mpc_handle;

/* Allocates per-robot persistent state for `batch` robots on HIP device `device`.
 * Mirrors MPCController.__init__/_setup_controller (mpc_controller.py:18-66). */
int rg_mpc_create(const rg_mpc_config *cfg, int32_t batch, int32_t device, rg_mpc_handle **out);

/* LocomotionController.reset() (via MPCController.reset, mpc_controller.py:108-109) for the
 * robots idx[0..n) (HOST pointer; NULL = all).  t0 = clock value at reset
 * (core/simulation.py:141-142).  Swing start positions are latched from the foot positions
 * of the first rg_mpc_step after the reset.  With an index list the call waits for its own small
 * host->device copy (the staging buffer is reused); with idx_host == NULL it is fully asynchronous. */
int rg_mpc_reset(rg_mpc_handle *h, const int32_t *idx_host, int32_t n, double t0, void *stream);

/* Same, with one clock value per robot: t0_host[k] applies to idx_host[k] (idx NULL = robots
 * 0..n-1).  Lets a vectorised env reset its sub-envs at different times in one call. */
int rg_mpc_reset_at(rg_mpc_handle *h, const int32_t *idx_host, const double *t0_host, int32_t n, void *stream);

/* MPCController.update_controller_params (mpc_controller.py:83-100): cmd = [3][B] device
 * pointer, offsets already added.  Copied into the handle. */
int rg_mpc_set_command(rg_mpc_handle *h, const float *cmd, void *stream);

/* Per-robot gait timing: the arguments of OpenloopGaitGenerator that _setup_controller takes from the robot's constants
 * (mpc_controller.py:30-35: stance_duration, duty_factor, initial_leg_phase, initial_leg_state), one row per robot instead
 * of one per handle.  Device pointers, float64 / int32, [4][B] each (leg-major); copied into the handle on `stream`.
 * init_state may be NULL (config-wide initial states).  All four NULL returns to the config-wide gait.  Call it before
 * rg_mpc_reset so that a reset starts from the new initial states.  A robot whose row is out of range
 * (stance <= 0, duty outside (0, 1], non-finite phase, state not SWING/STANCE) is a counted failure each tick. */
int rg_mpc_set_gait(rg_mpc_handle *h, const double *stance_duration, const double *duty_factor, const double *init_phase,
                    const int32_t *init_state, void *stream);

/* MPCController.get_action (mpc_controller.py:102-106) for all B robots at clock value t. */
int rg_mpc_step(rg_mpc_handle *h, double t, const rg_mpc_state_ptrs *in, const rg_mpc_out_ptrs *out, void *stream);

/* The same tick for a caller whose robot state lives on the HOST -- the drop-in plugin at batch 1 (the reference's
 * MPCController.get_action gathers its state from PyBullet getters, mpc_controller.py:102-106) and a vectorised env:
 * copies slab_bytes from host_slab (pinned memory: the caller's staging area for every input array) to dev_slab, runs
 * rg_mpc_step on `in` / `out` (device pointers, normally into dev_slab), copies the [B][60] action slab to action_host
 * (pinned) and WAITS for it: four stream operations and the synchronisation in one call across the ABI.  action_host may be
 * NULL (no copy back, no wait). */
int rg_mpc_step_host(rg_mpc_handle *h, double t, const void *host_slab, void *dev_slab, int64_t slab_bytes,
                     const rg_mpc_state_ptrs *in, const rg_mpc_out_ptrs *out, float *action_host, void *stream);

/* RobotMotorModel.convert_to_torque, HYBRID branch (model/robots/simple_motor.py:128-140):
 * action [B][60], q/qd [12][B] -> tau [B][12].  Device pointers. */
int rg_mpc_hybrid_to_torque(rg_mpc_handle *h, const float *action, const float *q, const float *qd, float *tau, void *stream);

/* The same motor model over the action-repeat loop of one control tick: the reference applies one 60-float command
 * ACTION_REPEAT = 10 times (core/simulation.py:175-179, core/sim_constants.py:7), each time on the joint state of that
 * simulation sub-step (robot.py:276-307 -> simple_motor.py:128-140).  q / qd [S][12][B] -> tau [S][B][12], S = substeps. */
int rg_mpc_hybrid_to_torque_substeps(rg_mpc_handle *h, const float *action, const float *q, const float *qd, float *tau,
                                     int32_t substeps, void *stream);

/* Introspection for benches/profilers: number of robots per stance-leg count in the last
 * step (HOST out[5]); synchronises the stream. */
int rg_mpc_last_bin_counts(rg_mpc_handle *h, int32_t *out5, void *stream);

/* Per-robot view of the last step (HOST arrays of length batch, either may be NULL; synchronises the stream):
 * solver iterations and the number of stance legs the robot's QP was solved for. */
int rg_mpc_last_iterations(rg_mpc_handle *h, int32_t *iters_B, int32_t *stance_legs_B, void *stream);

/* Solver statistics of the last step (synchronises the stream): sum and max of solver iterations
 * (ADMM iterations / active-set constraint additions) over the robots that had a QP, the number of
 * such robots, how many were handed to the exact solver (RG_SOLVER_AUTO: by ADMM at the iteration cap, or directly by the front
 * kernel, see rg_mpc_last_direct_count) and how many solves failed
 * (robots with a non-finite input or an out-of-range gait row -- counted once, given an all-zero command row and
 * left out of the QP --, active-set breakdowns (iteration cap, a step that cannot be taken; the working set itself always has room),
 * plus, under RG_SOLVER_ADMM only, robots ADMM left unconverged). */
int rg_mpc_last_solver_stats(rg_mpc_handle *h, int64_t *iters_sum, int32_t *iters_max, int32_t *qp_robots,
                             int32_t *retried, int32_t *failures, void *stream);

/* Direct routing (horizon 10, constant contacts): a robot whose QP the exact re-solve had to take over -- ADMM at the iteration
 * cap, an exact body whose working set outgrew its room -- is, while its contact set stays the same, sent straight to the exact
 * solver with room for it by the following ticks (every 16th tick it tries its first body again).  Under RG_SOLVER_HYBRID /
 * RG_SOLVER_ACTIVE_SET the first workgroups of the QP launch solve those robots; under RG_SOLVER_AUTO they get a launch of
 * their own next to the ADMM launch when a recent tick had exact solves.  direct_robots: robots routed that way in the last
 * step; concurrent_launches: ticks so far that used the separate concurrent launch (0 under the hybrid plan).
 * Synchronises the stream. */
int rg_mpc_last_direct_count(rg_mpc_handle *h, int32_t *direct_robots, int64_t *concurrent_launches, void *stream);

/* Audit lane statistics, cumulative since create (or since the last call with reset != 0).  Waits for the audit work
 * in flight (library side stream) and for `stream`.  audited: robots re-solved exactly next to their converged ADMM
 * solution; over_tol: of those, robots whose torque error exceeded audit_tol; max_rel: largest per-robot error
 * max_j |dtau_j| / max(max_j |tau_j|, 1 N m); max_rel_elem: largest per-joint error |dtau_j| / max(|tau_j|, 1 N m);
 * exact_failures: audit re-solves that broke down (their robots are not counted in `audited`); dropped: picks that
 * found the tick's RG_MPC_AUDIT_SLOTS slots full.  Any out pointer may be NULL.  The tick that reuses a slot ring entry
 * (RG_MPC_AUDIT_RING x RG_MPC_AUDIT_PERIOD ticks later) waits on its stream for the entry's re-solves. */
#define RG_MPC_AUDIT_RING 4
#define RG_MPC_AUDIT_SLOTS 256
#define RG_MPC_AUDIT_PERIOD 8
int rg_mpc_audit_stats(rg_mpc_handle *h, int64_t *audited, int64_t *over_tol, double *max_rel, double *max_rel_elem,
                       int64_t *exact_failures, int64_t *dropped, int32_t reset, void *stream);

/* Per-kernel timing with hipEvents recorded on the step's own stream, between the launches of
 * rg_mpc_step.  begin(max_steps) arms it; every following rg_mpc_step records one event after
 * each launch; end() synchronises the stream, returns the number of recorded steps and fills
 * avg_ms[6] = average duration of {front, qp nc=1, qp nc=2, qp nc=3, qp nc=4, whole step}
 * and robots[5] = robots per stance-leg bin in the last recorded step. */
int rg_mpc_profile_begin(rg_mpc_handle *h, int32_t max_steps);
/* Record events only on every stride-th step after profile_begin (default 1).  An event record costs
 * ~4-5 us of stream time on MI355X, so timing every step of a 0.3 ms tick inflates it by > 10 %. */
int rg_mpc_profile_stride(rg_mpc_handle *h, int32_t stride);
int rg_mpc_profile_end(rg_mpc_handle *h, float *avg_ms6, int32_t *robots5, void *stream);

/* Test hook: overwrite the LDS of every CU with NaN bit patterns before the next rg_mpc_step, so reads of
 * never-written LDS show up deterministically in the parity tests. */
int rg_mpc_debug_poison_lds(rg_mpc_handle *h, void *stream);

/* Comma-separated labels of the six avg_ms windows for this handle's launch plan (horizon 10 with the
 * ADMM solver uses one fused QP launch: {front, fused QP, exact re-solves, -, -, whole step}). */
const char *rg_mpc_profile_window_names(const rg_mpc_handle *h);

/* What rg_mpc_create chose for this handle, as one line of space-separated key=value pairs (keys: solver, horizon, batch,
 * lanes -- lanes per robot in the QP launch: 64 or 256, see lane_grid --, exact12 -- one / two stance legs on the exact body --,
 * mu -- "uniform" or "per_leg" kernel instantiations --, schedule, audit, direct).  For logs and tests; valid until destroy. */
const char *rg_mpc_plan_description(const rg_mpc_handle *h);

/* Names of the kernels launched by rg_mpc_step, for matching rocprof rows. */
const char *rg_mpc_kernel_names(void);

void rg_mpc_destroy(rg_mpc_handle *h);
const char *rg_mpc_last_error(const rg_mpc_handle *h); /* h may be NULL: error of the last failed create */
int rg_mpc_abi_version(void);
int rg_mpc_config_size(void); /* sizeof(rg_mpc_config), for binding self-checks */

#ifdef __cplusplus
}
#endif
#endif
