/*
 * mpc_oracle.h -- CPU restatement (float64, plain C) of the robot-gym convex-MPC gait
 * controller tick.  TEST INFRASTRUCTURE ONLY: nothing under robot_gym_amd/ may link, load
 * or call this.  Allowed callers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline.
 *
 * PARITY STATUS: "parity unpinned" for the MPC arithmetic.  The reference tree
 * (/root/reference) holds only the 148-line adapter
 * robot_gym/controllers/mpc/mpc_controller.py; every numeric step lives in the un-vendored
 * dependency motion_imitation==0.0.5 (requirements.txt:8; modules mpc_controller, mpc_osqp)
 * which is absent from the reference tree, not installed, and cannot be fetched.  The
 * functions below restate that package's published algorithm (Di Carlo et al., IROS 2018 as
 * implemented by google-research/motion_imitation) from memory; each such function is tagged
 * [UPSTREAM-RECALL].  Functions tagged [REF file:line] follow code that IS in the reference
 * tree and are pinned by golden vectors generated from it (tests/golden/).
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NLEG 4
#define ORC_NMOTOR 12
#define ORC_NX 13
#define ORC_MAXH 20
#define ORC_WINDOW_MAX 64

/* gait_generator.LegState [UPSTREAM-RECALL]; used by REF model/robots/ghost/ctrl_constants.py:32-37 */
enum { ORC_SWING = 0, ORC_STANCE = 1, ORC_EARLY_CONTACT = 2, ORC_LOSE_CONTACT = 3 };

typedef struct {
  /* --- stance MPC (torque_stance_leg_controller / mpc_osqp defaults) --- */
  int horizon;              /* 10 */
  double dt_plan;           /* 0.025 */
  double mass;              /* REF ghost/ctrl_constants.py:8  190/9.8 */
  double inertia[9];        /* REF ghost/ctrl_constants.py:9  row-major body inertia */
  double body_height;       /* REF ghost/ctrl_constants.py:10 */
  double weights[ORC_NX];   /* (rpy, xyz, omega, v, g) */
  double alpha;             /* 1e-5; P = 2 B'WB + alpha I */
  double mu[4];             /* 0.45 x4 */
  double fz_max_scale;      /* 10  -> fz_max = scale*m*g */
  double fz_min_scale;      /* 0.1 -> fz_min = scale*m*g */
  double gravity;           /* 9.8 */
  /* --- open-loop gait (REF mpc_controller.py:30-35 kwargs; ctrl_constants.py:13,28-37) --- */
  double stance_duration[4];
  double duty_factor[4];
  double init_phase[4];
  int init_state[4];
  double contact_phase_thresh; /* 0.1 */
  /* --- velocity estimator (REF mpc_controller.py:36 window_size=20) --- */
  int window;
  /* --- Raibert swing (REF mpc_controller.py:44-45) --- */
  double foot_clearance;    /* 0.01 */
  double swing_kp[3];       /* 0.03 each */
  double max_clearance;     /* 0.1 */
  double hip[4][3];         /* REF ghost/constants.py:31-36 */
  /* --- motors (REF ghost/motor_constants.py:5-19) --- */
  double motor_kp[12], motor_kd[12], motor_dir[12], motor_off[12];
  /* --- leg chain model (URDF-derived; REF util/pybullet_data/robots/ghost.urdf) --- */
  double jxyz[4][3][3];     /* joint origin translation */
  double jrpy[4][3][3];     /* joint origin rpy */
  double jaxis[4][3][3];    /* joint axis (in joint frame) */
  double toe_xyz[4][3];     /* fixed toe joint origin in lower-leg frame */
  double toe_com[4][3];     /* toe link COM offset (pybullet getLinkState()[0] is the COM) */
  double base_com[3];       /* base link COM offset (pybullet base frame = base COM) */
  int ik_iters;             /* fixed DLS-Newton iteration count */
  double ik_damping;        /* lambda^2 */
  double ik_max_step;       /* per-iteration joint step clamp [rad] */
  /* --- input mode --- */
  int kin_mode;             /* 0: foot_pos + jac supplied; 1: computed from q by chain model */
  int contact_lookahead;    /* EXTENSION (not upstream): contact flags per horizon step from the open-loop gait at t + k*dt_plan */
  /* --- recall-sensitive conventions [UPSTREAM-RECALL]: 0 = the reading restated here by default, 1 = the other plausible
   * one (same switches as rg_mpc_config.conv_*; tests/golden/make_upstream_golden.py + tests/test_upstream_golden.py say
   * which reading the upstream package has, on a machine where it can be imported) --- */
  int conv_alpha_doubled;   /* 1: P = 2 (B'WB + alpha I) instead of 2 B'WB + alpha I */
  int conv_feet_rotation;   /* 1: lever arms rotated with Ry(pitch) Rx(roll) (the inertia's order) instead of Rx(roll) Ry(pitch) */
  int conv_com_height;      /* 1: mean |z| of the contact feet instead of |mean z| */
  int conv_first_latch;     /* 1: the first update after a reset latches swing feet on a STANCE->SWING edge too */
  int conv_window_divide;   /* 1: the velocity window divides by the samples held while it fills, instead of by its size */
  int conv_friction_rows;   /* 1: mu[t] is the coefficient of cone row t (-fx, +fx, -fy, +fy) of every block instead of leg t's (exact QP only) */
} orc_config;

typedef struct {
  double reset_time;
  int need_latch;           /* latch foot positions at first step after reset */
  int first_update;         /* swing-controller list aliasing rule (see .c) */
  int last_desired[4];
  int desired[4];
  int leg_state[4];
  double phase[4];
  double ring[3][ORC_WINDOW_MAX];
  int ring_len, ring_head;
  double fsum[3], fcorr[3];
  double v_body[3];
  double latched[4][3];
  double swing_q[12];
  int swing_valid[12];
} orc_state;

typedef struct {
  double rpy[3];
  double rpy_rate[3];       /* body-frame angular velocity, REF robot.py:205-213 */
  double v_world[3];        /* REF robot.py:172-178 */
  double quat[4];           /* x,y,z,w  REF robot.py:180-183 */
  double q[12];             /* motor angles, REF robot.py:231-236 */
  double foot_pos[4][3];    /* base frame, REF robot.py:389-397 (kin_mode 0) */
  double jac[4][3][3];      /* d foot / d leg joints, base frame (kin_mode 0) */
  int contact[4];           /* REF robot.py:215-229 */
  double cmd[3];            /* vx, vy, wz AFTER offsets, REF mpc_controller.py:90-95 */
  /* EXTENSION (contact_lookahead only): caller-supplied contact schedule.  sched_valid != 0: bit k (1 <= k < H) of
   * sched[leg] = leg in contact at horizon step k; bit 0 is ignored (step 0 = the controller's own contact decision). */
  int sched_valid;
  int sched[4];
} orc_input;

typedef struct {
  float action[60];         /* hybrid (q*,kp,qd*,kd,tau)x12, REF simple_motor.py:15-22 */
  double grf[12];           /* first-step contact forces (already negated) */
  double tau[12];           /* stance feed-forward torques, all 12 joints */
  int desired[4], leg_state[4];
  double phase[4];
  double v_body[3];
  double foot_target[4][3]; /* swing trajectory point this tick (0 where not computed) */
  int qp_iters;             /* active-set iterations */
  double kkt[3];            /* stationarity, primal infeasibility, complementarity */
} orc_output;

void orc_default_config(orc_config *c);      /* ghost defaults, kin_mode 0 */
void orc_reset(const orc_config *c, orc_state *s, double t_now, const double *foot_pos /*[12] or NULL*/);
int orc_step(const orc_config *c, orc_state *s, double t_now, const orc_input *in, orc_output *out);

/* building blocks (exposed for tests) */
void orc_gait(const orc_config *c, double t, const int contact[4], int desired[4], int leg_state[4], double phase[4]);
double orc_filter_push(orc_state *s, int axis, int window, double v);
void orc_swing_trajectory(double phase, const double start[3], const double end[3], double max_clearance, double out[3]);
void orc_leg_fk(const orc_config *c, int leg, const double q3[3], double p[3], double J[9]);
int orc_leg_ik(const orc_config *c, int leg, const double target[3], const double q_init[3], double q_out[3]);
void orc_hybrid_to_torque(const float action[60], const double q[12], const double qd[12], double tau[12]);
void orc_force_to_torque(const orc_config *c, int leg, const double f[3], const double J[9], double tau3[3]);

/* MPC QP assembly for the contact-leg-reduced problem.
 * n = 3*nc*H variables ordered (step k, contact leg j, xyz).
 * P (n*n, row-major), qv (n), and legs[] = indices of contact legs.  Returns nc. */
int orc_mpc_build(const orc_config *c, const double rpy[3], const double omega[3], const double v_body[3],
                  const double foot_pos[12], const int contact[4], const double cmd[3],
                  double *P, double *qv, int legs[4], double *Ad /*169 or NULL*/, double *Bd /*13*12 or NULL*/);
/* Same with a per-step contact schedule sched[k*4+leg] (k = 0..H-1); variables ordered (step, leg in
 * contact at that step, xyz).  contact[] (step 0) is still what the CoM-height estimate uses.
 * Returns n; var_step/var_leg (size >= 4H) describe each 3-block. */
int orc_mpc_build_sched(const orc_config *c, const double rpy[3], const double omega[3], const double v_body[3],
                        const double foot_pos[12], const int contact[4], const int *sched, const double cmd[3],
                        double *P, double *qv, int *var_step, int *var_leg);
void orc_gait_desired(const orc_config *c, double t, int desired[4]);
/* exact dual active-set solve of  min 1/2 u'Pu + q'u  s.t. friction pyramid + fz box per 3-block.
 * Returns iterations (<0 on failure). */
/* the same with one friction coefficient per cone ROW (-fx, +fx, -fy, +fy) of every block: rg_mpc_config.conv_friction_rows */
int orc_qp_solve_rows(int n, const double *P, const double *qv, const double *mu_rows, double fz_lo, double fz_hi, double *x, double kkt[3]);
int orc_qp_solve(int n, const double *P, const double *qv, const double *mu_blk /* n/3 */, double fz_min, double fz_max,
                 double *u, double kkt[3]);

/* bench.py's cpu_baseline variant B1 only: mode 1 = the same over-relaxed ADMM as the GPU kernels with a FIXED iteration
 * count and dense linear algebra instead of the exact solver (process-wide; mode 0 = exact, the default and the parity oracle) */
void orc_set_qp_mode(int mode, int admm_iters, double rho, double relax);

/* batch helpers (OpenMP over robots) */
int orc_step_batch(const orc_config *c, orc_state *s, int B, double t_now, const orc_input *in, orc_output *out, int nthreads);
/* same with one config per robot (per-robot gait timing: BASELINE config 5 draws a random duty factor per robot) */
int orc_step_batch_cfgs(const orc_config *cfgs, orc_state *s, int B, double t_now, const orc_input *in, orc_output *out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
