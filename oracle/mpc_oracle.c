/*
 * mpc_oracle.c -- float64 CPU restatement of one robot-gym MPC controller tick.
 * TEST INFRASTRUCTURE ONLY (see mpc_oracle.h).  PARITY UNPINNED for [UPSTREAM-RECALL] parts.
 *
 * One tick == everything MPCController.get_action() does
 * (REF robot_gym/controllers/mpc/mpc_controller.py:102-106):
 *   LocomotionController.update()  -> gait, velocity estimator, swing latch     [UPSTREAM-RECALL]
 *   LocomotionController.get_action() -> swing 5-tuples, stance QP, merge to 60 [UPSTREAM-RECALL]
 * The algorithm is written the way the upstream library does it (dense 25x25 matrix
 * exponential, A_d powers, dense condensed B_qp, dense P) -- deliberately NOT the closed
 * forms the HIP kernels use, so that the two are independent derivations.
 */
#include "mpc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* per-thread workspace                                                                  */
/* ------------------------------------------------------------------------------------ */
/* Every dense temporary of a tick (the 13H x 12H condensed matrices, the Hessian, the active-set solver's factors: ~20
 * buffers, up to ~5 MB at horizon 20) comes from a per-thread bump arena instead of malloc / calloc / free: with one
 * heap call per buffer inside the OpenMP loop over robots the port stopped scaling at ~32 threads (the allocator's
 * locks and page faults, not the arithmetic: round-3 review).  The public entry points take a mark on entry and release
 * back to it on exit (stack discipline, so nested entry points work); the arena itself is allocated once per thread and
 * kept (untouched pages of it are never committed). */
#define WS_BYTES ((size_t)64 << 20)
static __thread unsigned char *ws_base;
static __thread size_t ws_top;
static void *ws_alloc(size_t bytes) {
  if (!ws_base) { ws_base = malloc(WS_BYTES); if (!ws_base) { fprintf(stderr, "mpc_oracle: workspace allocation failed\n"); abort(); } }
  bytes = (bytes + 63) & ~(size_t)63;
  if (ws_top + bytes > WS_BYTES) { fprintf(stderr, "mpc_oracle: per-thread workspace exhausted\n"); abort(); }
  void *p = ws_base + ws_top;
  ws_top += bytes;
  return p;
}
static void *ws_calloc(size_t n, size_t size) { void *p = ws_alloc(n * size); memset(p, 0, n * size); return p; }
static size_t ws_mark(void) { return ws_top; }
static void ws_release(size_t mark) { ws_top = mark; }

/* ------------------------------------------------------------------------------------ */
/* defaults: ghost                                                                       */
/* ------------------------------------------------------------------------------------ */
void orc_default_config(orc_config *c) {
  memset(c, 0, sizeof(*c));
  c->horizon = 10;                      /* [UPSTREAM-RECALL] _PLANNING_HORIZON_STEPS */
  c->dt_plan = 0.025;                   /* [UPSTREAM-RECALL] _PLANNING_TIMESTEP */
  c->mass = 190.0 / 9.8;                /* REF ghost/ctrl_constants.py:8 */
  c->inertia[0] = 0.07335; c->inertia[4] = 0.25068; c->inertia[8] = 0.25447; /* REF :9 */
  c->body_height = 0.42;                /* REF :10 */
  /* [UPSTREAM-RECALL] _MPC_WEIGHTS as listed in SURVEY.md 8a-18 */
  const double w[13] = {5, 5, 0.2, 0, 0, 10, 0.5, 0.5, 0.2, 0.2, 0.2, 0.1, 0};
  memcpy(c->weights, w, sizeof(w));
  c->alpha = 1e-5;
  for (int i = 0; i < 4; i++) c->mu[i] = 0.45;
  c->fz_max_scale = 10.0;
  c->fz_min_scale = 0.1;
  c->gravity = 9.8;
  for (int i = 0; i < 4; i++) { c->stance_duration[i] = 0.3; c->duty_factor[i] = 0.6; } /* REF :13,28 */
  c->init_phase[0] = 0.9; c->init_phase[1] = 0; c->init_phase[2] = 0; c->init_phase[3] = 0.9; /* REF :29 */
  c->init_state[0] = ORC_SWING; c->init_state[1] = ORC_STANCE;
  c->init_state[2] = ORC_STANCE; c->init_state[3] = ORC_SWING;                          /* REF :32-37 */
  c->contact_phase_thresh = 0.1;
  c->window = 20;                       /* REF mpc_controller.py:36 */
  c->foot_clearance = 0.01;             /* REF mpc_controller.py:45 */
  for (int i = 0; i < 3; i++) c->swing_kp[i] = 0.03;
  c->max_clearance = 0.1;
  const double hip[4][3] = {{0.22, -0.1, 0}, {0.22, 0.1, 0}, {-0.22, -0.1, 0}, {-0.22, 0.1, 0}}; /* REF ghost/constants.py:31-36 */
  memcpy(c->hip, hip, sizeof(hip));
  for (int i = 0; i < 12; i++) {
    c->motor_kp[i] = 220.0;                              /* REF ghost/motor_constants.py:13 */
    c->motor_kd[i] = (i % 3 == 0) ? 1.0 : 2.0;           /* REF :15 */
    c->motor_dir[i] = 1.0; c->motor_off[i] = 0.0;        /* REF :9,11 */
  }
  c->ik_iters = 8; c->ik_damping = 1e-10; c->ik_max_step = 0.5;
  c->kin_mode = 0;
}

/* ------------------------------------------------------------------------------------ */
/* small linear algebra                                                                  */
/* ------------------------------------------------------------------------------------ */
static void mat3_mul(const double *a, const double *b, double *c) {
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double s = 0; for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j]; t[3 * i + j] = s; }
  memcpy(c, t, sizeof(t));
}
static void mat3_vec(const double *a, const double *v, double *o) {
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void rot_x(double a, double *r) { double c = cos(a), s = sin(a); double m[9] = {1, 0, 0, 0, c, -s, 0, s, c}; memcpy(r, m, sizeof(m)); }
static void rot_y(double a, double *r) { double c = cos(a), s = sin(a); double m[9] = {c, 0, s, 0, 1, 0, -s, 0, c}; memcpy(r, m, sizeof(m)); }
static void rot_z(double a, double *r) { double c = cos(a), s = sin(a); double m[9] = {c, -s, 0, s, c, 0, 0, 0, 1}; memcpy(r, m, sizeof(m)); }
/* URDF fixed-axis rpy == Rz(y) Ry(p) Rx(r); also Eigen yaw*pitch*roll (ConvertRpyToRot) */
static void rot_rpy_zyx(const double rpy[3], double *r) {
  double rx[9], ry[9], rz[9], t[9];
  rot_x(rpy[0], rx); rot_y(rpy[1], ry); rot_z(rpy[2], rz);
  mat3_mul(rz, ry, t); mat3_mul(t, rx, r);
}
static void rot_axis_angle(const double ax[3], double ang, double *r) {
  double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  double x = ax[0] / n, y = ax[1] / n, z = ax[2] / n, c = cos(ang), s = sin(ang), C = 1 - c;
  double m[9] = {c + x * x * C, x * y * C - z * s, x * z * C + y * s,
                 y * x * C + z * s, c + y * y * C, y * z * C - x * s,
                 z * x * C - y * s, z * y * C + x * s, c + z * z * C};
  memcpy(r, m, sizeof(m));
}
static void cross3(const double *a, const double *b, double *c) {
  double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
  c[0] = t0; c[1] = t1; c[2] = t2;
}

/* ------------------------------------------------------------------------------------ */
/* gait  [UPSTREAM-RECALL openloop_gait_generator.OpenloopGaitGenerator.update]          */
/* kwargs pinned by REF mpc_controller.py:30-35                                          */
/* ------------------------------------------------------------------------------------ */
void orc_gait(const orc_config *c, double t, const int contact[4], int desired[4], int leg_state[4], double phase[4]) {
  for (int leg = 0; leg < 4; leg++) {
    int init = c->init_state[leg];
    int next = (init == ORC_SWING) ? ORC_STANCE : ORC_SWING;
    double ratio = (init == ORC_SWING) ? 1.0 - c->duty_factor[leg] : c->duty_factor[leg];
    double full = c->stance_duration[leg] / c->duty_factor[leg];
    double aug = t + c->init_phase[leg] * full;
    double ph = fmod(aug, full) / full;
    if (ph < ratio) { desired[leg] = init; phase[leg] = ph / ratio; }
    else { desired[leg] = next; phase[leg] = (ph - ratio) / (1.0 - ratio); }
    leg_state[leg] = desired[leg];
    if (phase[leg] < c->contact_phase_thresh) continue;
    if (leg_state[leg] == ORC_SWING && contact[leg]) leg_state[leg] = ORC_EARLY_CONTACT;
    if (leg_state[leg] == ORC_STANCE && !contact[leg]) leg_state[leg] = ORC_LOSE_CONTACT;
  }
}

/* desired (open-loop) leg states only, for the contact look-ahead extension */
void orc_gait_desired(const orc_config *c, double t, int desired[4]) {
  int contact[4] = {1, 1, 1, 1}, ls[4]; double ph[4];
  orc_gait(c, t, contact, desired, ls, ph);
}

/* ------------------------------------------------------------------------------------ */
/* moving-window filter with Neumaier compensation                                       */
/* [UPSTREAM-RECALL com_velocity_estimator.MovingWindowFilter]: divides by the window    */
/* size even before the window is full.                                                  */
/* ------------------------------------------------------------------------------------ */
static void neumaier(double *sum, double *corr, double v) {
  double ns = *sum + v;
  if (fabs(*sum) >= fabs(v)) *corr += (*sum - ns) + v; else *corr += (v - ns) + *sum;
  *sum = ns;
}
static double filter_push_div(orc_state *s, int axis, int window, double v, int by_samples) {
  /* ring_len/ring_head are advanced by the caller after all three axes were pushed */
  if (s->ring_len >= window) neumaier(&s->fsum[axis], &s->fcorr[axis], -s->ring[axis][s->ring_head]);
  s->ring[axis][s->ring_head] = v;
  neumaier(&s->fsum[axis], &s->fcorr[axis], v);
  const int held = s->ring_len >= window ? window : s->ring_len + 1;   /* samples in the window after this push */
  return (s->fsum[axis] + s->fcorr[axis]) / (double)(by_samples ? held : window);
}
double orc_filter_push(orc_state *s, int axis, int window, double v) { return filter_push_div(s, axis, window, v, 0); }

/* ------------------------------------------------------------------------------------ */
/* swing trajectory [UPSTREAM-RECALL raibert_swing_leg_controller._gen_swing_foot_trajectory] */
/* ------------------------------------------------------------------------------------ */
static double gen_parabola(double phase, double start, double mid, double end) {
  double mid_phase = 0.5;
  double d1 = mid - start, d2 = end - start, d3 = mid_phase * mid_phase - mid_phase;
  double a = (d1 - d2 * mid_phase) / d3;
  double b = (d2 * mid_phase * mid_phase - d1) / d3;
  return a * phase * phase + b * phase + start;
}
void orc_swing_trajectory(double input_phase, const double start[3], const double end[3], double max_clearance, double out[3]) {
  double phase;
  if (input_phase <= 0.5) phase = 0.8 * sin(input_phase * M_PI);
  else phase = 0.8 + (input_phase - 0.5) * 0.4;
  out[0] = (1 - phase) * start[0] + phase * end[0];
  out[1] = (1 - phase) * start[1] + phase * end[1];
  double mid = fmax(end[2], start[2]) + max_clearance;
  out[2] = gen_parabola(phase, start[2], mid, end[2]);
}

/* ------------------------------------------------------------------------------------ */
/* leg kinematics: generic 3-revolute chain from the URDF.                               */
/* Replaces REF controllers/mpc/kinematics.py:13-30 (pybullet calculateJacobian at the   */
/* toe link COM, local point (0,0,0)) and REF robot.py:367-397 (toe COM in base frame).  */
/* ------------------------------------------------------------------------------------ */
void orc_leg_fk(const orc_config *c, int leg, const double qm[3], double p[3], double J[9]) {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
  double ax_w[3][3], org[3][3];
  for (int j = 0; j < 3; j++) {
    double t[3], Rf[9], Rq[9];
    mat3_vec(R, c->jxyz[leg][j], t);
    o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
    rot_rpy_zyx(c->jrpy[leg][j], Rf);
    mat3_mul(R, Rf, R);
    mat3_vec(R, c->jaxis[leg][j], ax_w[j]);
    double n = sqrt(ax_w[j][0] * ax_w[j][0] + ax_w[j][1] * ax_w[j][1] + ax_w[j][2] * ax_w[j][2]);
    ax_w[j][0] /= n; ax_w[j][1] /= n; ax_w[j][2] /= n;
    org[j][0] = o[0]; org[j][1] = o[1]; org[j][2] = o[2];
    int m = 3 * leg + j;
    double joint = qm[j] * c->motor_dir[m] + c->motor_off[m]; /* inverse of REF robot.py:231-236 (dir = +-1) */
    rot_axis_angle(c->jaxis[leg][j], joint, Rq);
    mat3_mul(R, Rq, R);
  }
  double tip[3] = {c->toe_xyz[leg][0] + c->toe_com[leg][0], c->toe_xyz[leg][1] + c->toe_com[leg][1], c->toe_xyz[leg][2] + c->toe_com[leg][2]};
  double t[3];
  mat3_vec(R, tip, t);
  double pf[3] = {o[0] + t[0], o[1] + t[1], o[2] + t[2]};
  if (J) {
    for (int j = 0; j < 3; j++) {
      double d[3] = {pf[0] - org[j][0], pf[1] - org[j][1], pf[2] - org[j][2]}, col[3];
      cross3(ax_w[j], d, col);
      J[0 * 3 + j] = col[0]; J[1 * 3 + j] = col[1]; J[2 * 3 + j] = col[2]; /* JOINT-space column, like pybullet jv */
    }
  }
  p[0] = pf[0] - c->base_com[0]; p[1] = pf[1] - c->base_com[1]; p[2] = pf[2] - c->base_com[2];
}

static int solve3(const double *A, const double *b, double *x) {
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  if (det == 0.0) return -1;
  double inv = 1.0 / det;
  double i00 = c00 * inv, i01 = (A[2] * A[7] - A[1] * A[8]) * inv, i02 = (A[1] * A[5] - A[2] * A[4]) * inv;
  double i10 = c01 * inv, i11 = (A[0] * A[8] - A[2] * A[6]) * inv, i12 = (A[2] * A[3] - A[0] * A[5]) * inv;
  double i20 = c02 * inv, i21 = (A[1] * A[6] - A[0] * A[7]) * inv, i22 = (A[0] * A[4] - A[1] * A[3]) * inv;
  x[0] = i00 * b[0] + i01 * b[1] + i02 * b[2];
  x[1] = i10 * b[0] + i11 * b[1] + i12 * b[2];
  x[2] = i20 * b[0] + i21 * b[1] + i22 * b[2];
  return 0;
}

/* Replaces REF controllers/mpc/kinematics.py:98-133 (pybullet calculateInverseKinematics,
 * DLS solver, started from the current joint state).  Fixed-count damped Newton so the
 * result is a deterministic function of (target, q_init): dq = J'(JJ' + lambda^2 I)^-1 e. */
int orc_leg_ik(const orc_config *c, int leg, const double target[3], const double q_init[3], double q_out[3]) {
  double q[3] = {q_init[0], q_init[1], q_init[2]};
  for (int it = 0; it < c->ik_iters; it++) {
    double p[3], J[9], e[3], A[9], y[3];
    orc_leg_fk(c, leg, q, p, J);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[3 * i + j] *= c->motor_dir[3 * leg + j]; /* d p / d motor angle */
    for (int i = 0; i < 3; i++) e[i] = target[i] - p[i];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      double s = 0; for (int k = 0; k < 3; k++) s += J[3 * i + k] * J[3 * j + k];
      A[3 * i + j] = s + (i == j ? c->ik_damping : 0.0);
    }
    if (solve3(A, e, y)) break;
    for (int j = 0; j < 3; j++) {
      double dq = J[0 * 3 + j] * y[0] + J[1 * 3 + j] * y[1] + J[2 * 3 + j] * y[2];
      if (dq > c->ik_max_step) dq = c->ik_max_step;
      if (dq < -c->ik_max_step) dq = -c->ik_max_step;
      q[j] += dq;
    }
  }
  q_out[0] = q[0]; q_out[1] = q[1]; q_out[2] = q[2];
  return 0;
}

/* REF controllers/mpc/kinematics.py:40-53: all = f(1x3) . jv ; tau_j = all[6+joint]*MOTOR_DIRECTION.
 * J is the 3x3 block of jv (joint space) for this leg's joints: J[i][j] = d foot_i / d joint_j. */
void orc_force_to_torque(const orc_config *c, int leg, const double f[3], const double J[9], double tau3[3]) {
  for (int j = 0; j < 3; j++) tau3[j] = (f[0] * J[0 * 3 + j] + f[1] * J[1 * 3 + j] + f[2] * J[2 * 3 + j]) * c->motor_dir[3 * leg + j];
}

/* REF model/robots/simple_motor.py:128-140 (HYBRID branch; strength ratio 1, no limits) */
void orc_hybrid_to_torque(const float action[60], const double q[12], const double qd[12], double tau[12]) {
  for (int j = 0; j < 12; j++) {
    double qs = action[5 * j + 0], kp = action[5 * j + 1], qds = action[5 * j + 2], kd = action[5 * j + 3], ff = action[5 * j + 4];
    tau[j] = -1.0 * (kp * (q[j] - qs)) - kd * (qd[j] - qds) + ff;
  }
}

/* ------------------------------------------------------------------------------------ */
/* MPC QP assembly [UPSTREAM-RECALL mpc_osqp.cc ConvexMpc::ComputeContactForces]          */
/* ------------------------------------------------------------------------------------ */
static void matmul(int m, int k, int n, const double *A, const double *B, double *C) {
  for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) {
    double s = 0; for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j]; C[i * n + j] = s; }
}
/* dense matrix exponential by scaling-and-squaring Taylor (stand-in for Eigen's .exp()) */
static void expm_dense(int n, const double *M, double *E) {
  double nrm = 0;
  for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < n; j++) s += fabs(M[i * n + j]); if (s > nrm) nrm = s; }
  int sq = 0; double sc = 1.0;
  while (nrm * sc > 0.25) { sc *= 0.5; sq++; }
  double *A = ws_alloc(sizeof(double) * n * n), *T = ws_alloc(sizeof(double) * n * n), *T2 = ws_alloc(sizeof(double) * n * n);
  for (int i = 0; i < n * n; i++) A[i] = M[i] * sc;
  for (int i = 0; i < n * n; i++) { E[i] = 0; T[i] = 0; }
  for (int i = 0; i < n; i++) { E[i * n + i] = 1; T[i * n + i] = 1; }
  for (int k = 1; k <= 24; k++) {
    matmul(n, n, n, T, A, T2);
    for (int i = 0; i < n * n; i++) { T[i] = T2[i] / k; E[i] += T[i]; }
  }
  for (int s = 0; s < sq; s++) { matmul(n, n, n, E, E, T2); memcpy(E, T2, sizeof(double) * n * n); }
}

/* dense, all four legs: P (12H x 12H) and q (12H); caller frees */
static void mpc_build_dense(const orc_config *c, const double rpy_in[3], const double omega[3], const double v_body[3],
                            const double foot_pos[12], const int contact[4], const double cmd[3],
                            double **P_out, double **q_out, double *Ad_out, double *Bd_out) {
  const int H = c->horizon, NX = ORC_NX, NU = 12;
  /* yaw-aligned frame: yaw zeroed  [UPSTREAM-RECALL torque_stance_leg_controller.get_action] */
  double rpy[3] = {rpy_in[0], rpy_in[1], 0.0};
  /* feet to the world-aligned frame: AngleAxis(roll,X)*AngleAxis(pitch,Y)*AngleAxis(yaw,Z) */
  double rx[9], ry[9], rz[9], Rfeet[9], t9[9];
  rot_x(rpy[0], rx); rot_y(rpy[1], ry); rot_z(rpy[2], rz);
  if (c->conv_feet_rotation) { mat3_mul(ry, rx, t9); mat3_mul(rz, t9, Rfeet); }   /* the inertia's order: Rz Ry Rx */
  else { mat3_mul(rx, ry, t9); mat3_mul(t9, rz, Rfeet); }
  double fw[4][3];
  for (int i = 0; i < 4; i++) mat3_vec(Rfeet, &foot_pos[3 * i], fw[i]);
  /* EstimateCoMHeightSimple: |mean z of contact feet| */
  int ncontact = 0; double hz = 0;
  for (int i = 0; i < 4; i++) if (contact[i]) { hz += c->conv_com_height ? fabs(fw[i][2]) : fw[i][2]; ncontact++; }
  double com_z = ncontact > 0 ? fabs(hz / ncontact) : 0.0;
  double x0[13] = {rpy[0], rpy[1], rpy[2], 0, 0, com_z, omega[0], omega[1], omega[2], v_body[0], v_body[1], v_body[2], -c->gravity};
  double *xd = ws_calloc((size_t)NX * H, sizeof(double));
  for (int i = 0; i < H; i++) {
    double *d = xd + i * NX;
    d[0] = 0; d[1] = 0; d[2] = rpy[2] + c->dt_plan * (i + 1) * cmd[2];
    d[3] = c->dt_plan * (i + 1) * cmd[0]; d[4] = c->dt_plan * (i + 1) * cmd[1]; d[5] = c->body_height;
    d[6] = 0; d[7] = 0; d[8] = cmd[2];
    d[9] = cmd[0]; d[10] = cmd[1]; d[11] = 0;
    d[12] = -c->gravity;
  }
  /* CalculateAMat */
  double A[13 * 13]; memset(A, 0, sizeof(A));
  double cy = cos(rpy[2]), sy = sin(rpy[2]), cp = cos(rpy[1]), tp = tan(rpy[1]);
  A[0 * 13 + 6] = cy / cp; A[0 * 13 + 7] = sy / cp; A[0 * 13 + 8] = 0;
  A[1 * 13 + 6] = -sy;     A[1 * 13 + 7] = cy;      A[1 * 13 + 8] = 0;
  A[2 * 13 + 6] = cy * tp; A[2 * 13 + 7] = sy * tp; A[2 * 13 + 8] = 1;
  A[3 * 13 + 9] = 1; A[4 * 13 + 10] = 1; A[5 * 13 + 11] = 1; A[11 * 13 + 12] = 1;
  /* inverse inertia in world-aligned frame: R I^-1 R', R = ConvertRpyToRot = Rz Ry Rx */
  double Rb[9], Iinv[9], Rt[9], Iw[9];
  rot_rpy_zyx(rpy, Rb);
  {
    const double *I = c->inertia;
    double c00 = I[4] * I[8] - I[5] * I[7], c01 = I[5] * I[6] - I[3] * I[8], c02 = I[3] * I[7] - I[4] * I[6];
    double det = I[0] * c00 + I[1] * c01 + I[2] * c02, inv = 1.0 / det;
    Iinv[0] = c00 * inv; Iinv[1] = (I[2] * I[7] - I[1] * I[8]) * inv; Iinv[2] = (I[1] * I[5] - I[2] * I[4]) * inv;
    Iinv[3] = c01 * inv; Iinv[4] = (I[0] * I[8] - I[2] * I[6]) * inv; Iinv[5] = (I[2] * I[3] - I[0] * I[5]) * inv;
    Iinv[6] = c02 * inv; Iinv[7] = (I[1] * I[6] - I[0] * I[7]) * inv; Iinv[8] = (I[0] * I[4] - I[1] * I[3]) * inv;
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = Rb[3 * j + i];
  mat3_mul(Rb, Iinv, t9); mat3_mul(t9, Rt, Iw);
  /* CalculateBMat (all four legs; non-contact columns are dropped below like the qpOASES path) */
  double B[13 * 12]; memset(B, 0, sizeof(B));
  for (int i = 0; i < 4; i++) {
    double sk[9] = {0, -fw[i][2], fw[i][1], fw[i][2], 0, -fw[i][0], -fw[i][1], fw[i][0], 0}, blk[9];
    mat3_mul(Iw, sk, blk);
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) B[(6 + r) * 12 + 3 * i + cc] = blk[3 * r + cc];
    B[9 * 12 + 3 * i] = 1.0 / c->mass; B[10 * 12 + 3 * i + 1] = 1.0 / c->mass; B[11 * 12 + 3 * i + 2] = 1.0 / c->mass;
  }
  /* CalculateExponentials: expm([[A,B],[0,0]] dt) */
  const int NE = NX + NU;
  double M[25 * 25], E[25 * 25]; memset(M, 0, sizeof(M));
  for (int i = 0; i < NX; i++) {
    for (int j = 0; j < NX; j++) M[i * NE + j] = A[i * 13 + j] * c->dt_plan;
    for (int j = 0; j < NU; j++) M[i * NE + NX + j] = B[i * 12 + j] * c->dt_plan;
  }
  expm_dense(NE, M, E);
  double Ad[13 * 13], Bd[13 * 12];
  for (int i = 0; i < NX; i++) {
    for (int j = 0; j < NX; j++) Ad[i * 13 + j] = E[i * NE + j];
    for (int j = 0; j < NU; j++) Bd[i * 12 + j] = E[i * NE + NX + j];
  }
  if (Ad_out) memcpy(Ad_out, Ad, sizeof(Ad));
  if (Bd_out) memcpy(Bd_out, Bd, sizeof(Bd));
  /* CalculateQpMats: a_qp blocks = Ad^(k+1); anb[i] = Ad^i Bd; dense b_qp */
  double *aqp = ws_calloc((size_t)NX * H * NX, sizeof(double));
  double *anb = ws_calloc((size_t)NX * H * NU, sizeof(double));
  memcpy(aqp, Ad, sizeof(Ad));
  memcpy(anb, Bd, sizeof(Bd));
  for (int i = 1; i < H; i++) {
    matmul(NX, NX, NX, Ad, aqp + (size_t)(i - 1) * NX * NX, aqp + (size_t)i * NX * NX);
    matmul(NX, NX, NU, Ad, anb + (size_t)(i - 1) * NX * NU, anb + (size_t)i * NX * NU);
  }
  const int NR = NX * H, NC = NU * H;
  double *bqp = ws_calloc((size_t)NR * NC, sizeof(double));
  for (int i = 0; i < H; i++) for (int j = 0; j <= i; j++) {
    const double *blk = anb + (size_t)(i - j) * NX * NU;
    for (int r = 0; r < NX; r++) for (int cc = 0; cc < NU; cc++) bqp[(size_t)(i * NX + r) * NC + j * NU + cc] = blk[r * NU + cc];
  }
  /* P = 2 B'WB + alpha I ; q = 2 B'W (A_qp x0 - X*) */
  double *WB = ws_alloc(sizeof(double) * NR * NC);
  for (int r = 0; r < NR; r++) { double w = c->weights[r % NX]; for (int cc = 0; cc < NC; cc++) WB[(size_t)r * NC + cc] = w * bqp[(size_t)r * NC + cc]; }
  double *P = ws_calloc((size_t)NC * NC, sizeof(double));
  for (int i = 0; i < NC; i++) for (int j = i; j < NC; j++) {
    double s = 0; for (int r = 0; r < NR; r++) s += bqp[(size_t)r * NC + i] * WB[(size_t)r * NC + j];
    P[(size_t)i * NC + j] = 2 * s; P[(size_t)j * NC + i] = 2 * s;
  }
  for (int i = 0; i < NC; i++) P[(size_t)i * NC + i] += (c->conv_alpha_doubled ? 2.0 : 1.0) * c->alpha;
  double *sd = ws_alloc(sizeof(double) * NR);
  for (int i = 0; i < H; i++) for (int r = 0; r < NX; r++) {
    double s = 0; for (int k = 0; k < NX; k++) s += aqp[(size_t)i * NX * NX + r * NX + k] * x0[k];
    sd[i * NX + r] = s - xd[i * NX + r];
  }
  double *qf = ws_alloc(sizeof(double) * NC);
  for (int i = 0; i < NC; i++) { double s = 0; for (int r = 0; r < NR; r++) s += WB[(size_t)r * NC + i] * sd[r]; qf[i] = 2 * s; }
  *P_out = P; *q_out = qf;
}

static int orc_mpc_build_impl(const orc_config *c, const double rpy_in[3], const double omega[3], const double v_body[3],
                  const double foot_pos[12], const int contact[4], const double cmd[3],
                  double *Pr, double *qr, int legs[4], double *Ad_out, double *Bd_out) {
  const int H = c->horizon, NU = 12, NC = NU * H;
  double *P, *qf;
  mpc_build_dense(c, rpy_in, omega, v_body, foot_pos, contact, cmd, &P, &qf, Ad_out, Bd_out);
  /* reduce to contact legs (qpOASES path CopyToMatrix/CopyToVec) */
  int nc = 0; for (int i = 0; i < 4; i++) if (contact[i]) legs[nc++] = i;
  const int n = 3 * nc * H;
  for (int ka = 0; ka < H; ka++) for (int la = 0; la < nc; la++) for (int da = 0; da < 3; da++) {
    int ir = (ka * nc + la) * 3 + da, i_f = ka * NU + legs[la] * 3 + da;
    qr[ir] = qf[i_f];
    for (int kb = 0; kb < H; kb++) for (int lb = 0; lb < nc; lb++) for (int db = 0; db < 3; db++) {
      int jr = (kb * nc + lb) * 3 + db, jf = kb * NU + legs[lb] * 3 + db;
      Pr[(size_t)ir * n + jr] = P[(size_t)i_f * NC + jf];
    }
  }
  return nc;
}
int orc_mpc_build(const orc_config *c, const double rpy_in[3], const double omega[3], const double v_body[3],
                  const double foot_pos[12], const int contact[4], const double cmd[3],
                  double *Pr, double *qr, int legs[4], double *Ad_out, double *Bd_out) {
  const size_t mark = ws_mark();   /* per-thread workspace: everything this call allocates is released on return */
  const int r = orc_mpc_build_impl(c, rpy_in, omega, v_body, foot_pos, contact, cmd, Pr, qr, legs, Ad_out, Bd_out);
  ws_release(mark);
  return r;
}

/* EXTENSION: per-step contact schedule.  Same dense P, q; a (step, leg) block
 * exists only where sched says the leg is in contact at that step. */
static int orc_mpc_build_sched_impl(const orc_config *c, const double rpy_in[3], const double omega[3], const double v_body[3],
                        const double foot_pos[12], const int contact[4], const int *sched, const double cmd[3],
                        double *Pr, double *qr, int *var_step, int *var_leg) {
  const int H = c->horizon, NU = 12, NC = NU * H;
  double *P, *qf;
  mpc_build_dense(c, rpy_in, omega, v_body, foot_pos, contact, cmd, &P, &qf, NULL, NULL);
  int nb = 0;
  for (int k = 0; k < H; k++) for (int l = 0; l < 4; l++) if (sched[k * 4 + l]) { var_step[nb] = k; var_leg[nb] = l; nb++; }
  const int n = 3 * nb;
  for (int ba = 0; ba < nb; ba++) for (int da = 0; da < 3; da++) {
    int ir = 3 * ba + da, i_f = var_step[ba] * NU + var_leg[ba] * 3 + da;
    qr[ir] = qf[i_f];
    for (int bb = 0; bb < nb; bb++) for (int db = 0; db < 3; db++) {
      int jr = 3 * bb + db, jf = var_step[bb] * NU + var_leg[bb] * 3 + db;
      Pr[(size_t)ir * n + jr] = P[(size_t)i_f * NC + jf];
    }
  }
  return n;
}
int orc_mpc_build_sched(const orc_config *c, const double rpy_in[3], const double omega[3], const double v_body[3],
                        const double foot_pos[12], const int contact[4], const int *sched, const double cmd[3],
                        double *Pr, double *qr, int *var_step, int *var_leg) {
  const size_t mark = ws_mark();   /* per-thread workspace: everything this call allocates is released on return */
  const int r = orc_mpc_build_sched_impl(c, rpy_in, omega, v_body, foot_pos, contact, sched, cmd, Pr, qr, var_step, var_leg);
  ws_release(mark);
  return r;
}

/* ------------------------------------------------------------------------------------ */
/* exact QP: Goldfarb-Idnani dual active set, dense, own implementation.                 */
/* Stand-in for qpOASES [UPSTREAM-RECALL mpc_osqp.cc, QPOASES branch]; both return the    */
/* unique minimiser of the strictly convex QP.                                           */
/* constraint id = 6*block + type: 0: -fx+mu fz>=0  1: fx+mu fz>=0  2: -fy+mu fz>=0        */
/*                                 3: fy+mu fz>=0  4: fz-fmin>=0   5: fmax-fz>=0          */
/* (the REF cone rows' upper bound (mu+1) fz_max can never be active inside the box)      */
/* ------------------------------------------------------------------------------------ */
typedef struct { int i0, i1; double v0, v1, c0; } crow;
/* conv_friction_rows with unequal coefficients: the coefficient of cone row ty (-fx, +fx, -fy, +fy) of EVERY block, set by
 * orc_step around its QP solve; NULL: the block's (= its leg's) coefficient mu_blk[b].  Per thread, like the workspace. */
static __thread const double *g_mu_rows = NULL;
static crow make_row(int id, const double *mu_blk, double fz_lo, double fz_hi) {
  int b = id / 6, ty = id % 6; crow r; double mu = (g_mu_rows && ty < 4) ? g_mu_rows[ty] : mu_blk[b];
  switch (ty) {
    case 0: r.i0 = 3 * b;     r.v0 = -1; r.i1 = 3 * b + 2; r.v1 = mu; r.c0 = 0; break;
    case 1: r.i0 = 3 * b;     r.v0 = 1;  r.i1 = 3 * b + 2; r.v1 = mu; r.c0 = 0; break;
    case 2: r.i0 = 3 * b + 1; r.v0 = -1; r.i1 = 3 * b + 2; r.v1 = mu; r.c0 = 0; break;
    case 3: r.i0 = 3 * b + 1; r.v0 = 1;  r.i1 = 3 * b + 2; r.v1 = mu; r.c0 = 0; break;
    case 4: r.i0 = 3 * b + 2; r.v0 = 1;  r.i1 = 3 * b + 2; r.v1 = 0;  r.c0 = -fz_lo; break;
    default: r.i0 = 3 * b + 2; r.v0 = -1; r.i1 = 3 * b + 2; r.v1 = 0; r.c0 = fz_hi; break;
  }
  return r;
}
static double row_eval(const crow *r, const double *x) { return r->v0 * x[r->i0] + r->v1 * x[r->i1] + r->c0; }

static void gi_delete(int n, double *R, double *J, int *A, double *u, int *iq, int l) {
  int q = *iq, qq = -1;
  for (int i = 0; i < q; i++) if (A[i] == l) { qq = i; break; }
  if (qq < 0) return;
  for (int i = qq; i < q - 1; i++) {
    A[i] = A[i + 1]; u[i] = u[i + 1];
    for (int j = 0; j < n; j++) R[j * n + i] = R[j * n + i + 1];
  }
  A[q - 1] = A[q]; u[q - 1] = u[q]; A[q] = 0; u[q] = 0;
  for (int j = 0; j < q; j++) R[j * n + q - 1] = 0;
  q--; *iq = q;
  if (q == 0) return;
  for (int j = qq; j < q; j++) {
    double cc = R[j * n + j], ss = R[(j + 1) * n + j], h = hypot(cc, ss);
    if (h == 0.0) continue;
    cc /= h; ss /= h; R[(j + 1) * n + j] = 0;
    if (cc < 0) { R[j * n + j] = -h; cc = -cc; ss = -ss; } else R[j * n + j] = h;
    double xny = ss / (1.0 + cc);
    for (int k = j + 1; k < q; k++) {
      double t1 = R[j * n + k], t2 = R[(j + 1) * n + k];
      R[j * n + k] = t1 * cc + t2 * ss; R[(j + 1) * n + k] = xny * (t1 + R[j * n + k]) - t2;
    }
    for (int k = 0; k < n; k++) {
      double t1 = J[k * n + j], t2 = J[k * n + j + 1];
      J[k * n + j] = t1 * cc + t2 * ss; J[k * n + j + 1] = xny * (J[k * n + j] + t1) - t2;
    }
  }
}
static int gi_add(int n, double *R, double *J, double *d, int *iq, double *rnorm) {
  int q = *iq;
  for (int j = n - 1; j >= q + 1; j--) {
    double cc = d[j - 1], ss = d[j], h = hypot(cc, ss);
    if (h == 0.0) continue;
    d[j] = 0; ss /= h; cc /= h;
    if (cc < 0) { cc = -cc; ss = -ss; d[j - 1] = -h; } else d[j - 1] = h;
    double xny = ss / (1.0 + cc);
    for (int k = 0; k < n; k++) {
      double t1 = J[k * n + j - 1], t2 = J[k * n + j];
      J[k * n + j - 1] = t1 * cc + t2 * ss; J[k * n + j] = xny * (t1 + J[k * n + j - 1]) - t2;
    }
  }
  q++; *iq = q;
  for (int i = 0; i < q; i++) R[i * n + q - 1] = d[i];
  if (fabs(d[q - 1]) <= 2.2e-16 * (*rnorm)) return 0;
  if (fabs(d[q - 1]) > *rnorm) *rnorm = fabs(d[q - 1]);
  return 1;
}

/* ---- CPU-baseline variant B1: the SAME over-relaxed ADMM as the GPU kernels, a fixed number of
 * iterations, dense linear algebra: Cholesky of P + rho I once, then two triangular solves and the exact pyramid projection
 * per iteration.  Selected process-wide by orc_set_qp_mode(1, ...); used only by bench.py's cpu_baseline variants -- the parity
 * oracle is the exact solver below (mode 0, the default). */
static int g_qp_mode = 0, g_admm_iters = 50;
static double g_admm_rho = 1e-4, g_admm_relax = 1.8;
void orc_set_qp_mode(int mode, int admm_iters, double rho, double relax) { g_qp_mode = mode; g_admm_iters = admm_iters; g_admm_rho = rho; g_admm_relax = relax; }

/* Euclidean projection of (a, b, c) onto { |x| <= mu z, |y| <= mu z, lo <= z <= hi } */
static void project_pyramid(double a, double b, double c, double mu, double lo, double hi, double *o) {
  double aa = fabs(a), bb = fabs(b), mn = aa < bb ? aa : bb, mx = aa < bb ? bb : aa;
  double zA = (c + mu * (aa + bb)) / (1.0 + 2.0 * mu * mu), zB = (c + mu * mx) / (1.0 + mu * mu);
  double z = (mu * zA < mn) ? zA : ((mu * zB < mx) ? zB : c);
  z = z < lo ? lo : (z > hi ? hi : z);
  double lim = mu * z;
  o[0] = a < -lim ? -lim : (a > lim ? lim : a);
  o[1] = b < -lim ? -lim : (b > lim ? lim : b);
  o[2] = z;
}

static int qp_admm_fixed(int n, const double *P, const double *qv, const double *mu_blk, double fz_lo, double fz_hi, double *x) {
  const int nb = n / 3;
  double *L = ws_alloc(sizeof(double) * n * n), *z = ws_calloc(n, sizeof(double)), *y = ws_calloc(n, sizeof(double)), *u = ws_alloc(sizeof(double) * n);
  for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) L[i * n + j] = P[i * n + j] + (i == j ? g_admm_rho : 0.0);
  for (int j = 0; j < n; j++) {
    double s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (!(s > 0)) { return -1; }
    double d = sqrt(s); L[j * n + j] = d;
    for (int i = j + 1; i < n; i++) { double t = L[i * n + j]; for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k]; L[i * n + j] = t / d; }
  }
  for (int b = 0; b < nb; b++) z[3 * b + 2] = fz_lo;
  for (int it = 0; it < g_admm_iters; it++) {
    for (int i = 0; i < n; i++) { double t = g_admm_rho * (z[i] - y[i]) - qv[i]; for (int k = 0; k < i; k++) t -= L[i * n + k] * u[k]; u[i] = t / L[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double t = u[i]; for (int k = i + 1; k < n; k++) t -= L[k * n + i] * u[k]; u[i] = t / L[i * n + i]; }
    for (int b = 0; b < nb; b++) {
      double w[3], pz[3];
      for (int a = 0; a < 3; a++) w[a] = g_admm_relax * u[3 * b + a] + (1.0 - g_admm_relax) * z[3 * b + a] + y[3 * b + a];
      project_pyramid(w[0], w[1], w[2], mu_blk[b], fz_lo, fz_hi, pz);
      for (int a = 0; a < 3; a++) { y[3 * b + a] = w[a] - pz[a]; z[3 * b + a] = pz[a]; }
    }
  }
  memcpy(x, z, sizeof(double) * n);
  return g_admm_iters;
}

static int orc_qp_solve_impl(int n, const double *P, const double *qv, const double *mu_blk, double fz_lo, double fz_hi, double *x, double kkt[3]) {
  const int nb = n / 3, m = 6 * nb;
  if (n == 0) { if (kkt) kkt[0] = kkt[1] = kkt[2] = 0; return 0; }
  if (g_qp_mode == 1) { if (kkt) kkt[0] = kkt[1] = kkt[2] = 0; if (g_mu_rows) return -1; /* per-row coefficients: an asymmetric pyramid, exact solver only */ return qp_admm_fixed(n, P, qv, mu_blk, fz_lo, fz_hi, x); }
  double *L = ws_alloc(sizeof(double) * n * n), *J = ws_calloc((size_t)n * n, sizeof(double)), *R = ws_calloc((size_t)n * n, sizeof(double));
  double *d = ws_alloc(sizeof(double) * n), *z = ws_alloc(sizeof(double) * n), *r = ws_alloc(sizeof(double) * n);
  double *u = ws_calloc((size_t)m + 1, sizeof(double)), *s = ws_alloc(sizeof(double) * m);
  double *x_old = ws_alloc(sizeof(double) * n), *u_old = ws_alloc(sizeof(double) * (m + 1));
  int *A = ws_calloc((size_t)m + 1, sizeof(int)), *A_old = ws_alloc(sizeof(int) * (m + 1)), *iai = ws_alloc(sizeof(int) * m), *excl = ws_alloc(sizeof(int) * m);
  int iter = 0, ret = -1;
  /* Cholesky P = L L' */
  memcpy(L, P, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    double sdiag = L[j * n + j];
    for (int k = 0; k < j; k++) sdiag -= L[j * n + k] * L[j * n + k];
    if (sdiag <= 0) goto done;
    double lj = sqrt(sdiag); L[j * n + j] = lj;
    for (int i = j + 1; i < n; i++) {
      double v = L[i * n + j];
      for (int k = 0; k < j; k++) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / lj;
    }
  }
  /* J = L^-T : solve L' J = I column by column */
  for (int col = 0; col < n; col++) {
    for (int i = n - 1; i >= 0; i--) {
      double v = (i == col) ? 1.0 : 0.0;
      for (int k = i + 1; k < n; k++) v -= L[k * n + i] * J[k * n + col];
      J[i * n + col] = v / L[i * n + i];
    }
  }
  /* x = -P^-1 q  via L */
  for (int i = 0; i < n; i++) { double v = -qv[i]; for (int k = 0; k < i; k++) v -= L[i * n + k] * z[k]; z[i] = v / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double v = z[i]; for (int k = i + 1; k < n; k++) v -= L[k * n + i] * x[k]; x[i] = v / L[i * n + i]; }
  int iq = 0; double rnorm = 1.0;
  const double tol = 1e-10;
  for (int i = 0; i < m; i++) iai[i] = i;
  int ip = 0;
  for (;;) { /* l1 */
    if (++iter > 20 * m + 100) goto done;
    for (int i = 0; i < m; i++) iai[i] = i;
    for (int i = 0; i < iq; i++) iai[A[i]] = -1;
    double psi = 0;
    for (int i = 0; i < m; i++) { crow rw = make_row(i, mu_blk, fz_lo, fz_hi); excl[i] = 1; s[i] = row_eval(&rw, x); if (s[i] < 0) psi += s[i]; }
    if (fabs(psi) <= tol) { ret = iter; break; }
    memcpy(x_old, x, sizeof(double) * n); memcpy(u_old, u, sizeof(double) * (m + 1)); memcpy(A_old, A, sizeof(int) * (m + 1));
    int iq_old = iq;
  l2:;
    double ss = 0; ip = -1;
    for (int i = 0; i < m; i++) if (s[i] < ss && iai[i] != -1 && excl[i]) { ss = s[i]; ip = i; }
    if (ip < 0 || ss >= -tol) { ret = iter; break; }
    crow np = make_row(ip, mu_blk, fz_lo, fz_hi);
    u[iq] = 0; A[iq] = ip;
    for (;;) { /* l2a */
      for (int j = 0; j < n; j++) d[j] = J[np.i0 * n + j] * np.v0 + J[np.i1 * n + j] * np.v1;
      for (int i = 0; i < n; i++) { double v = 0; for (int j = iq; j < n; j++) v += J[i * n + j] * d[j]; z[i] = v; }
      for (int i = iq - 1; i >= 0; i--) { double v = d[i]; for (int j = i + 1; j < iq; j++) v -= R[i * n + j] * r[j]; r[i] = v / R[i * n + i]; }
      int l = -1; double t1 = INFINITY, t2 = INFINITY;
      for (int k = 0; k < iq; k++) if (r[k] > 0.0 && u[k] / r[k] < t1) { t1 = u[k] / r[k]; l = A[k]; }
      double zz = 0; for (int i = 0; i < n; i++) zz += z[i] * z[i];
      double znp = np.v0 * z[np.i0] + np.v1 * z[np.i1];
      if (zz > 1e-300 && fabs(znp) > 1e-300) t2 = -s[ip] / znp;
      double t = t1 < t2 ? t1 : t2;
      if (isinf(t)) goto done; /* infeasible: cannot happen (box nonempty) */
      if (isinf(t2)) {
        for (int k = 0; k < iq; k++) u[k] -= t * r[k];
        u[iq] += t; iai[l] = l; gi_delete(n, R, J, A, u, &iq, l);
        continue;
      }
      for (int i = 0; i < n; i++) x[i] += t * z[i];
      for (int k = 0; k < iq; k++) u[k] -= t * r[k];
      u[iq] += t;
      if (t == t2) {
        if (!gi_add(n, R, J, d, &iq, &rnorm)) {
          /* degenerate: back out and exclude ip */
          excl[ip] = 0; gi_delete(n, R, J, A, u, &iq, ip);
          for (int i = 0; i < m; i++) iai[i] = i;
          for (int i = 0; i < iq_old; i++) { A[i] = A_old[i]; iai[A[i]] = -1; }
          memcpy(u, u_old, sizeof(double) * (m + 1)); memcpy(x, x_old, sizeof(double) * n);
          iq = iq_old; /* NOTE: R,J may be rotated; acceptable since we only restore within the same column space */
          goto l2;
        }
        iai[ip] = -1;
        break; /* -> l1 */
      }
      iai[l] = l; gi_delete(n, R, J, A, u, &iq, l);
      s[ip] = row_eval(&np, x);
    }
  }
  if (kkt) {
    /* KKT residuals with the final multipliers */
    double st = 0, pf = 0, cs = 0;
    double *g = ws_alloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) { double v = qv[i]; for (int j = 0; j < n; j++) v += P[(size_t)i * n + j] * x[j]; g[i] = v; }
    for (int k = 0; k < iq; k++) { crow rw = make_row(A[k], mu_blk, fz_lo, fz_hi); g[rw.i0] -= u[k] * rw.v0; g[rw.i1] -= u[k] * rw.v1; double sv = row_eval(&rw, x); if (fabs(u[k] * sv) > cs) cs = fabs(u[k] * sv); if (u[k] < -1e-9) cs = fmax(cs, -u[k]); }
    for (int i = 0; i < n; i++) if (fabs(g[i]) > st) st = fabs(g[i]);
    for (int i = 0; i < m; i++) { crow rw = make_row(i, mu_blk, fz_lo, fz_hi); double sv = row_eval(&rw, x); if (-sv > pf) pf = -sv; }
    kkt[0] = st; kkt[1] = pf; kkt[2] = cs;
  }
done:
  return ret;
}
/* ... with mu_rows[t] the coefficient of cone row t (-fx, +fx, -fy, +fy) of every block (conv_friction_rows) */
int orc_qp_solve(int n, const double *P, const double *qv, const double *mu_blk, double fz_lo, double fz_hi, double *x, double kkt[3]);
int orc_qp_solve_rows(int n, const double *P, const double *qv, const double *mu_rows, double fz_lo, double fz_hi, double *x, double kkt[3]) {
  double *mu_blk = (double *)malloc(sizeof(double) * (size_t)(n / 3 + 1));
  for (int b = 0; b < n / 3; b++) mu_blk[b] = mu_rows[0];
  g_mu_rows = mu_rows;
  const int r = orc_qp_solve(n, P, qv, mu_blk, fz_lo, fz_hi, x, kkt);
  g_mu_rows = NULL;
  free(mu_blk);
  return r;
}
int orc_qp_solve(int n, const double *P, const double *qv, const double *mu_blk, double fz_lo, double fz_hi, double *x, double kkt[3]) {
  const size_t mark = ws_mark();   /* per-thread workspace: everything this call allocates is released on return */
  const int r = orc_qp_solve_impl(n, P, qv, mu_blk, fz_lo, fz_hi, x, kkt);
  ws_release(mark);
  return r;
}

/* ------------------------------------------------------------------------------------ */
/* reset / step  [UPSTREAM-RECALL locomotion_controller.LocomotionController]            */
/* ------------------------------------------------------------------------------------ */
void orc_reset(const orc_config *c, orc_state *s, double t_now, const double *foot_pos) {
  memset(s, 0, sizeof(*s));
  s->reset_time = t_now;
  for (int i = 0; i < 4; i++) { s->desired[i] = c->init_state[i]; s->leg_state[i] = c->init_state[i]; s->last_desired[i] = c->init_state[i]; }
  /* RaibertSwingLegController.reset(): _last_leg_state = gait_generator.desired_leg_state is the
   * SAME list object the gait generator then mutates in place, so the first update() after a
   * reset can never see a transition; update() ends with a deepcopy which breaks the alias. */
  s->first_update = 1;
  if (foot_pos) { memcpy(s->latched, foot_pos, sizeof(double) * 12); s->need_latch = 0; }
  else s->need_latch = 1;
}

static void quat_inv_rotate(const double q[4], const double v[3], double o[3]) {
  /* rotate v by the inverse of unit quaternion q=(x,y,z,w): REF pybullet invertTransform +
   * multiplyTransforms as used by com_velocity_estimator / REF robot.py:185-203 */
  double x = -q[0], y = -q[1], z = -q[2], w = q[3];
  double n = x * x + y * y + z * z + w * w; (void)n;
  double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
  o[0] = v[0] + w * tx + (y * tz - z * ty);
  o[1] = v[1] + w * ty + (z * tx - x * tz);
  o[2] = v[2] + w * tz + (x * ty - y * tx);
}

static int orc_step_impl(const orc_config *c, orc_state *s, double t_now, const orc_input *in, orc_output *out) {
  memset(out, 0, sizeof(*out));
  const int H = c->horizon;
  /* kinematics inputs */
  double foot[4][3], jac[4][9];
  for (int leg = 0; leg < 4; leg++) {
    if (c->kin_mode == 1) orc_leg_fk(c, leg, &in->q[3 * leg], foot[leg], jac[leg]);
    else { memcpy(foot[leg], in->foot_pos[leg], sizeof(double) * 3); memcpy(jac[leg], in->jac[leg], sizeof(double) * 9); }
  }
  if (s->need_latch) { memcpy(s->latched, foot, sizeof(double) * 12); s->need_latch = 0; }
  /* ---- update() ---- */
  double t = t_now - s->reset_time;
  orc_gait(c, t, in->contact, s->desired, s->leg_state, s->phase);
  double vf[3];
  for (int a = 0; a < 3; a++) vf[a] = filter_push_div(s, a, c->window, in->v_world[a], c->conv_window_divide);
  s->ring_head = (s->ring_head + 1) % c->window;
  if (s->ring_len < c->window) s->ring_len++;
  quat_inv_rotate(in->quat, vf, s->v_body);
  /* swing update: latch feet at desired STANCE->SWING transitions */
  if (!s->first_update || c->conv_first_latch) {
    for (int leg = 0; leg < 4; leg++)
      if (s->desired[leg] == ORC_SWING && s->desired[leg] != s->last_desired[leg]) memcpy(s->latched[leg], foot[leg], sizeof(double) * 3);
  }
  s->first_update = 0;
  for (int leg = 0; leg < 4; leg++) s->last_desired[leg] = s->desired[leg];
  /* ---- swing get_action ---- */
  double cv[3] = {s->v_body[0], s->v_body[1], 0.0};
  double yaw_dot = in->rpy_rate[2];
  for (int leg = 0; leg < 4; leg++) {
    int ls = s->leg_state[leg];
    if (ls == ORC_STANCE || ls == ORC_EARLY_CONTACT) continue;
    const double *hip = c->hip[leg];
    double tw[3] = {-hip[1], hip[0], 0.0}, target[3], des_h[3] = {0, 0, c->body_height - c->foot_clearance};
    double des_v[3] = {in->cmd[0], in->cmd[1], 0.0};
    for (int a = 0; a < 3; a++) {
      double hv = cv[a] + yaw_dot * tw[a];
      double thv = des_v[a] + in->cmd[2] * tw[a];
      target[a] = (hv * c->stance_duration[leg] / 2 - c->swing_kp[a] * (thv - hv)) - des_h[a] + (a < 2 ? hip[a] : 0.0);
    }
    double fp[3];
    orc_swing_trajectory(s->phase[leg], s->latched[leg], target, c->max_clearance, fp);
    memcpy(out->foot_target[leg], fp, sizeof(fp));
    double qo[3];
    orc_leg_ik(c, leg, fp, &in->q[3 * leg], qo);
    for (int j = 0; j < 3; j++) { s->swing_q[3 * leg + j] = qo[j]; s->swing_valid[3 * leg + j] = 1; }
  }
  /* ---- stance get_action ---- */
  int contact[4];
  for (int leg = 0; leg < 4; leg++) contact[leg] = (s->desired[leg] == ORC_STANCE || s->desired[leg] == ORC_EARLY_CONTACT);
  int nc_guess = 0; for (int i = 0; i < 4; i++) nc_guess += contact[i];
  const int rows_differ = c->conv_friction_rows && !(c->mu[0] == c->mu[1] && c->mu[1] == c->mu[2] && c->mu[2] == c->mu[3]);
  double grf[12] = {0};
  if (c->contact_lookahead) {
    /* EXTENSION: step k uses the open-loop desired state at t + k dt_plan (k = 0 is the current tick) */
    int *sched = ws_alloc(sizeof(int) * 4 * H), nb = 0;
    for (int k = 0; k < H; k++) {
      int des[4];
      if (in->sched_valid) { for (int l = 0; l < 4; l++) des[l] = ((in->sched[l] >> k) & 1) ? ORC_STANCE : ORC_SWING; }   /* caller's schedule */
      else orc_gait_desired(c, t + k * c->dt_plan, des);
      for (int l = 0; l < 4; l++) { sched[k * 4 + l] = (des[l] == ORC_STANCE); nb += sched[k * 4 + l]; }
    }
    for (int l = 0; l < 4; l++) sched[l] = contact[l];
    nb = 0; for (int i = 0; i < 4 * H; i++) nb += sched[i];
    int n = 3 * nb;
    if (n > 0 && nc_guess > 0) {
      double *P = ws_alloc(sizeof(double) * n * n), *qv = ws_alloc(sizeof(double) * n), *u = ws_alloc(sizeof(double) * n), *mu_blk = ws_alloc(sizeof(double) * nb);
      int *vs = ws_alloc(sizeof(int) * nb), *vl = ws_alloc(sizeof(int) * nb);
      orc_mpc_build_sched(c, in->rpy, in->rpy_rate, s->v_body, (const double *)foot, contact, sched, in->cmd, P, qv, vs, vl);
      for (int b2 = 0; b2 < nb; b2++) mu_blk[b2] = c->mu[vl[b2]];   /* the block's leg */
      double mg = c->mass * c->gravity;
      g_mu_rows = rows_differ ? c->mu : NULL;
      out->qp_iters = orc_qp_solve(n, P, qv, mu_blk, mg * c->fz_min_scale, mg * c->fz_max_scale, u, out->kkt);
      g_mu_rows = NULL;
      int bad = out->qp_iters < 0;
      if (!bad) for (int b2 = 0; b2 < nb && vs[b2] == 0; b2++) for (int a = 0; a < 3; a++) grf[3 * vl[b2] + a] = -u[3 * b2 + a];
      if (bad) { return -1; }
    }
  } else {
  int n = 3 * nc_guess * H;
  if (n > 0) {
    double *P = ws_alloc(sizeof(double) * n * n), *qv = ws_alloc(sizeof(double) * n), *u = ws_alloc(sizeof(double) * n), *mu_blk = ws_alloc(sizeof(double) * (n / 3));
    int legs[4];
    int nc = orc_mpc_build(c, in->rpy, in->rpy_rate, s->v_body, (const double *)foot, contact, in->cmd, P, qv, legs, NULL, NULL);
    /* mu[l] is LEG l's friction coefficient (what the name foot_friction_coeffs says).  [UPSTREAM-RECALL]: the package's
     * UpdateConstraintsMatrix is recalled to put friction_coeff[0..3] on the four cone ROWS (-x, +x, -y, +y) of every block
     * instead -- the same thing whenever the four are equal, as in every shipped config (0.45 x 4); with unequal values the
     * reading is a switch (conv_friction_rows): 0 per leg, 1 per cone row. */
    for (int b = 0; b < n / 3; b++) mu_blk[b] = c->mu[legs[b % nc]];   /* blocks are ordered (step, contact-leg slot) */
    double mg = c->mass * c->gravity;
    g_mu_rows = rows_differ ? c->mu : NULL;
    out->qp_iters = orc_qp_solve(n, P, qv, mu_blk, mg * c->fz_min_scale, mg * c->fz_max_scale, u, out->kkt);
    g_mu_rows = NULL;
    if (out->qp_iters < 0) { return -1; }
    for (int l = 0; l < nc; l++) for (int a = 0; a < 3; a++) grf[3 * legs[l] + a] = -u[3 * l + a]; /* negated first step */
  }
  }
  memcpy(out->grf, grf, sizeof(grf));
  for (int leg = 0; leg < 4; leg++) orc_force_to_torque(c, leg, &grf[3 * leg], jac[leg], &out->tau[3 * leg]);
  /* ---- merge: swing tuple if stored and desired == SWING else stance tuple ---- */
  for (int j = 0; j < 12; j++) {
    int leg = j / 3;
    if (s->swing_valid[j] && s->desired[leg] == ORC_SWING) {
      out->action[5 * j + 0] = (float)s->swing_q[j]; out->action[5 * j + 1] = (float)c->motor_kp[j];
      out->action[5 * j + 2] = 0.f; out->action[5 * j + 3] = (float)c->motor_kd[j]; out->action[5 * j + 4] = 0.f;
    } else {
      out->action[5 * j + 0] = 0.f; out->action[5 * j + 1] = 0.f; out->action[5 * j + 2] = 0.f; out->action[5 * j + 3] = 0.f;
      out->action[5 * j + 4] = (float)out->tau[j];
    }
  }
  for (int leg = 0; leg < 4; leg++) { out->desired[leg] = s->desired[leg]; out->leg_state[leg] = s->leg_state[leg]; out->phase[leg] = s->phase[leg]; }
  memcpy(out->v_body, s->v_body, sizeof(double) * 3);
  return 0;
}
int orc_step(const orc_config *c, orc_state *s, double t_now, const orc_input *in, orc_output *out) {
  const size_t mark = ws_mark();   /* per-thread workspace: everything this call allocates is released on return */
  const int r = orc_step_impl(c, s, t_now, in, out);
  ws_release(mark);
  return r;
}

int orc_step_batch_cfgs(const orc_config *cfgs, orc_state *s, int B, double t_now, const orc_input *in, orc_output *out, int nthreads) {
  int bad = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bad)
  for (int b = 0; b < B; b++) if (orc_step(&cfgs[b], &s[b], t_now, &in[b], &out[b])) bad++;
  return bad;
}

int orc_step_batch(const orc_config *c, orc_state *s, int B, double t_now, const orc_input *in, orc_output *out, int nthreads) {
  int bad = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bad)
  for (int b = 0; b < B; b++) if (orc_step(c, &s[b], t_now, &in[b], &out[b])) bad++;
  return bad;
}
