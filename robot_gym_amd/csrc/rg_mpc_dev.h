// rg_mpc_dev.h -- device-side types and math for the batched convex-MPC controller (gfx950).
// Everything here is float64: the condensed QP Hessian has condition number ~4e5 (alpha = 1e-5 against O(1)
// angular terms), which float32 cannot carry to the 1e-4 torque tolerance.  CDNA4 issues v_fma_f64 at 4 cycles
// per wave instruction -- the issue cost one wave pays for a (non-packed) f32 FMA too -- so float64 costs register
// space (two VGPRs per value), not issue slots ("Precision").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RG_MAXH 20
#define RG_WARM_N 256 // doubles per robot for the stored ADMM iterate (n <= 240)
#define RG_EXACT_BIG_WS 32   // stored working sets larger than this run the QP launch's exact body with room for RG_DIRECT_Q constraints (RG_EXACT_Q = 40 is the small body's capacity)
#define RG_REC_N 96  // doubles per robot in the front->QP record
// work lists: [0..4] robots per stance-leg count, [5..9] exact re-solve lists, [10..10+RG_COST_CLASSES) cost
// classes of the fused launch (entry = robot | stance legs << 24), most expensive class first
#define RG_COST_CLASSES 16
#define RG_NLISTS (10 + RG_COST_CLASSES)
#define RG_NCOUNTS 32 // [0..4] list lengths, [7] failures, [8..12] re-solve list lengths, [16..16+RG_COST_CLASSES) class lengths

// record layout (doubles)
#define REC_ROLL 0
#define REC_PITCH 1
#define REC_COMZ 2
#define REC_OMEGA 3
#define REC_VBODY 6
#define REC_CMD 9
#define REC_FEETW 12
#define REC_IWINV 24
#define REC_INVCP 33
#define REC_TANP 34
#define REC_JAC 35
#define REC_SWINGQ 71
#define REC_EMIT 83
#define REC_CONTACT 84
#define REC_SCHED 88    // 4 doubles: per leg, bit k = in contact at horizon step k (look-ahead extension)

struct DevCfg {
  int H, window, kin_mode, ik_iters, admm_iters;
  int accel_from;        // first ADMM iteration at which a vote may extrapolate the iterate along its dominant mode (0 = never)
  double dt, mass, inv_mass, body_height, alpha, mu, fz_min, fz_max, g;   // mu: the friction coefficient when all four legs share one
  double mu4[4];         // per leg (FR, FL, RR, RL); the MU4 kernel instantiations read this (rg_mpc_create picks them when the four differ)
  double Iinv[9];
  double w[13];
  double stance_dur[4], duty[4], init_phase[4];
  int init_state[4];
  double contact_thresh, foot_clearance, max_clearance;
  double swing_kp[3];
  double hip[12];
  double kp[12], kd[12], mdir[12], moff[12];
  double jxyz[36];    // [leg][joint][3]
  double jRf[108];    // [leg][joint][9] fixed rotation of the joint origin (Rz Ry Rx of URDF rpy)
  double jaxis[36];   // normalised
  double tip[12];     // toe_xyz + toe_com
  double base_com[3];
  double ik_damping, ik_max_step;
  double rho, relax;
  double admm_abs_tol;   // admm_tol * m * g  (0 = fixed count): largest force change over admm_check iterations
  double admm_prim_tol;  // 10 * admm_abs_tol: largest |x - z| (unprojected vs projected force) accepted at convergence
  int admm_check, lookahead;
  int solver, warm;
  int exact12;           // 1: one- and two-leg robots run the exact active-set body (their st.iters counts applications of G); 2: every robot does
  int plan, admm_switch; // plan 1: the front kernel sorts robots into cost classes for the fused QP launch (RG_COST_*); admm_switch: first-stage iterations
  double rho2;           // second-stage ADMM rho (0 = single stage)
  double rho34_scale;    // first-stage rho of the wrench-space ADMM body (three / four legs, horizon 10) = rho x this
  double rho_sched_scale; // ... of the schedule body (contact schedules; three / four legs at horizon 20)
  double accel_k[4];     // thresholds of the extrapolation test {accel_cos2 0.9, accel_rmax 0.98, accel_rmin 0.5, accel_rate_cap 0.999}
                         // (rg_mpc_config): read from here (scalar loads) because as literals they were materialised in VGPR pairs at
                         // kernel entry and spilled to scratch by every workgroup
  double audit_tol;      // audit lane: per-robot torque error counted as over tolerance
  double admm_extrap;    // geometric-extrapolation convergence guard, in units of the movement tolerance (+inf = off)
  int conv_feet_rotation, conv_com_height, conv_first_latch, conv_window_divide;   // recall-sensitive conventions (rg_mpc.h); alpha doubling is folded into `alpha`
  double Ntab[RG_MAXH * RG_MAXH];  // N_ab = H - max(a,b)
  double Stab[RG_MAXH * RG_MAXH];  // S_ab = sum_{k>max(a,b)}^{H} (k-a-1/2)(k-b-1/2)
  // (appended: nothing above moves)
  double *as_spill;          // horizon-20 re-solve: per-workgroup slabs of global memory for the rows of the packed (C_A G C_A')^-1 beyond the LDS part (SchedLds::SPILL doubles each); null: none
  int as_spill_audit_base;   // first slab of the audit launch's workgroups (the re-solve launch uses 0 .. its grid - 1)
  int mu_rows;               // rg_mpc_config.conv_friction_rows with unequal coefficients: mu4[t] belongs to cone row t (-x, +x, -y, +y) of EVERY block, not to leg t
};

struct DevState {
  double *reset_time;   // [B]
  int *flags;           // [B] bit0 need_latch, bit1 first_update
  int *last_desired;    // [B] 4 x 1 bit
  float *ring;          // [3][W][B]   (inputs are f32, stored losslessly)
  int *ring_len, *ring_head;  // [B]
  double *fsum, *fcorr; // [3][B]
  double *latched;      // [12][B]
  double *swing_q;      // [12][B]
  int *swing_valid;     // [B] 12-bit mask
  double *ik_in;        // [6][4B] front kernel -> swing IK lanes: the swing foot's target (base frame) and the leg's current joint angles
  int *ik_flag;         // [4B] bit 0: this (robot, leg) swings this tick (its target needs the IK)
  float *cmd;           // [3][B] rg_mpc_set_command copy
  double *rec;          // [B][RG_REC_N]
  float *warm_z, *warm_y;   // [B][RG_WARM_N] previous-tick ADMM iterate (warm start); float32: it is only a starting point
  int *warm_key;        // [B] contact mask the stored iterate / working set belongs to (-1 = none)
  unsigned char *ws_ids; // [B][RG_WS_MAX] exact body: constraint ids of the robot's working set at the end of the previous tick
  int *ws_cnt;          // [B] ... and their number
  int *bins;            // [RG_NLISTS][B] work lists (see RG_NLISTS)
  int *counts;          // [RG_NCOUNTS] list lengths and failure count (see RG_NCOUNTS)
  int *ncs;             // [B] stance-leg count of each robot in the last tick
  int *counts_next;     // [RG_NCOUNTS] the other half of the double-buffered counters: zeroed by the front kernel for the next tick (no memset node)
  int *iters;           // [B] solver iterations of the last tick (ADMM, plus the exact re-solve's if it ran)
  // optional per-robot gait timing (rg_mpc_set_gait), [4][B] each; null = the config-wide gait of DevCfg
  const double *g_stance, *g_duty, *g_phase;
  const int *g_init;
  // audit lane (rg_qp_common.inc: audit_capture / audit_compare).  The ADMM bodies copy the record and the first-step forces
  // of ~audit_k converged robots per tick into slot ring `audit_ring`; the exact bodies re-solve them on a side stream.
  double *audit_rec;    // [RG_AUDIT_RING][RG_AUDIT_SLOTS][RG_REC_N]
  double *audit_f;      // [RG_AUDIT_RING][RG_AUDIT_SLOTS][12] ADMM first-step forces (as in the grf output)
  int *audit_idx;       // [RG_AUDIT_RING][RG_AUDIT_SLOTS] robot | re-solve list (stance legs) << 24
  int *audit_cnt;       // [RG_AUDIT_RING] picks of the tick that owns the ring entry (may exceed RG_AUDIT_SLOTS: the rest is dropped)
  unsigned long long *audit_stat;   // [8] audited, over_tol, max_rel bits, max_rel_elem bits, exact failures, dropped
  // direct routing of persistently hard robots (horizon 10, constant contacts, RG_SOLVER_AUTO): a robot whose QP the exact
  // solver had to take over is, while its contact set stays the same, sent straight to the exact lists by the front kernel
  // (work lists 1..4, unused by the fused plan otherwise) instead of running ADMM to the cap again
  int *hard;            // [B] contact mask + 1 of the tick in which the exact solver last solved the robot (0: none)
  int *hint_host;       // pinned HOST word: exact solves (direct + re-solved) of a recent tick, read by rg_mpc_step without waiting
  int *hint_dev;        // device copy of the value last written there
  int direct_on;        // feature switch for this launch
  int tick;             // rg_mpc_step count (every 16th tick a hard robot tries ADMM again)
  int audit_k;          // expected picks per tick for THIS launch (0: no capture)
  int audit_ring;       // ring entry of this tick
  unsigned audit_seed;  // per-tick hash seed
};
#define RG_AUDIT_RING 4
#define RG_AUDIT_SLOTS 256
// The audit runs on the first tick and then on every RG_AUDIT_PERIOD-th one, with RG_AUDIT_PERIOD x audit_k expected picks: an
// exact solve takes 100-350 us, and one side-stream launch per tick (its solves in parallel, but launches of one stream in
// series) took longer than the tick itself -- the caller's stream then stalled on ring reuse (measured: -20 %).  Same
// audited robots per second, an eighth of the launches.  Measured and not kept: every exact solve on a one-wave body (the
// wrench-space one for three and four legs takes 425 us against 300 us on 256 lanes) with the three stance-leg classes on
// three side streams (each cross-stream event costs the caller's stream ~8 us: -7 % instead of -5 %).
#define RG_AUDIT_PERIOD 8

struct DevIn {
  const float *rpy, *rpy_rate, *v_world, *quat, *q, *foot_pos, *jac, *cmd;
  const int *contact;
  const int *contact_sched;   // [4][B] optional caller-supplied contact schedule (bit k = in contact at horizon step k)
  const double *t_robot;      // [B] optional per-robot clock values (null: every robot at the step's scalar t)
};
struct DevOut {
  float *action, *grf, *tau_stance, *phase, *foot_target, *v_body;
  int *leg_state, *desired_state;
};

// ------------------------------------------------------------------------------------
__device__ __forceinline__ void m3mul(const double *a, const double *b, double *c) {
  double t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) c[i] = t[i];
}
__device__ __forceinline__ void m3vec(const double *a, const double *v, double *o) {
  double t0 = a[0] * v[0] + a[1] * v[1] + a[2] * v[2];
  double t1 = a[3] * v[0] + a[4] * v[1] + a[5] * v[2];
  double t2 = a[6] * v[0] + a[7] * v[1] + a[8] * v[2];
  o[0] = t0; o[1] = t1; o[2] = t2;
}

// sin and cos of a joint angle (|x| up to a few turns): one Cody-Waite reduction by pi/2 and the fdlibm kernel
// polynomials, < 1 ulp on the reduced range.  ~35 instructions instead of the ~150 of the general sincos
// (large-argument path, special cases); the IK calls it three times per Newton step.
__device__ __forceinline__ void sincos_joint(double x, double *sn, double *cs) {
  const double kf = rint(x * 0.63661977236758138);           // 2/pi
  const int q = (int)kf;
  double r = fma(-kf, 1.57079632673412561417e+00, x);        // pi/2 split in three parts (fdlibm pio2_1, _2, _3)
  r = fma(-kf, 6.07710050650619224932e-11, r);
  r = fma(-kf, 2.02226624879595063154e-21, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                            -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                            2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const double s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// Leg forward kinematics + joint-space Jacobian of the toe COM in the base (COM) frame,
// generic 3-revolute URDF chain.  Replaces reference controllers/mpc/kinematics.py:13-30
// and model/robots/robot.py:367-397.
__device__ inline void leg_fk(const DevCfg *c, int leg, const double qm[3], double p[3], double J[9]) {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
  double axw[3][3], org[3][3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double t[3];
    m3vec(R, &c->jxyz[(leg * 3 + j) * 3], t);
    o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
    m3mul(R, &c->jRf[(leg * 3 + j) * 9], R);
    const double *ax = &c->jaxis[(leg * 3 + j) * 3];
    m3vec(R, ax, axw[j]);
    org[j][0] = o[0]; org[j][1] = o[1]; org[j][2] = o[2];
    int m = leg * 3 + j;
    double th = qm[j] * c->mdir[m] + c->moff[m];
    double s, cs;
    sincos_joint(th, &s, &cs);
    double C = 1.0 - cs, x = ax[0], y = ax[1], z = ax[2];
    double Rq[9] = {cs + x * x * C, x * y * C - z * s, x * z * C + y * s,
                    y * x * C + z * s, cs + y * y * C, y * z * C - x * s,
                    z * x * C - y * s, z * y * C + x * s, cs + z * z * C};
    m3mul(R, Rq, R);
  }
  double t[3];
  m3vec(R, &c->tip[leg * 3], t);
  double pf[3] = {o[0] + t[0], o[1] + t[1], o[2] + t[2]};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double d0 = pf[0] - org[j][0], d1 = pf[1] - org[j][1], d2 = pf[2] - org[j][2];
    J[0 * 3 + j] = axw[j][1] * d2 - axw[j][2] * d1;
    J[1 * 3 + j] = axw[j][2] * d0 - axw[j][0] * d2;
    J[2 * 3 + j] = axw[j][0] * d1 - axw[j][1] * d0;
  }
  p[0] = pf[0] - c->base_com[0]; p[1] = pf[1] - c->base_com[1]; p[2] = pf[2] - c->base_com[2];
}

__device__ inline bool solve3(const double *A, const double *b, double *x) {
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  if (det == 0.0) return false;
  double inv = 1.0 / det;
  double i01 = (A[2] * A[7] - A[1] * A[8]) * inv, i02 = (A[1] * A[5] - A[2] * A[4]) * inv;
  double i11 = (A[0] * A[8] - A[2] * A[6]) * inv, i12 = (A[2] * A[3] - A[0] * A[5]) * inv;
  double i21 = (A[1] * A[6] - A[0] * A[7]) * inv, i22 = (A[0] * A[4] - A[1] * A[3]) * inv;
  x[0] = c00 * inv * b[0] + i01 * b[1] + i02 * b[2];
  x[1] = c01 * inv * b[0] + i11 * b[1] + i12 * b[2];
  x[2] = c02 * inv * b[0] + i21 * b[1] + i22 * b[2];
  return true;
}

// Fixed-count damped-Newton IK (dq = J'(JJ' + lambda^2 I)^-1 e), started from the current
// joint angles.  Replaces reference controllers/mpc/kinematics.py:98-133.
__device__ inline void leg_ik(const DevCfg *c, int leg, const double target[3], const double q0[3], double qo[3]) {
  double q[3] = {q0[0], q0[1], q0[2]};
  for (int it = 0; it < c->ik_iters; it++) {
    double p[3], J[9], e[3], A[9], y[3];
    leg_fk(c, leg, q, p, J);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) J[3 * i + j] *= c->mdir[3 * leg + j];
#pragma unroll
    for (int i = 0; i < 3; i++) e[i] = target[i] - p[i];
    // converged to 1e-9 m: the remaining fixed-count iterations of the CPU arithmetic move q by < 1e-8 rad, a tenth of
    // the float32 resolution of the joint angle that goes into the command row (Newton is quadratic here: waiting for
    // 1e-12 m cost one more forward-kinematics pass, 0.5 us of the front kernel)
    if (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] < 1e-18) break;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        A[3 * i + j] = J[3 * i] * J[3 * j] + J[3 * i + 1] * J[3 * j + 1] + J[3 * i + 2] * J[3 * j + 2] + (i == j ? c->ik_damping : 0.0);
    if (!solve3(A, e, y)) break;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double dq = J[j] * y[0] + J[3 + j] * y[1] + J[6 + j] * y[2];
      dq = fmin(fmax(dq, -c->ik_max_step), c->ik_max_step);
      q[j] += dq;
    }
  }
  qo[0] = q[0]; qo[1] = q[1]; qo[2] = q[2];
}

// Friction coefficient of a leg.  MU4 = false (the four coefficients are equal: every shipped robot, reference 0.45 x 4
// [UPSTREAM-RECALL, SURVEY 8a-18]): one wave-uniform value that stays in SGPRs across the solver loops.  MU4 = true: the
// leg's own, a per-lane value -- its own kernel instantiations, so that the uniform case pays no register for it.
template <bool MU4>
__device__ __forceinline__ double leg_mu(const DevCfg *__restrict__ c, const int leg) {
  if constexpr (MU4) return c->mu4[leg & 3]; else return c->mu;
}
// ... of cone row `ty` (0: -fx, 1: +fx, 2: -fy, 3: +fy) of a block of leg `leg`: the exact bodies ask per constraint, so that
// they can also follow the other recalled reading of upstream's four coefficients (conv_friction_rows: one per cone ROW)
template <bool MU4>
__device__ __forceinline__ double row_mu(const DevCfg *__restrict__ c, const int leg, const int ty) {
  if constexpr (MU4) return c->mu4[(c->mu_rows ? ty : leg) & 3]; else return c->mu;
}

// Euclidean projection onto { |x| <= mu z, |y| <= mu z, lo <= z <= hi }.
// kA = 1/(1+2mu^2), kB = 1/(1+mu^2) are hoisted by the caller (no divisions in the ADMM loop).
__device__ __forceinline__ void proj_pyramid(double a, double b, double c, double mu, double lo, double hi,
                                             double kA, double kB, double &x, double &y, double &z) {
  double aa = fabs(a), bb = fabs(b);
  double mn = fmin(aa, bb), mx = fmax(aa, bb);
  double zA = (c + mu * (aa + bb)) * kA;
  double zB = (c + mu * mx) * kB;
  // region tests written without 1/mu:  zA < mn/mu  <=>  mu zA < mn
  double zz = (mu * zA < mn) ? zA : ((mu * zB < mx) ? zB : c);
  zz = fmin(fmax(zz, lo), hi);
  double lim = mu * zz;
  x = fmin(fmax(a, -lim), lim);
  y = fmin(fmax(b, -lim), lim);
  z = zz;
}

// Cross-lane move of a double with a DPP control word (two v_mov_b32_dpp, no LDS):
// 0xB1 quad_perm[1,0,3,2] (lane^1), 0x4E quad_perm[2,3,0,1] (lane^2), 0x141 row_half_mirror
// (7-i inside each 8 lanes), 0x128 row_ror:8 (lane^8 inside each 16 lanes).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_masked(double x) {   // lanes outside ROW_MASK keep x
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}

// Minimum over the 64 lanes of a wave, all in DPP + readlane (an LDS butterfly costs six dependent
// ds_bpermute round trips).  Every lane gets the result.
__device__ __forceinline__ double wave_min_f64(double v) {
  v = fmin(v, dpp_f64<0xB1>(v));             // lane ^ 1
  v = fmin(v, dpp_f64<0x4E>(v));             // lane ^ 2
  v = fmin(v, dpp_f64<0x141>(v));            // 7 - i  : 8 lanes
  v = fmin(v, dpp_f64<0x140>(v));            // 15 - i : 16 lanes
  v = fmin(v, dpp_f64_masked<0x142, 0xA>(v)); // row_bcast15 into rows 1, 3
  v = fmin(v, dpp_f64_masked<0x143, 0xC>(v)); // row_bcast31 into rows 2, 3
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_or_zero(double x) {   // lanes outside ROW_MASK get 0
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes of a wave (same DPP ladder as wave_min_f64).  Every lane gets the result.
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  v += dpp_f64_or_zero<0x142, 0xA>(v);   // rows 1, 3 += lane 15 of the row before
  v += dpp_f64_or_zero<0x143, 0xC>(v);   // rows 2, 3 += lane 31
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// 1/d from v_rcp_f64 plus two Newton steps (every lane computes it redundantly; the IEEE
// division sequence is ~4x longer and sat on the critical path of every pivot step).
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = fma(-d, x, 1.0);
  x = fma(x, e, x);
  e = fma(-d, x, 1.0);
  return fma(x, e, x);
}

// A wave-uniform double pinned into SGPRs (two v_readfirstlane): keeps per-stage constants that come out of VALU
// arithmetic (1 / (alpha + rho) ...) from occupying VGPRs across the solver loops.
__device__ __forceinline__ double uniform_f64(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

__device__ __forceinline__ void neumaier_add(double &sum, double &corr, double v) {
  double ns = sum + v;
  if (fabs(sum) >= fabs(v)) corr += (sum - ns) + v; else corr += (v - ns) + sum;
  sum = ns;
}
