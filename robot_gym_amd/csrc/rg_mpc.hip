// rg_mpc.hip -- MI355X (gfx950) batched convex-MPC gait controller: C-ABI host side + small kernels.
//
// Path: one control tick of robot-gym's MPCController.get_action()
// (reference robot_gym/controllers/mpc/mpc_controller.py:102-106) for B robots.  Launch plan per tick:
//   rg_front_kernel            one lane per (robot, leg), coalesced SoA reads: gait phase, CoM velocity filter,
//                              Raibert swing foothold + trajectory, the stance-QP record, and the work lists
//                              (cost classes predicted from each robot's previous-tick iteration count).
//   rg_qp_fused_kernel<H, ..>  one workgroup per robot (one wave at H = 10 -- 256 lanes for small batches, rg_mpc_config.lane_grid --,
//                              256 lanes at H = 20), every stance-leg count in one launch: closed-form Kronecker QP assembly,
//                              in-register symmetric sweep inverse, over-relaxed friction-pyramid ADMM (force space for 1-2
//                              legs, wrench space for 3-4 legs and for contact schedules), J' f, 60-float action row; the
//                              swing-leg IK lanes ride along as the grid's last workgroups.
//                              RG_SOLVER_HYBRID (default; horizons 10 and 20): one- and two-leg robots are solved EXACTLY in
//                              this launch by a dual active-set body warm-started from the robot's previous working set
//                              (rg_qp_exact_kernel.inc: one wave does the solve, on 256-lane grids the other three serve its
//                              mat-vecs); three and four legs keep the wrench-space ADMM body (H = 20: the schedule body).
//   rg_qp_resolve_kernel / rg_qp_sched_retry_kernel
//                              exact dual active-set re-solve (one body per kernel) of the robots the launch above could
//                              not finish: ADMM at its iteration cap, an exact working set that overflowed.  The same
//                              kernel also serves two side streams: the audit lane (converged ADMM robots re-solved and
//                              compared, rg_qp_common.inc) and the direct lists (persistently hard robots solved next to
//                              the ADMM launch).
// Every accepted configuration has a GPU-tested instantiation; anything else is rejected by rg_mpc_create.
// No MFMA: the per-robot blocks are 6..12 wide and every robot has its own operands.
#include "rg_mpc_dev.h"
#include "../../include/rg_mpc.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <initializer_list>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rg_front_kernel.inc"
#include "rg_qp_common.inc"
#include "rg_qp_tile_kernel.inc"
#include "rg_qp_sym6.inc"
#include "rg_qp_wrench_kernel.inc"
#include "rg_qp_exact_kernel.inc"
#include "rg_qp_sched_kernel.inc"
#include "rg_qp_fused_kernel.inc"

// ------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------
__global__ void rg_reset_kernel(const DevCfg *__restrict__ c, DevState st, const int *idx, const double *t0v, int n, double t0, int B) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int b = idx ? idx[k] : k;
  if (b < 0 || b >= B) return;
  st.reset_time[b] = t0v ? t0v[k] : t0;
  st.flags[b] = 3;
  int ld = 0;
  for (int l = 0; l < 4; l++) ld |= ((st.g_init ? st.g_init[l * B + b] : c->init_state[l]) & 1) << l;
  st.last_desired[b] = ld;
  st.ring_len[b] = 0; st.ring_head[b] = 0;
  for (int a = 0; a < 3; a++) { st.fsum[a * B + b] = 0.0; st.fcorr[a * B + b] = 0.0; }
  st.swing_valid[b] = 0;
  st.warm_key[b] = -1;
  st.hard[b] = 0;    // a reset robot is a fresh robot: ADMM first, no direct routing, no cost prediction from before
  st.iters[b] = 0;
}

// RobotMotorModel.convert_to_torque HYBRID (reference model/robots/simple_motor.py:128-140), for S consecutive
// simulation sub-steps of one control tick: the reference applies the same 60-float command ACTION_REPEAT (10) times,
// each time with fresh joint angles and velocities (core/simulation.py:175-179 -> robot.py:276-307).
// q / qd [S][12][B] -> tau [S][B][12]; the command row is read once per (robot, joint) lane and kept in registers.
__global__ void rg_hybrid_to_torque_kernel(const float *__restrict__ action, const float *__restrict__ q,
                                           const float *__restrict__ qd, float *__restrict__ tau, int B, int S) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * 12) return;
  const int j = e / B, b = e - j * B;   // lane = robot: q / qd loads coalesce
  const float *a = action + (size_t)b * 60 + 5 * j;
  const double qs = a[0], kp = a[1], qds = a[2], kd = a[3], ff = a[4];
  for (int s = 0; s < S; s++) {
    const size_t in = ((size_t)s * 12 + j) * B + b;
    const double t = -1.0 * (kp * ((double)q[in] - qs)) - kd * ((double)qd[in] - qds) + ff;
    tau[((size_t)s * B + b) * 12 + j] = (float)t;
  }
}

// ------------------------------------------------------------------------------------
// host side / C-ABI
// ------------------------------------------------------------------------------------
struct rg_mpc_handle {
  rg_mpc_config cfg;
  DevCfg hcfg;
  DevCfg *dcfg = nullptr;
  DevState st{};
  int B = 0, device = 0;
  int *idx_dev = nullptr;
  double *t0_dev = nullptr;
  int cu_count = 256;
  std::vector<void *> allocs;
  std::string err;
  std::string plan;                 // rg_mpc_plan_description
  // optional per-kernel event timing
  std::vector<hipEvent_t> ev;   // RG_PROF_EV events per profiled step
  int prof_max = 0, prof_n = 0, prof_stride = 1;   // events are recorded on every prof_stride-th step
  long long tick = 0;
  bool fused = false;               // one QP launch for all stance-leg counts, work order = the front kernel's cost classes (every plan except exact + contact schedule)
  bool exact12 = false;             // ... in which one- and two-leg robots run the exact active-set body (RG_SOLVER_HYBRID / RG_SOLVER_ACTIVE_SET, constant contacts)
  bool wide = false;                // ... on 256 lanes per robot instead of one wave (horizon 10, hybrid plan: rg_mpc_config.lane_grid)
  bool mu4 = false;                 // the four legs' friction coefficients differ: the kernel instantiations with a per-lane coefficient (leg_mu)
  int *counts2 = nullptr;           // [2][RG_NCOUNTS] double-buffered work-list counters
  bool auto_retry = false;          // RG_SOLVER_AUTO: robots ADMM left unconverged are re-solved exactly
  int retry_max_nc = 0;             // ... for robots with up to this many stance legs
  double *gait_buf = nullptr;       // [3][4][B] per-robot stance duration / duty factor / initial phase (rg_mpc_set_gait)
  int *gait_init_buf = nullptr;     // [4][B] per-robot initial leg state
  // audit lane: exact re-solves of ~audit_k converged robots per tick on a side stream (rg_qp_common.inc)
  bool audit_on = false;
  hipStream_t audit_stream = nullptr;
  hipEvent_t audit_fused[RG_AUDIT_RING] = {};   // recorded on the caller's stream after the ADMM launch that filled the ring entry
  hipEvent_t audit_done[RG_AUDIT_RING] = {};    // recorded on the side stream after the entry's exact re-solves
  bool audit_inflight[RG_AUDIT_RING] = {};
  long long steps = 0;              // rg_mpc_step calls (ring entry and hash seed of the audit picks)
  bool direct_on = false;           // persistently hard robots go straight to the exact lists (horizon 10, constant contacts, RG_SOLVER_AUTO)
  hipStream_t direct_stream = nullptr;
  hipEvent_t front_done = nullptr, direct_done = nullptr;
  int *hint_host = nullptr;         // pinned: exact solves of a recent tick, written by the end-of-tick launch
  long long direct_launches = 0;    // ticks whose direct lists had their own concurrent launch
};

// Every entry point runs on the handle's device and leaves the calling thread's current device as it found it: a process that
// drives several GPUs (one handle + one stream per device) must not have its device switched under it between
// its own torch / HIP calls.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess && prev >= 0; }
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

static thread_local std::string g_create_err;

// The audit lane's side stream and its events come from a per-device pool and go back to it when a handle is destroyed;
// they are never handed to hipStreamDestroy / hipEventDestroy.  Destroying them corrupted the HOST heap of the process under
// ROCm 7.2: in ~7 % of fresh processes that create and destroy a few handles, some 100 ms after rg_mpc_destroy a 32-bit word
// of a recycled ~900-byte heap block is decremented by one and another one zeroed -- a reference count released and a field
// cleared by the runtime in an object it had already freed (tests/studies/host_corruption_hunt.py caught it as a changed
// element of a numpy input array: a parity "failure" of a robot whose quaternion had lost its w component; with audit_k = 0,
// i.e. without these streams and events, 0 of 60 processes).  A stream with cross-stream event waits behind it seems to be
// what the runtime mishandles; recycling the objects sidesteps it and saves their creation cost per handle.
struct AuditLane {
  hipStream_t stream = nullptr;                                      // audit lane: exact re-solves of converged robots
  hipEvent_t fused[RG_AUDIT_RING] = {}, done[RG_AUDIT_RING] = {};
  hipStream_t direct = nullptr;                                      // direct lists: exact solves of persistently hard robots next to the ADMM launch
  hipEvent_t front_done = nullptr, direct_done = nullptr;
};
static std::mutex g_lane_mu;
static std::map<int, std::vector<AuditLane>> g_lane_pool;   // per device

static hipError_t audit_lane_acquire(int device, AuditLane *lane) {
  {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    std::vector<AuditLane> &pool = g_lane_pool[device];
    if (!pool.empty()) { *lane = pool.back(); pool.pop_back(); return hipSuccess; }
  }
  // lowest priority: the exact re-solves fill the gaps the tick's own launches leave (the tail of the ADMM launch), they must
  // not compete with them for CUs
  int prio_low = 0, prio_high = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&lane->stream, hipStreamNonBlocking, prio_low);
  for (int k = 0; k < RG_AUDIT_RING && e == hipSuccess; k++) {
    e = hipEventCreateWithFlags(&lane->fused[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&lane->done[k], hipEventDisableTiming);
  }
  // highest priority: its few workgroups (a whole CU each) must be placed BEFORE the ADMM launch's 4096, which becomes ready at
  // the same moment -- at the default priority they only found a CU in that launch's tail and the tick waited for them after all
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&lane->direct, hipStreamNonBlocking, prio_high);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&lane->front_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&lane->direct_done, hipEventDisableTiming);
  if (e != hipSuccess) {
    // a half-built lane is neither handed out nor destroyed (destroying is what the pool exists to avoid): it is parked where
    // it stays reachable, and the handle's create fails
    static std::vector<AuditLane> graveyard;
    std::lock_guard<std::mutex> lk(g_lane_mu);
    graveyard.push_back(*lane);
    *lane = AuditLane();
  }
  return e;
}
static void audit_lane_release(int device, const AuditLane &lane) {
  if (!lane.stream) return;
  (void)hipStreamSynchronize(lane.stream);
  (void)hipStreamSynchronize(lane.direct);
  std::lock_guard<std::mutex> lk(g_lane_mu);
  g_lane_pool[device].push_back(lane);
}

#define RG_PROF_EV 11  // [0] step start, [1] front end, [2+2k],[3+2k] QP nc=k+1 start/end, [10] step end
#define HIPCHK(h, call)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      char buf_[512];                                                                      \
      snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (h)->err = buf_;                                                                     \
      return RG_MPC_ERR_HIP;                                                               \
    }                                                                                      \
  } while (0)

static void rot_zyx_host(const double *rpy, double *R) {
  double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
  double m[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                 sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                 -sp, cp * sr, cp * cr};
  memcpy(R, m, sizeof(m));
}

static int build_devcfg(const rg_mpc_config *c, DevCfg *d, std::string &err) {
  memset(d, 0, sizeof(*d));
  if (c->abi_version != RG_MPC_ABI_VERSION) { err = "abi_version mismatch"; return RG_MPC_ERR_INVALID; }
  // the reference cannot set the horizon at all (mpc_controller.py:47-56 passes none: upstream default 10); 10 and 20
  // (BASELINE configs[4]) are the horizons with compiled and GPU-tested solver bodies
  if (c->horizon != 10 && c->horizon != 20) { err = "horizon must be 10 or 20"; return RG_MPC_ERR_INVALID; }
  if (c->reserved0 != 0 || c->reserved2 != 0 || c->reserved3 != 0) { err = "reserved fields must be 0"; return RG_MPC_ERR_INVALID; }
  if (c->lane_grid < 0 || c->lane_grid > 2) { err = "lane_grid must be 0 (by batch), 1 (one wave per robot) or 2 (256 lanes per robot)"; return RG_MPC_ERR_INVALID; }
  for (int v : {c->conv_alpha_doubled, c->conv_feet_rotation, c->conv_com_height, c->conv_first_latch, c->conv_window_divide, c->conv_friction_rows})
    if (v != 0 && v != 1) { err = "convention switches (conv_*) are 0 or 1"; return RG_MPC_ERR_INVALID; }
  if (!(c->accel_cos2 > 0 && c->accel_cos2 < 1) || !(c->accel_rmin > 0 && c->accel_rmin < c->accel_rmax && c->accel_rmax < 1) || !(c->accel_rate_cap > 0 && c->accel_rate_cap < 1)) { err = "extrapolation thresholds out of range: 0 < accel_cos2 < 1, 0 < accel_rmin < accel_rmax < 1, 0 < accel_rate_cap < 1"; return RG_MPC_ERR_INVALID; }
  if (c->audit_k < 0 || c->audit_k > RG_AUDIT_SLOTS / (2 * RG_AUDIT_PERIOD) || !(c->audit_tol > 0)) { err = "audit_k must be in [0, 16] and audit_tol positive"; return RG_MPC_ERR_INVALID; }
  if (!(c->admm_rho2 >= 0) || c->admm_switch < 0 || !(c->admm_extrap >= 0) || c->admm_accel < 0 || !(c->admm_rho34_scale > 0 && c->admm_rho34_scale <= 10) || !(c->admm_rho_sched_scale > 0 && c->admm_rho_sched_scale <= 10)) { err = "bad second-stage / convergence ADMM parameters"; return RG_MPC_ERR_INVALID; }
  if (c->window < 1 || c->window > 64) { err = "window out of range [1,64]"; return RG_MPC_ERR_INVALID; }
  for (int i = 0; i < 4; i++) if (!(c->mu[i] > 0 && c->mu[i] <= 100.0)) { err = "friction coefficients must be positive (and finite)"; return RG_MPC_ERR_INVALID; }
  const bool mu_differ = !(c->mu[0] == c->mu[1] && c->mu[1] == c->mu[2] && c->mu[2] == c->mu[3]);
  if (c->conv_friction_rows && mu_differ && c->solver != RG_SOLVER_ACTIVE_SET) { err = "conv_friction_rows = 1 with unequal friction coefficients (one per cone row: an asymmetric pyramid) needs solver = RG_SOLVER_ACTIVE_SET -- the ADMM bodies project onto a symmetric pyramid"; return RG_MPC_ERR_INVALID; }
  if (!(c->mass > 0) || !(c->dt_plan > 0) || !(c->alpha > 0)) { err = "mass, dt_plan and alpha must be positive"; return RG_MPC_ERR_INVALID; }
  if (c->kin_mode != 0 && c->kin_mode != 1) { err = "kin_mode must be 0 or 1"; return RG_MPC_ERR_INVALID; }
  if (c->solver != RG_SOLVER_ADMM && c->solver != RG_SOLVER_ACTIVE_SET && c->solver != RG_SOLVER_AUTO && c->solver != RG_SOLVER_HYBRID) { err = "unsupported solver"; return RG_MPC_ERR_INVALID; }
  if (c->solver == RG_SOLVER_ACTIVE_SET && c->horizon != 10) { err = "RG_SOLVER_ACTIVE_SET (every robot solved exactly) needs horizon 10"; return RG_MPC_ERR_INVALID; }
  if (!(c->admm_rho > 0) || c->admm_iters < 1 || !(c->admm_relax > 0 && c->admm_relax < 2) || !(c->admm_tol >= 0) || (c->admm_tol > 0 && c->admm_check < 1)) { err = "bad ADMM parameters"; return RG_MPC_ERR_INVALID; }
  for (int i = 0; i < 12; i++) if (!(c->motor_dir[i] == 1.0 || c->motor_dir[i] == -1.0)) { err = "motor_dir must be +-1"; return RG_MPC_ERR_INVALID; }
  for (int i = 0; i < 4; i++) {
    if (!(c->duty_factor[i] > 0 && c->duty_factor[i] <= 1) || !(c->stance_duration[i] > 0)) { err = "bad gait timing"; return RG_MPC_ERR_INVALID; }
    if (c->init_state[i] != RG_LEG_SWING && c->init_state[i] != RG_LEG_STANCE) { err = "init_state must be SWING or STANCE"; return RG_MPC_ERR_INVALID; }
  }
  d->H = c->horizon; d->window = c->window; d->kin_mode = c->kin_mode; d->ik_iters = c->ik_iters; d->admm_iters = c->admm_iters; d->accel_from = c->admm_accel; d->accel_k[0] = c->accel_cos2; d->accel_k[1] = c->accel_rmax; d->accel_k[2] = c->accel_rmin; d->accel_k[3] = c->accel_rate_cap; d->audit_tol = c->audit_tol;
  d->dt = c->dt_plan; d->mass = c->mass; d->inv_mass = 1.0 / c->mass; d->body_height = c->body_height;
  d->alpha = c->conv_alpha_doubled ? 2.0 * c->alpha : c->alpha;   // P = 2 (B'WB + alpha I) is the default form with twice the regulariser
  d->conv_feet_rotation = c->conv_feet_rotation; d->conv_com_height = c->conv_com_height; d->conv_first_latch = c->conv_first_latch; d->conv_window_divide = c->conv_window_divide;
  d->mu = c->mu[0]; d->g = c->gravity;
  for (int i = 0; i < 4; i++) d->mu4[i] = c->mu[i];   // (per leg -- per cone row under conv_friction_rows: read by the MU4 kernel instantiations only)
  d->mu_rows = (c->conv_friction_rows && mu_differ) ? 1 : 0;
  d->fz_min = c->mass * c->gravity * c->fz_min_scale; d->fz_max = c->mass * c->gravity * c->fz_max_scale;
  {
    const double *I = c->inertia;
    double c00 = I[4] * I[8] - I[5] * I[7], c01 = I[5] * I[6] - I[3] * I[8], c02 = I[3] * I[7] - I[4] * I[6];
    double det = I[0] * c00 + I[1] * c01 + I[2] * c02;
    if (!(det > 0)) { err = "inertia not positive definite"; return RG_MPC_ERR_INVALID; }
    double inv = 1.0 / det;
    d->Iinv[0] = c00 * inv; d->Iinv[1] = (I[2] * I[7] - I[1] * I[8]) * inv; d->Iinv[2] = (I[1] * I[5] - I[2] * I[4]) * inv;
    d->Iinv[3] = c01 * inv; d->Iinv[4] = (I[0] * I[8] - I[2] * I[6]) * inv; d->Iinv[5] = (I[2] * I[3] - I[0] * I[5]) * inv;
    d->Iinv[6] = c02 * inv; d->Iinv[7] = (I[1] * I[6] - I[0] * I[7]) * inv; d->Iinv[8] = (I[0] * I[4] - I[1] * I[3]) * inv;
  }
  memcpy(d->w, c->weights, sizeof(d->w));
  for (int i = 0; i < 4; i++) { d->stance_dur[i] = c->stance_duration[i]; d->duty[i] = c->duty_factor[i]; d->init_phase[i] = c->init_phase[i]; d->init_state[i] = c->init_state[i]; }
  d->contact_thresh = c->contact_phase_thresh; d->foot_clearance = c->foot_clearance; d->max_clearance = c->max_clearance;
  memcpy(d->swing_kp, c->swing_kp, sizeof(d->swing_kp));
  memcpy(d->hip, c->hip, sizeof(d->hip));
  memcpy(d->kp, c->motor_kp, sizeof(d->kp)); memcpy(d->kd, c->motor_kd, sizeof(d->kd));
  memcpy(d->mdir, c->motor_dir, sizeof(d->mdir)); memcpy(d->moff, c->motor_off, sizeof(d->moff));
  memcpy(d->jxyz, c->jxyz, sizeof(d->jxyz));
  for (int lj = 0; lj < 12; lj++) {
    rot_zyx_host(&c->jrpy[3 * lj], &d->jRf[9 * lj]);
    const double *a = &c->jaxis[3 * lj];
    double nrm = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (c->kin_mode == 1 && !(nrm > 0)) { err = "zero joint axis in chain model"; return RG_MPC_ERR_INVALID; }
    for (int k = 0; k < 3; k++) d->jaxis[3 * lj + k] = nrm > 0 ? a[k] / nrm : 0.0;
  }
  for (int i = 0; i < 12; i++) d->tip[i] = c->toe_xyz[i] + c->toe_com[i];
  memcpy(d->base_com, c->base_com, sizeof(d->base_com));
  d->ik_damping = c->ik_damping; d->ik_max_step = c->ik_max_step;
  d->rho = c->admm_rho; d->relax = c->admm_relax; d->rho2 = c->admm_rho2; d->rho34_scale = c->admm_rho34_scale; d->rho_sched_scale = c->admm_rho_sched_scale; d->admm_switch = c->admm_switch; d->admm_extrap = c->admm_extrap > 0.0 ? c->admm_extrap : INFINITY;
  d->admm_abs_tol = c->admm_tol * c->mass * c->gravity; d->admm_prim_tol = 10.0 * d->admm_abs_tol; d->admm_check = c->admm_check; d->lookahead = c->contact_lookahead ? 1 : 0; d->solver = (c->solver == RG_SOLVER_ADMM) ? RG_SOLVER_ADMM : RG_SOLVER_AUTO /* device side: do the ADMM bodies hand unconverged robots to the exact re-solve lists */; d->warm = (c->warm_start && !c->contact_lookahead) ? 1 : 0;
  const int H = c->horizon;
  for (int a = 0; a < H; a++)
    for (int b = 0; b < H; b++) {
      int mx = a > b ? a : b;
      d->Ntab[a * H + b] = (double)(H - mx);
      double s = 0;
      for (int k = mx + 1; k <= H; k++) s += ((double)(k - a) - 0.5) * ((double)(k - b) - 0.5);
      d->Stab[a * H + b] = s;
    }
  return RG_MPC_OK;
}

template <typename T>
static int dev_alloc(rg_mpc_handle *h, T **p, size_t count) {
  void *v = nullptr;
  hipError_t e = hipMalloc(&v, count * sizeof(T));
  if (e != hipSuccess) { h->err = std::string("hipMalloc failed: ") + hipGetErrorString(e); return RG_MPC_ERR_ALLOC; }
  (void)hipMemset(v, 0, count * sizeof(T));
  h->allocs.push_back(v);
  *p = (T *)v;
  return RG_MPC_OK;
}

extern "C" {

int rg_mpc_abi_version(void) { return RG_MPC_ABI_VERSION; }
int rg_mpc_config_size(void) { return (int)sizeof(rg_mpc_config); }
const char *rg_mpc_kernel_names(void) { return "rg_front_kernel,rg_qp_fused_kernel,rg_qp_resolve_kernel,rg_qp_sched_kernel,rg_qp_sched_retry_kernel,rg_swing_ik_kernel,rg_hybrid_to_torque_kernel,rg_reset_kernel"; }

const char *rg_mpc_plan_description(const rg_mpc_handle *h) { return h ? h->plan.c_str() : ""; }

const char *rg_mpc_last_error(const rg_mpc_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int rg_mpc_create(const rg_mpc_config *cfg, int32_t batch, int32_t device, rg_mpc_handle **out) {
  if (!cfg || !out || batch < 1) { g_create_err = "null config/out or batch < 1"; return RG_MPC_ERR_INVALID; }
  if (batch > (1 << 24)) { g_create_err = "batch > 2^24 robots per handle (work-list entries pack robot | stance legs << 24)"; return RG_MPC_ERR_INVALID; }
  *out = nullptr;
  rg_mpc_handle *h = new rg_mpc_handle();
  h->cfg = *cfg; h->B = batch; h->device = device;
  // Launch plans: front -> one QP launch over all stance-leg counts -> exact re-solve launch (normally empty).
  //   RG_SOLVER_HYBRID (horizons 10 and 20, constant contacts): exact body for one / two legs, wrench-space ADMM for three /
  //     four (horizon 20: the schedule body with a constant schedule); with a contact schedule it is RG_SOLVER_AUTO.
  //   RG_SOLVER_ACTIVE_SET (horizon 10): the same launch with three / four legs on the wrench-space exact body
  //     (qp_exact_wrench_robot); the re-solve launch only sees a working set that overflowed or a degenerate stance.  With
  //     a contact schedule every robot goes to the re-solve launch directly (front kernel plan 0: list = stance-leg bin).
  const bool as_only = cfg->solver == RG_SOLVER_ACTIVE_SET;
  // (horizon 20: the hybrid plan has its exact body for one / two legs too -- 256 lanes, the solve on one wave of them)
  h->exact12 = ((cfg->solver == RG_SOLVER_HYBRID || as_only) && cfg->horizon == 10 && !cfg->contact_lookahead) ||
               (cfg->solver == RG_SOLVER_HYBRID && cfg->horizon == 20 && !cfg->contact_lookahead);
  h->mu4 = !(cfg->mu[0] == cfg->mu[1] && cfg->mu[1] == cfg->mu[2] && cfg->mu[2] == cfg->mu[3]);
  // lanes per robot of the default plan's QP launch at horizon 10 (rg_mpc_config.lane_grid)
  h->wide = cfg->solver == RG_SOLVER_HYBRID && cfg->horizon == 10 && !cfg->contact_lookahead &&
            (cfg->lane_grid == 2 || (cfg->lane_grid == 0 && batch <= RG_MPC_WIDE_BATCH));
  h->fused = !(as_only && cfg->contact_lookahead);
  h->auto_retry = cfg->solver != RG_SOLVER_ADMM;
  // exact re-solve bodies behind every robot the QP launch hands on: horizon 10 -- one-wave bodies (force-space exact body
  // for one / two legs, the schedule QP's wrench-space active-set body for the rest); horizon 20 -- the 256-lane wrench-space one
  h->retry_max_nc = h->auto_retry ? 4 : 0;
  int rc = build_devcfg(cfg, &h->hcfg, h->err);
  h->hcfg.plan = h->fused ? 1 : 0;
  h->hcfg.exact12 = h->exact12 ? (cfg->solver == RG_SOLVER_ACTIVE_SET ? 2 : 1) : 0;
  if (rc) { g_create_err = h->err; delete h; return rc; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_err = "no HIP device available"; delete h; return RG_MPC_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; delete h; return RG_MPC_ERR_INVALID; }
#define CR(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_err = std::string(#call " failed: ") + hipGetErrorString(e_); rg_mpc_destroy(h); return RG_MPC_ERR_HIP; } } while (0)
#define AL(p, n) do { int r_ = dev_alloc(h, &(p), (n)); if (r_) { g_create_err = h->err; rg_mpc_destroy(h); return r_; } } while (0)
  DeviceScope dev_(device);
  CR(dev_.err);
  hipDeviceProp_t prop;
  CR(hipGetDeviceProperties(&prop, device));
  h->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const size_t B = batch, W = cfg->window;
  // the horizon-20 re-solve keeps the rows of its packed inverse beyond the LDS part in global memory: one slab per workgroup of
  // the re-solve launch, then one per workgroup of the audit launch (sched_retry_grid: the launcher's own grid rule).  Only
  // for handles that can launch it (an exact re-solve behind ADMM, or the audit lane).
  if (cfg->horizon == 20 && (cfg->solver != RG_SOLVER_ADMM || cfg->audit_k > 0)) {
    using Resolve20 = SchedLds<20, 4, true>;
    h->hcfg.as_spill_audit_base = sched_retry_grid(h->cu_count, batch, false, 0);
    AL(h->hcfg.as_spill, (size_t)(h->hcfg.as_spill_audit_base + sched_retry_grid(h->cu_count, batch, true, cfg->audit_k * RG_AUDIT_PERIOD)) * Resolve20::SPILL);
  }
  AL(h->dcfg, 1);
  CR(hipMemcpy(h->dcfg, &h->hcfg, sizeof(DevCfg), hipMemcpyHostToDevice));
  AL(h->st.reset_time, B); AL(h->st.flags, B); AL(h->st.last_desired, B);
  AL(h->st.ring, 3 * W * B); AL(h->st.ring_len, B); AL(h->st.ring_head, B);
  AL(h->st.fsum, 3 * B); AL(h->st.fcorr, 3 * B);
  AL(h->st.latched, 12 * B); AL(h->st.swing_q, 12 * B); AL(h->st.swing_valid, B); AL(h->st.ik_in, 24 * B); AL(h->st.ik_flag, 4 * B);
  AL(h->st.cmd, 3 * B); AL(h->st.rec, B * RG_REC_N);
  if (cfg->warm_start) { AL(h->st.warm_z, B * RG_WARM_N); AL(h->st.warm_y, B * RG_WARM_N); }
  if (cfg->warm_start && h->exact12) { AL(h->st.ws_ids, B * RG_WS_MAX); AL(h->st.ws_cnt, B); }
  AL(h->st.warm_key, B); AL(h->st.bins, RG_NLISTS * B); AL(h->counts2, 2 * RG_NCOUNTS); AL(h->st.iters, B); AL(h->st.ncs, B);
  h->st.counts = h->counts2; h->st.counts_next = h->counts2 + RG_NCOUNTS;
  AL(h->idx_dev, B); AL(h->t0_dev, B);
  AL(h->st.hard, B);
  // direct lists (work lists 1..4 read by rg_qp_resolve_kernel): persistently hard robots of the ADMM bodies, and -- exact
  // solver with a contact schedule -- every robot
  h->direct_on = h->auto_retry && cfg->horizon == 10 && (!cfg->contact_lookahead || !h->fused);
  if (h->direct_on) {
    CR(hipHostMalloc((void **)&h->hint_host, 64, hipHostMallocDefault));
    *h->hint_host = 0;
    void *dp = nullptr;
    CR(hipHostGetDevicePointer(&dp, h->hint_host, 0));
    h->st.hint_host = (int *)dp;
    AL(h->st.hint_dev, 1);
  }
  h->audit_on = h->fused && cfg->audit_k > 0 && !as_only;   // the audit re-solves CONVERGED ADMM robots; the exact plan has none
  if (h->audit_on || h->direct_on) {
    AuditLane lane;
    CR(audit_lane_acquire(device, &lane));
    h->audit_stream = lane.stream;
    for (int k = 0; k < RG_AUDIT_RING; k++) { h->audit_fused[k] = lane.fused[k]; h->audit_done[k] = lane.done[k]; }
    h->direct_stream = lane.direct; h->front_done = lane.front_done; h->direct_done = lane.direct_done;
  }
  if (h->audit_on) {
    AL(h->st.audit_rec, (size_t)RG_AUDIT_RING * RG_AUDIT_SLOTS * RG_REC_N); AL(h->st.audit_f, (size_t)RG_AUDIT_RING * RG_AUDIT_SLOTS * 12);
    AL(h->st.audit_idx, (size_t)RG_AUDIT_RING * RG_AUDIT_SLOTS); AL(h->st.audit_cnt, RG_AUDIT_RING); AL(h->st.audit_stat, 8);
  }
#undef CR
#undef AL
  {
    static const char *solver_name[] = {"admm", "active_set", "auto", "hybrid"};
    char buf[256];
    const int lanes = (cfg->horizon == 20 || h->wide) ? 256 : 64;
    snprintf(buf, sizeof(buf), "solver=%s horizon=%d batch=%d lanes=%d exact12=%d mu=%s schedule=%d audit=%d direct=%d", solver_name[cfg->solver], cfg->horizon, batch, lanes,
             h->exact12 ? 1 : 0, h->mu4 ? "per_leg" : "uniform", cfg->contact_lookahead ? 1 : 0, h->audit_on ? 1 : 0, h->direct_on ? 1 : 0);
    h->plan = buf;
  }
  *out = h;
  int r = rg_mpc_reset(h, nullptr, batch, 0.0, nullptr);
  if (r) { g_create_err = h->err; rg_mpc_destroy(h); *out = nullptr; return r; }
  {
    hipError_t e_ = hipDeviceSynchronize();   // the reset kernel has run: create() reports device faults itself
    if (e_ != hipSuccess) { g_create_err = std::string("create: ") + hipGetErrorString(e_); rg_mpc_destroy(h); *out = nullptr; return RG_MPC_ERR_HIP; }
  }
  return RG_MPC_OK;
}

void rg_mpc_destroy(rg_mpc_handle *h) {
  if (!h) return;
  {
    DeviceScope dev_(h->device);
    if (h->audit_stream) {   // back to the pool, not destroyed (see AuditLane)
      AuditLane lane;
      lane.stream = h->audit_stream;
      for (int k = 0; k < RG_AUDIT_RING; k++) { lane.fused[k] = h->audit_fused[k]; lane.done[k] = h->audit_done[k]; }
      lane.direct = h->direct_stream; lane.front_done = h->front_done; lane.direct_done = h->direct_done;
      audit_lane_release(h->device, lane);
    }
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->hint_host) (void)hipHostFree(h->hint_host);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  }
  delete h;
}

static int reset_impl(rg_mpc_handle *h, const int32_t *idx_host, const double *t0_host, int32_t n, double t0, void *stream) {
  if (!h) return RG_MPC_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  const int *idx = nullptr;
  const double *t0v = nullptr;
  if (t0_host) {
    if (n < 0 || n > h->B) { h->err = "reset: n out of range"; return RG_MPC_ERR_INVALID; }
    if (n == 0) return RG_MPC_OK;
    HIPCHK(h, hipMemcpyAsync(h->t0_dev, t0_host, sizeof(double) * n, hipMemcpyHostToDevice, s));
    t0v = h->t0_dev;
  }
  if (idx_host) {
    if (n < 0 || n > h->B) { h->err = "reset: n out of range"; return RG_MPC_ERR_INVALID; }
    for (int i = 0; i < n; i++) if (idx_host[i] < 0 || idx_host[i] >= h->B) { h->err = "reset: index out of range"; return RG_MPC_ERR_INVALID; }
    if (n == 0) return RG_MPC_OK;
    HIPCHK(h, hipMemcpyAsync(h->idx_dev, idx_host, sizeof(int) * n, hipMemcpyHostToDevice, s));
    idx = h->idx_dev;
  } else if (!t0_host) n = h->B;
  hipLaunchKernelGGL(rg_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, s, h->dcfg, h->st, idx, t0v, n, t0, h->B);
  HIPCHK(h, hipGetLastError());
  if (idx_host || t0_host) HIPCHK(h, hipStreamSynchronize(s)); // staging buffers are reused by the next reset
  return RG_MPC_OK;
}

int rg_mpc_reset(rg_mpc_handle *h, const int32_t *idx_host, int32_t n, double t0, void *stream) {
  return reset_impl(h, idx_host, nullptr, n, t0, stream);
}

int rg_mpc_reset_at(rg_mpc_handle *h, const int32_t *idx_host, const double *t0_host, int32_t n, void *stream) {
  if (h && !t0_host) { h->err = "reset_at: null t0 array"; return RG_MPC_ERR_INVALID; }
  return reset_impl(h, idx_host, t0_host, n, 0.0, stream);
}

int rg_mpc_set_command(rg_mpc_handle *h, const float *cmd, void *stream) {
  if (!h || !cmd) { if (h) h->err = "set_command: null pointer"; return RG_MPC_ERR_INVALID; }
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  HIPCHK(h, hipMemcpyAsync(h->st.cmd, cmd, sizeof(float) * 3 * h->B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return RG_MPC_OK;
}

int rg_mpc_set_gait(rg_mpc_handle *h, const double *stance_duration, const double *duty_factor, const double *init_phase,
                    const int32_t *init_state, void *stream) {
  if (!h) return RG_MPC_ERR_INVALID;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  hipStream_t s = (hipStream_t)stream;
  const size_t n4 = (size_t)4 * h->B;
  if (!stance_duration && !duty_factor && !init_phase && !init_state) {   // back to the config-wide gait
    h->st.g_stance = h->st.g_duty = h->st.g_phase = nullptr; h->st.g_init = nullptr;
    return RG_MPC_OK;
  }
  if (!stance_duration || !duty_factor || !init_phase) { h->err = "set_gait: stance_duration, duty_factor and init_phase go together (init_state is optional)"; return RG_MPC_ERR_INVALID; }
  if (!h->gait_buf) {
    int r_ = dev_alloc(h, &h->gait_buf, 3 * n4); if (r_) return r_;
    r_ = dev_alloc(h, &h->gait_init_buf, n4); if (r_) return r_;
  }
  HIPCHK(h, hipMemcpyAsync(h->gait_buf, stance_duration, sizeof(double) * n4, hipMemcpyDeviceToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->gait_buf + n4, duty_factor, sizeof(double) * n4, hipMemcpyDeviceToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->gait_buf + 2 * n4, init_phase, sizeof(double) * n4, hipMemcpyDeviceToDevice, s));
  if (init_state) HIPCHK(h, hipMemcpyAsync(h->gait_init_buf, init_state, sizeof(int) * n4, hipMemcpyDeviceToDevice, s));
  // validated on the device (a bad entry makes its robot a counted failure every tick, like a non-finite state)
  h->st.g_stance = h->gait_buf; h->st.g_duty = h->gait_buf + n4; h->st.g_phase = h->gait_buf + 2 * n4;
  h->st.g_init = init_state ? h->gait_init_buf : nullptr;
  return RG_MPC_OK;
}

int rg_mpc_step(rg_mpc_handle *h, double t, const rg_mpc_state_ptrs *in, const rg_mpc_out_ptrs *out, void *stream) {
  if (!h || !in || !out) { if (h) h->err = "step: null argument"; return RG_MPC_ERR_INVALID; }
  if (!in->rpy || !in->rpy_rate || !in->v_world || !in->quat || !in->q || !in->contact) { h->err = "step: missing required state pointer"; return RG_MPC_ERR_INVALID; }
  if (h->cfg.kin_mode == 0 && (!in->foot_pos || !in->jac)) { h->err = "step: kin_mode 0 needs foot_pos and jac"; return RG_MPC_ERR_INVALID; }
  if (!out->action) { h->err = "step: action output required"; return RG_MPC_ERR_INVALID; }
  if (in->contact_sched && !h->cfg.contact_lookahead) { h->err = "step: contact_sched needs contact_lookahead = 1 in the config"; return RG_MPC_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  const int B = h->B, H = h->cfg.horizon;
  DevIn di{in->rpy, in->rpy_rate, in->v_world, in->quat, in->q, in->foot_pos, in->jac, in->cmd, in->contact, in->contact_sched, in->t_robot};
  DevOut dout{out->action, out->grf, out->tau_stance, out->phase, out->foot_target, out->v_body, out->leg_state, out->desired_state};
  { int *t_ = h->st.counts; h->st.counts = h->st.counts_next; h->st.counts_next = t_; }   // this tick's counters were zeroed by the previous tick's front kernel
  // audit lane: this tick captures into ring entry steps % RING.  The exact re-solves that read the entry's previous
  // contents were launched RING ticks ago on the side stream; this stream waits for them before the entry is rewritten (an
  // already-signalled event in the steady state: the host runs many ticks ahead of the GPU, so it cannot tell by a query)
  int ring = -1;
  h->st.audit_k = 0;
  if (h->audit_on && (h->steps == 0 || h->steps % RG_AUDIT_PERIOD == RG_AUDIT_PERIOD / 2)) {
    ring = (int)(((h->steps + RG_AUDIT_PERIOD / 2) / RG_AUDIT_PERIOD) % RG_AUDIT_RING);
    if (h->audit_inflight[ring]) HIPCHK(h, hipStreamWaitEvent(s, h->audit_done[ring], 0));
    h->st.audit_k = h->cfg.audit_k * RG_AUDIT_PERIOD; h->st.audit_ring = ring;
    h->st.audit_seed = (unsigned)((unsigned long long)h->steps * 0x632BE5ABull + 0x9E3779B9ull);
  }
  // direct routing of persistently hard robots: the front kernel needs the tick; the exact solves of the direct lists get a
  // launch of their own next to the ADMM launch when a recent tick had exact solves (pinned word, read without waiting: the
  // host runs ticks ahead of the GPU, so this is a hint, and both paths are correct whatever it says)
  h->st.direct_on = h->direct_on ? 1 : 0;
  h->st.tick = (int)(h->steps & 0x7fffffff);
  // (plans with an exact body in the QP launch: the launch's own head workgroups take the one- and two-leg direct lists,
  // rg_qp_fused_kernel.inc RG_DIRECT_HEAD -- no side launch, no hint)
  const bool head = h->direct_on && h->fused && h->exact12 && H == 10 && !h->cfg.contact_lookahead;
  const bool direct_now = !head && h->direct_on && h->fused && *(volatile int *)h->hint_host > 0;
  h->steps++;
  hipEvent_t *pev = (h->prof_n < h->prof_max && (h->tick++ % h->prof_stride) == 0) ? &h->ev[(size_t)h->prof_n * RG_PROF_EV] : nullptr;
  if (pev) HIPCHK(h, hipEventRecord(pev[0], s));
  hipLaunchKernelGGL(rg_front_kernel, dim3((4 * B + 63) / 64), dim3(64), 0, s, h->dcfg, h->st, di, dout, t, B);   // one wave per workgroup: 16 k lanes spread over all CUs
  HIPCHK(h, hipGetLastError());
  if (pev) HIPCHK(h, hipEventRecord(pev[1], s));
  if (direct_now) {
    HIPCHK(h, hipEventRecord(h->front_done, s));
    HIPCHK(h, hipStreamWaitEvent(h->direct_stream, h->front_done, 0));
    HIPCHK(h, launch_qp_resolve_h10(h->mu4, h->dcfg, h->st, dout, B, h->cu_count, h->direct_stream, RETRY_DIRECT));
    HIPCHK(h, hipEventRecord(h->direct_done, h->direct_stream));
    h->direct_launches++;
  }
  // one QP launch over all stance-leg counts, then the (normally empty) exact re-solve lists
  // four events per profiled step: [0] start, [1] front end, [3] QP launch end, [5] re-solve end
  // (a contact schedule puts every robot on the schedule body: its own launch, same work lists; the exact solver with a
  // contact schedule has no QP launch of its own: the front kernel's stance-leg bins are the re-solve launch's direct lists)
  if (h->fused) {
    if (h->cfg.contact_lookahead) HIPCHK(h, launch_qp_sched_any(H, h->mu4, h->dcfg, h->st, dout, B, s));
    else HIPCHK(h, launch_qp_fused_any(H, h->exact12 ? (h->cfg.solver == RG_SOLVER_ACTIVE_SET ? 2 : 1) : 0, h->wide, h->mu4, h->dcfg, h->st, dout, B, s));
  } else {   // no QP launch to carry the swing IK lanes: a launch of their own
    hipLaunchKernelGGL(rg_swing_ik_kernel, dim3((4 * B + 63) / 64), dim3(64), 0, s, h->dcfg, h->st, dout, B);
    HIPCHK(h, hipGetLastError());
  }
  if (pev) HIPCHK(h, hipEventRecord(pev[3], s));
  if (ring >= 0) HIPCHK(h, hipEventRecord(h->audit_fused[ring], s));
  if (direct_now) HIPCHK(h, hipStreamWaitEvent(s, h->direct_done, 0));   // the direct robots' actions are part of this tick
  if (h->auto_retry && H == 10) HIPCHK(h, launch_qp_resolve_h10(h->mu4, h->dcfg, h->st, dout, B, h->cu_count, s, head ? RETRY_AFTER_HEAD : (direct_now ? RETRY_LISTS : RETRY_ALL)));
  else if (h->auto_retry) HIPCHK(h, launch_qp_sched_retry_h20(h->mu4, h->dcfg, h->st, dout, B, h->cu_count, s, 0));
  if (pev) { HIPCHK(h, hipEventRecord(pev[5], s)); h->prof_n++; }
  if (ring >= 0) {
    // the same exact body, in audit mode, over the captured records: side stream, ordered after the ADMM launch only
    HIPCHK(h, hipStreamWaitEvent(h->audit_stream, h->audit_fused[ring], 0));
    if (H == 10) HIPCHK(h, launch_qp_resolve_h10(h->mu4, h->dcfg, h->st, dout, B, h->cu_count, h->audit_stream, RETRY_AUDIT));
    else HIPCHK(h, launch_qp_sched_retry_h20(h->mu4, h->dcfg, h->st, dout, B, h->cu_count, h->audit_stream, 1));
    HIPCHK(h, hipEventRecord(h->audit_done[ring], h->audit_stream));
    h->audit_inflight[ring] = true;
  }
  return RG_MPC_OK;
}

int rg_mpc_step_host(rg_mpc_handle *h, double t, const void *host_slab, void *dev_slab, int64_t slab_bytes,
                     const rg_mpc_state_ptrs *in, const rg_mpc_out_ptrs *out, float *action_host, void *stream) {
  if (!h || !host_slab || !dev_slab || slab_bytes <= 0 || !in || !out) { if (h) h->err = "step_host: null argument"; return RG_MPC_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  {
    DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
    HIPCHK(h, hipMemcpyAsync(dev_slab, host_slab, (size_t)slab_bytes, hipMemcpyHostToDevice, s));
  }
  const int rc = rg_mpc_step(h, t, in, out, stream);
  if (rc) return rc;
  if (action_host) {
    DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
    HIPCHK(h, hipMemcpyAsync(action_host, out->action, sizeof(float) * 60 * (size_t)h->B, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
  }
  return RG_MPC_OK;
}

int rg_mpc_profile_begin(rg_mpc_handle *h, int32_t max_steps) {
  if (!h || max_steps < 1 || max_steps > 100000) { if (h) h->err = "profile_begin: bad max_steps"; return RG_MPC_ERR_INVALID; }
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  while ((int)h->ev.size() < max_steps * RG_PROF_EV) {
    hipEvent_t e;
    // timing-only events: no system-scope fence when they are recorded (a default event costs ~1.3 us more of stream time per record)
    HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    h->ev.push_back(e);
  }
  h->prof_max = max_steps; h->prof_n = 0; h->tick = 0;
  return RG_MPC_OK;
}

const char *rg_mpc_profile_window_names(const rg_mpc_handle *h) {
  if (h && !h->fused) return "rg_front_kernel,rg_swing_ik_kernel,rg_qp_resolve_kernel,-,-,step_total";   // the plan without a QP launch
  if (h && h->cfg.contact_lookahead) return h->cfg.horizon == 10 ? "rg_front_kernel,rg_qp_sched_kernel,rg_qp_resolve_kernel,-,-,step_total" : "rg_front_kernel,rg_qp_sched_kernel,rg_qp_sched_retry_kernel,-,-,step_total";
  return (!h || h->cfg.horizon == 10) ? "rg_front_kernel,rg_qp_fused_kernel,rg_qp_resolve_kernel,-,-,step_total" : "rg_front_kernel,rg_qp_fused_kernel,rg_qp_sched_retry_kernel,-,-,step_total";
}

// robots per stance-leg count of the last tick, from the per-robot record (the work lists may be cost classes)
static int host_bin_counts(rg_mpc_handle *h, int out5[5]) {
  std::vector<int> nc((size_t)h->B);
  HIPCHK(h, hipMemcpy(nc.data(), h->st.ncs, sizeof(int) * (size_t)h->B, hipMemcpyDeviceToHost));
  for (int k = 0; k < 5; k++) out5[k] = 0;
  for (int v : nc) if (v >= 0 && v <= 4) out5[v]++;
  return RG_MPC_OK;
}

int rg_mpc_profile_stride(rg_mpc_handle *h, int32_t stride) {
  if (!h || stride < 1) { if (h) h->err = "profile_stride: stride must be >= 1"; return RG_MPC_ERR_INVALID; }
  h->prof_stride = stride;
  return RG_MPC_OK;
}

int rg_mpc_profile_end(rg_mpc_handle *h, float *avg_ms6, int32_t *robots5, void *stream) {
  if (!h || !avg_ms6) { if (h) h->err = "profile_end: null output"; return RG_MPC_ERR_INVALID; }
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  int n = h->prof_n;
  h->prof_max = 0;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < n; k++) {
    hipEvent_t *e = &h->ev[(size_t)k * RG_PROF_EV];
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, e[0], e[1])); acc[0] += ms;
    HIPCHK(h, hipEventElapsedTime(&ms, e[1], e[3])); acc[1] += ms;
    HIPCHK(h, hipEventElapsedTime(&ms, e[3], e[5])); acc[2] += ms;
    HIPCHK(h, hipEventElapsedTime(&ms, e[0], e[5])); acc[5] += ms;
  }
  for (int j = 0; j < 6; j++) avg_ms6[j] = n > 0 ? (float)(acc[j] / n) : 0.f;
  if (robots5) { int tmp[5]; int r_ = host_bin_counts(h, tmp); if (r_) return r_; for (int k = 0; k < 5; k++) robots5[k] = tmp[k]; }
  return n;
}

// Test hook: fill the LDS of every CU with NaN bit patterns (LDS keeps its contents between kernels), so a
// kernel that reads LDS it never wrote fails deterministically instead of once in a while.
__global__ void rg_debug_poison_lds_kernel(int ndoubles) {
  for (int e = threadIdx.x; e < ndoubles; e += blockDim.x) smem[e] = __longlong_as_double(0x7ff8dead0000beefLL);
  __syncthreads();
  if (smem[(threadIdx.x * 7) % ndoubles] == 0.0) smem[0] = 1.0;   // keep the stores observable
}

int rg_mpc_debug_poison_lds(rg_mpc_handle *h, void *stream) {
  if (!h) return RG_MPC_ERR_INVALID;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  const int bytes = 160 * 1024;
  static bool attr_done[64] = {};
  if (lds_attr_needed(attr_done)) HIPCHK(h, hipFuncSetAttribute((const void *)rg_debug_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  rg_debug_poison_lds_kernel<<<dim3(h->cu_count * 4), dim3(256), bytes, (hipStream_t)stream>>>(bytes / 8);
  HIPCHK(h, hipGetLastError());
  return RG_MPC_OK;
}

int rg_mpc_hybrid_to_torque_substeps(rg_mpc_handle *h, const float *action, const float *q, const float *qd, float *tau, int32_t substeps, void *stream) {
  if (!h || !action || !q || !qd || !tau) { if (h) h->err = "hybrid_to_torque: null pointer"; return RG_MPC_ERR_INVALID; }
  if (substeps < 1 || substeps > 1024) { h->err = "hybrid_to_torque: substeps out of range [1,1024]"; return RG_MPC_ERR_INVALID; }
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  const int total = h->B * 12;
  hipLaunchKernelGGL(rg_hybrid_to_torque_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, action, q, qd, tau, h->B, substeps);
  HIPCHK(h, hipGetLastError());
  return RG_MPC_OK;
}

int rg_mpc_hybrid_to_torque(rg_mpc_handle *h, const float *action, const float *q, const float *qd, float *tau, void *stream) {
  return rg_mpc_hybrid_to_torque_substeps(h, action, q, qd, tau, 1, stream);
}

int rg_mpc_last_solver_stats(rg_mpc_handle *h, int64_t *iters_sum, int32_t *iters_max, int32_t *qp_robots,
                             int32_t *retried, int32_t *failures, void *stream) {
  if (!h) return RG_MPC_ERR_INVALID;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  int cnt[RG_NCOUNTS], per_nc[5];
  HIPCHK(h, hipMemcpy(cnt, h->st.counts, sizeof(cnt), hipMemcpyDeviceToHost));
  { int r_ = host_bin_counts(h, per_nc); if (r_) return r_; }
  if (iters_sum || iters_max) {   // per-robot iteration counts, reduced here (the kernels keep no global atomics for them)
    std::vector<int> it((size_t)h->B);
    HIPCHK(h, hipMemcpy(it.data(), h->st.iters, sizeof(int) * (size_t)h->B, hipMemcpyDeviceToHost));
    int64_t sum = 0; int mx = 0;
    for (int v : it) { sum += v; if (v > mx) mx = v; }
    if (iters_sum) *iters_sum = sum;
    if (iters_max) *iters_max = mx;
  }
  if (qp_robots) *qp_robots = per_nc[1] + per_nc[2] + per_nc[3] + per_nc[4];
  // robots the ADMM pass left unconverged: re-solved exactly where the plan has a re-solve pass, failures otherwise
  int resolved = 0, unresolved = 0;
  for (int nc = 1; nc <= 4; nc++) (nc <= h->retry_max_nc ? resolved : unresolved) += cnt[8 + nc];
  if (h->direct_on) resolved += cnt[1] + cnt[2] + cnt[3] + cnt[4];   // robots the front kernel sent straight to the exact solver
  if (retried) *retried = resolved;
  if (failures) *failures = cnt[7] + unresolved;
  return RG_MPC_OK;
}

int rg_mpc_audit_stats(rg_mpc_handle *h, int64_t *audited, int64_t *over_tol, double *max_rel, double *max_rel_elem,
                       int64_t *exact_failures, int64_t *dropped, int32_t reset, void *stream) {
  if (!h) return RG_MPC_ERR_INVALID;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  unsigned long long st8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (h->audit_on) {
    HIPCHK(h, hipStreamSynchronize(h->audit_stream));
    for (int k = 0; k < RG_AUDIT_RING; k++) h->audit_inflight[k] = false;
    HIPCHK(h, hipMemcpy(st8, h->st.audit_stat, sizeof(st8), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(h, hipMemset(h->st.audit_stat, 0, sizeof(st8)));
  }
  double mr, me;
  memcpy(&mr, &st8[2], sizeof(double)); memcpy(&me, &st8[3], sizeof(double));
  if (audited) *audited = (int64_t)st8[0];
  if (over_tol) *over_tol = (int64_t)st8[1];
  if (max_rel) *max_rel = mr;
  if (max_rel_elem) *max_rel_elem = me;
  if (exact_failures) *exact_failures = (int64_t)st8[4];
  if (dropped) *dropped = (int64_t)st8[5];
  return RG_MPC_OK;
}

int rg_mpc_last_direct_count(rg_mpc_handle *h, int32_t *direct_robots, int64_t *concurrent_launches, void *stream) {
  if (!h) return RG_MPC_ERR_INVALID;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  int cnt[RG_NCOUNTS];
  HIPCHK(h, hipMemcpy(cnt, h->st.counts, sizeof(cnt), hipMemcpyDeviceToHost));
  if (direct_robots) *direct_robots = h->direct_on ? cnt[1] + cnt[2] + cnt[3] + cnt[4] : 0;
  if (concurrent_launches) *concurrent_launches = h->direct_launches;
  return RG_MPC_OK;
}

int rg_mpc_last_iterations(rg_mpc_handle *h, int32_t *iters_B, int32_t *stance_legs_B, void *stream) {
  if (!h || (!iters_B && !stance_legs_B)) return RG_MPC_ERR_INVALID;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  if (iters_B) HIPCHK(h, hipMemcpy(iters_B, h->st.iters, sizeof(int) * (size_t)h->B, hipMemcpyDeviceToHost));
  if (stance_legs_B) HIPCHK(h, hipMemcpy(stance_legs_B, h->st.ncs, sizeof(int) * (size_t)h->B, hipMemcpyDeviceToHost));
  return RG_MPC_OK;
}

int rg_mpc_last_bin_counts(rg_mpc_handle *h, int32_t *out5, void *stream) {
  if (!h || !out5) return RG_MPC_ERR_INVALID;
  DeviceScope dev_(h->device); HIPCHK(h, dev_.err);
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  int tmp[5];
  { int r_ = host_bin_counts(h, tmp); if (r_) return r_; }
  for (int k = 0; k < 5; k++) out5[k] = tmp[k];
  return RG_MPC_OK;
}

} // extern "C"
