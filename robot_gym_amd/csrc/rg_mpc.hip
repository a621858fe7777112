// rg_mpc.hip -- MI355X (gfx950) batched convex-MPC gait controller: kernels + C-ABI.
//
// Path: one control tick of robot-gym's MPCController.get_action()
// (reference robot_gym/controllers/mpc/mpc_controller.py:102-106) for B robots:
//   rg_front_kernel   lane = robot, coalesced SoA reads: gait phase, CoM velocity filter,
//                     Raibert swing foothold + trajectory + IK, stance-QP record, binning
//                     of robots by number of stance legs.
//   rg_qp_admm_kernel one robot per workgroup (1 wave for <=64 QP variables, 2 above):
//                     closed-form condensed QP assembly (Kronecker structure), in-LDS
//                     symmetric sweep inversion of (P + rho I), fixed-count over-relaxed
//                     ADMM with exact friction-pyramid projection, J' f, 60-float action.
// No MFMA: the per-robot blocks are 6..12 wide and every robot has its own operands.
#include "rg_mpc_dev.h"
#include "../../include/rg_mpc.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------
// front kernel
// ------------------------------------------------------------------------------------
// Open-loop gait, bit-exact with the float64 reference arithmetic: no FMA contraction.
__device__ inline void gait_leg(const DevCfg *c, int leg, double t, int contact, int &desired, int &leg_state, double &phase) {
#pragma clang fp contract(off)
  int init = c->init_state[leg];
  int next = (init == RG_LEG_SWING) ? RG_LEG_STANCE : RG_LEG_SWING;
  double ratio = (init == RG_LEG_SWING) ? 1.0 - c->duty[leg] : c->duty[leg];
  double full = c->stance_dur[leg] / c->duty[leg];
  double aug = t + c->init_phase[leg] * full;
  double ph = fmod(aug, full) / full;
  if (ph < ratio) { desired = init; phase = ph / ratio; }
  else { desired = next; phase = (ph - ratio) / (1.0 - ratio); }
  leg_state = desired;
  if (!(phase < c->contact_thresh)) {
    if (leg_state == RG_LEG_SWING && contact) leg_state = RG_LEG_EARLY_CONTACT;
    if (leg_state == RG_LEG_STANCE && !contact) leg_state = RG_LEG_LOSE_CONTACT;
  }
}

// One lane per (robot, leg): the four lanes of a quad share a robot.  Per-leg work (gait state, swing
// target / trajectory / IK, FK, lever arms) runs in parallel; per-robot values are combined with
// quad ballots/shuffles and written by the leg-0 lane.  (A lane-per-robot version left 4096 robots
// on 64 waves with a ~290k-cycle serial chain each.)
// Open-loop desired state of one leg at horizon step k (look-ahead extension): t + k*dt without FMA
// contraction, like the float64 CPU arithmetic.
__device__ inline int gait_desired_at(const DevCfg *c, int leg, double t, int k) {
  const double tk = __dadd_rn(t, __dmul_rn((double)k, c->dt));
  int desired, ls; double ph;
  gait_leg(c, leg, tk, 1, desired, ls, ph);
  return desired;
}

__global__ void __launch_bounds__(256)
rg_front_kernel(const DevCfg *__restrict__ c, DevState st, DevIn in, DevOut out, double t_now, int B) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 2, leg = gid & 3;
  if (b >= B) return;  // B*4 is padded to whole quads by construction (4 lanes per robot)
  const int lane = threadIdx.x & 63, qbase = lane & ~3;
  const int W = c->window;
  // ---- inputs ----
  double rpy[3], rate[3], vw[3], quat[4], q3[3], cmd[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { rpy[i] = in.rpy[i * B + b]; rate[i] = in.rpy_rate[i * B + b]; vw[i] = in.v_world[i * B + b]; }
#pragma unroll
  for (int i = 0; i < 4; i++) quat[i] = in.quat[i * B + b];
#pragma unroll
  for (int i = 0; i < 3; i++) q3[i] = in.q[(3 * leg + i) * B + b];
  const int contact = in.contact[leg * B + b];
  const float *cmdp = in.cmd ? in.cmd : st.cmd;
#pragma unroll
  for (int i = 0; i < 3; i++) cmd[i] = cmdp[i * B + b];
  double foot[3], jac[9];
  if (c->kin_mode == 1) leg_fk(c, leg, q3, foot, jac);
  else {
#pragma unroll
    for (int i = 0; i < 3; i++) foot[i] = in.foot_pos[(3 * leg + i) * B + b];
#pragma unroll
    for (int i = 0; i < 9; i++) jac[i] = in.jac[(9 * leg + i) * B + b];
  }
  const int flags = st.flags[b];
  if (flags & 1) {
#pragma unroll
    for (int i = 0; i < 3; i++) st.latched[(3 * leg + i) * B + b] = foot[i];
  }
  // ---- gait (own leg) ----
  const double t = t_now - st.reset_time[b];
  int desired, lstate;
  double phase;
  gait_leg(c, leg, t, contact, desired, lstate, phase);
  // ---- velocity estimator (all four lanes compute it, leg 0 stores it) ----
  const int rlen = st.ring_len[b], rhead = st.ring_head[b];
  double vf[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double sm = st.fsum[a * B + b], cr = st.fcorr[a * B + b];
    const size_t slot = ((size_t)a * W + rhead) * B + b;
    if (rlen >= W) neumaier_add(sm, cr, -(double)st.ring[slot]);
    neumaier_add(sm, cr, vw[a]);
    vf[a] = (sm + cr) / (double)W;
    // all reads of this robot's filter state happen before leg 0 overwrites it
    __builtin_amdgcn_wave_barrier();
    if (leg == 0) { st.ring[slot] = (float)vw[a]; st.fsum[a * B + b] = sm; st.fcorr[a * B + b] = cr; }
  }
  double vb[3];
  {
    double x = -quat[0], y = -quat[1], z = -quat[2], w = quat[3];
    double tx = 2 * (y * vf[2] - z * vf[1]), ty = 2 * (z * vf[0] - x * vf[2]), tz = 2 * (x * vf[1] - y * vf[0]);
    vb[0] = vf[0] + w * tx + (y * tz - z * ty);
    vb[1] = vf[1] + w * ty + (z * tx - x * tz);
    vb[2] = vf[2] + w * tz + (x * ty - y * tx);
  }
  // ---- swing update: latch at desired STANCE->SWING (skipped on the first update after reset) ----
  const int last = st.last_desired[b];
  if (!(flags & 2) && desired == RG_LEG_SWING && ((last >> leg) & 1) != RG_LEG_SWING) {
#pragma unroll
    for (int a = 0; a < 3; a++) st.latched[(3 * leg + a) * B + b] = foot[a];
  }
  const unsigned long long des_ballot = __ballot(desired == RG_LEG_STANCE);
  const int desired_bits = (int)((des_ballot >> qbase) & 0xF);   // bit l = leg l desired STANCE (== desired value)
  // ---- swing get_action (own leg) ----
  const int valid_old = st.swing_valid[b];
  double swq[3];
#pragma unroll
  for (int j = 0; j < 3; j++) swq[j] = st.swing_q[(3 * leg + j) * B + b];
  double ftarget[3] = {0.0, 0.0, 0.0};
  const bool do_swing = !(lstate == RG_LEG_STANCE || lstate == RG_LEG_EARLY_CONTACT);
  if (do_swing) {
    const double *hip = &c->hip[3 * leg];
    double tw[3] = {-hip[1], hip[0], 0.0};
    double cv[3] = {vb[0], vb[1], 0.0}, dv[3] = {cmd[0], cmd[1], 0.0};
    double dh[3] = {0.0, 0.0, c->body_height - c->foot_clearance};
    double target[3], start[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      double hv = cv[a] + rate[2] * tw[a];
      double thv = dv[a] + cmd[2] * tw[a];
      target[a] = (hv * c->stance_dur[leg] / 2 - c->swing_kp[a] * (thv - hv)) - dh[a] + (a < 2 ? hip[a] : 0.0);
      start[a] = st.latched[(3 * leg + a) * B + b];
    }
    double ph;
    if (phase <= 0.5) ph = 0.8 * sin(phase * M_PI); else ph = 0.8 + (phase - 0.5) * 0.4;
    double fp[3];
    fp[0] = (1 - ph) * start[0] + ph * target[0];
    fp[1] = (1 - ph) * start[1] + ph * target[1];
    {
      double mid = fmax(target[2], start[2]) + c->max_clearance;
      double d1 = mid - start[2], d2 = target[2] - start[2], d3 = 0.25 - 0.5;
      double ca = (d1 - d2 * 0.5) / d3, cb = (d2 * 0.25 - d1) / d3;
      fp[2] = ca * ph * ph + cb * ph + start[2];
    }
    leg_ik(c, leg, fp, q3, swq);
#pragma unroll
    for (int j = 0; j < 3; j++) { st.swing_q[(3 * leg + j) * B + b] = swq[j]; ftarget[j] = fp[j]; }
  }
  const int swing_bits = (int)((__ballot(do_swing) >> qbase) & 0xF);
  int valid = valid_old;
#pragma unroll
  for (int l = 0; l < 4; l++) if ((swing_bits >> l) & 1) valid |= 7 << (3 * l);
  int emit = 0;
#pragma unroll
  for (int j = 0; j < 12; j++) if (((valid >> j) & 1) && !((desired_bits >> (j / 3)) & 1)) emit |= 1 << j;
  // ---- stance record ----
  const int cmask = desired_bits;   // contact for the MPC = desired STANCE
  const int nc = __builtin_popcount(cmask);
  int sched = 0;                    // bit k: this leg in contact at horizon step k
  if (c->lookahead) {
    sched = (cmask >> leg) & 1;
    for (int k = 1; k < c->H; k++) sched |= (gait_desired_at(c, leg, t, k) == RG_LEG_STANCE) << k;
  }
  double sr, cr_, sp, cp;
  sincos(rpy[0], &sr, &cr_);
  sincos(rpy[1], &sp, &cp);
  // own foot -> world-aligned frame with Rx(roll) Ry(pitch)   (yaw zeroed)
  double fw[3];
  fw[0] = cp * foot[0] + sp * foot[2];
  fw[1] = sr * sp * foot[0] + cr_ * foot[1] - sr * cp * foot[2];
  fw[2] = -cr_ * sp * foot[0] + sr * foot[1] + cr_ * cp * foot[2];
  double hz = ((cmask >> leg) & 1) ? fw[2] : 0.0;
  hz += __shfl_xor(hz, 1);
  hz += __shfl_xor(hz, 2);
  double *rec = st.rec + (size_t)b * RG_REC_N;
#pragma unroll
  for (int i = 0; i < 3; i++) { rec[REC_FEETW + 3 * leg + i] = fw[i]; rec[REC_SWINGQ + 3 * leg + i] = swq[i]; }
#pragma unroll
  for (int i = 0; i < 9; i++) rec[REC_JAC + 9 * leg + i] = jac[i];
  rec[REC_SCHED + leg] = (double)sched;
  // look-ahead: every robot with a stance leg now solves the full four-leg problem (blocks of a leg that
  // is not in contact at a step are pinned to zero by the projection)
  const int bin = c->lookahead ? (nc > 0 ? 4 : 0) : nc;
  if (leg == 0) {
    rec[REC_ROLL] = rpy[0]; rec[REC_PITCH] = rpy[1];
    rec[REC_COMZ] = nc > 0 ? fabs(hz / nc) : 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) { rec[REC_OMEGA + i] = rate[i]; rec[REC_VBODY + i] = vb[i]; rec[REC_CMD + i] = cmd[i]; }
    // body rotation for the inertia: Ry(pitch) Rx(roll)
    double Rb[9] = {cp, sp * sr, sp * cr_, 0, cr_, -sr, -sp, cp * sr, cp * cr_};
    double T1[9], Rt[9], Iw[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Rt[3 * i + j] = Rb[3 * j + i];
    m3mul(Rb, c->Iinv, T1);
    m3mul(T1, Rt, Iw);
#pragma unroll
    for (int i = 0; i < 9; i++) rec[REC_IWINV + i] = Iw[i];
    rec[REC_INVCP] = 1.0 / cp;
    rec[REC_TANP] = sp / cp;
    rec[REC_EMIT] = (double)emit;
    rec[REC_CONTACT] = (double)(c->lookahead ? 15 : cmask);
    // per-robot persistent scalars
    st.ring_head[b] = (rhead + 1) % W;
    if (rlen < W) st.ring_len[b] = rlen + 1;
    st.last_desired[b] = desired_bits;
    st.flags[b] = 0;
    st.swing_valid[b] = valid;
    const int slot = atomicAdd(&st.counts[bin], 1);
    st.bins[(size_t)bin * B + slot] = b;
    if (out.v_body)
#pragma unroll
      for (int i = 0; i < 3; i++) out.v_body[b * 3 + i] = (float)vb[i];
  }
  // ---- optional outputs (per leg) ----
  if (out.leg_state) out.leg_state[b * 4 + leg] = lstate;
  if (out.desired_state) out.desired_state[b * 4 + leg] = desired;
  if (out.phase) out.phase[b * 4 + leg] = (float)phase;
  if (out.foot_target)
#pragma unroll
    for (int i = 0; i < 3; i++) out.foot_target[b * 12 + 3 * leg + i] = (float)ftarget[i];
  if (nc == 0) {
    // no stance leg: forces are zero, the action row is complete here
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      float *a = out.action + (size_t)b * 60 + 5 * j;
      if ((emit >> j) & 1) { a[0] = (float)swq[jj]; a[1] = (float)c->kp[j]; a[2] = 0.f; a[3] = (float)c->kd[j]; a[4] = 0.f; }
      else { a[0] = 0.f; a[1] = 0.f; a[2] = 0.f; a[3] = 0.f; a[4] = 0.f; }
      if (out.grf) out.grf[b * 12 + j] = 0.f;
      if (out.tau_stance) out.tau_stance[b * 12 + j] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------
// QP kernel (ADMM).  One robot per workgroup; thread i owns QP variable i = (step a, leg l, axis d),
// i.e. row/column i of the symmetric matrix held in LDS (stored so that a thread's
// "row" is read at consecutive addresses across lanes: element (j, i) at j*LD + i).
// ------------------------------------------------------------------------------------
// Pin C row registers (see pin_row below): ten per empty asm statement.
template <int C>
__device__ __forceinline__ void pin_array(double (&t)[C]) {
#pragma unroll
  for (int i = 0; i + 9 < C; i += 10)
    asm volatile("" : "+v"(t[i]), "+v"(t[i + 1]), "+v"(t[i + 2]), "+v"(t[i + 3]), "+v"(t[i + 4]), "+v"(t[i + 5]), "+v"(t[i + 6]), "+v"(t[i + 7]), "+v"(t[i + 8]), "+v"(t[i + 9]));
#pragma unroll
  for (int i = (C / 10) * 10; i < C; i++) asm volatile("" : "+v"(t[i]));
}

// index of the k-th set bit of a 4-bit contact mask (k-th stance leg) without a scratch array
__device__ __forceinline__ int nth_leg(int mask, int k) {
  int l0 = __builtin_ctz(mask | 16);
  int m1 = mask & (mask - 1);
  int l1 = __builtin_ctz(m1 | 16);
  int m2 = m1 & (m1 - 1);
  int l2 = __builtin_ctz(m2 | 16);
  int m3_ = m2 & (m2 - 1);
  int l3 = __builtin_ctz(m3_ | 16);
  return k == 0 ? l0 : (k == 1 ? l1 : (k == 2 ? l2 : l3));
}

extern __shared__ __attribute__((aligned(16))) double smem[];

template <int NT>
__global__ void __launch_bounds__(NT)
rg_qp_admm_kernel(const DevCfg *__restrict__ c, DevState st, DevOut out, int nc, int B) {
  const int H = c->H;
  const int m3 = 3 * nc;        // variables per step
  const int n = m3 * H;
  const int LD = n | 1;
  const int tid = threadIdx.x;
  const int count = st.counts[nc];
  // LDS carve
  double *M = smem;                  // n*LD
  double *piv = M + (size_t)n * LD;  // n   (pivot row / rhs vector)
  double *wv = piv + n;              // n   (projection input)
  double *GU = wv + n;               // m3*m3
  double *GV = GU + m3 * m3;         // m3*m3
  double *c1 = GV + m3 * m3;         // H*m3
  double *c2 = c1 + H * m3;          // H*m3
  double *Bw = c2 + H * m3;          // 3*m3   Iw^-1 [r]x  (omega rows)
  double *TBw = Bw + 3 * m3;         // 3*m3   T * Bw
  double *rec = TBw + 3 * m3;        // RG_REC_N
  double *grf = rec + RG_REC_N;      // 12 forces + 12 torques
  const double *Nt = c->Ntab, *St = c->Stab;
  const double rho = c->rho, relax = c->relax, mu = c->mu, lo = c->fz_min, hi = c->fz_max;
  const double kA = 1.0 / (1.0 + 2.0 * mu * mu), kB = 1.0 / (1.0 + mu * mu);
  const double dt = c->dt;

  for (int work = blockIdx.x; work < count; work += gridDim.x) {
    const int b = st.bins[(size_t)nc * B + work];
    __syncthreads();
    for (int e = tid; e < RG_REC_N; e += NT) rec[e] = st.rec[(size_t)b * RG_REC_N + e];
    if (tid < 24) grf[tid] = 0.0;
    __syncthreads();
    const int cmask = (int)rec[REC_CONTACT];
    // ---- Bw = Iw^-1 [r_l]x ; TBw = T Bw,  T = [[1/cp,0,0],[0,1,0],[tan p,0,1]] ----
    if (tid < m3) {
      int l = nth_leg(cmask, tid / 3), d = tid % 3;
      const double *r = &rec[REC_FEETW + 3 * l];
      // column d of skew(r): skew = [[0,-rz,ry],[rz,0,-rx],[-ry,rx,0]]
      double s0 = (d == 0) ? 0.0 : (d == 1 ? -r[2] : r[1]);
      double s1 = (d == 0) ? r[2] : (d == 1 ? 0.0 : -r[0]);
      double s2 = (d == 0) ? -r[1] : (d == 1 ? r[0] : 0.0);
      const double *Iw = &rec[REC_IWINV];
      double b0 = Iw[0] * s0 + Iw[1] * s1 + Iw[2] * s2;
      double b1 = Iw[3] * s0 + Iw[4] * s1 + Iw[5] * s2;
      double b2 = Iw[6] * s0 + Iw[7] * s1 + Iw[8] * s2;
      Bw[0 * m3 + tid] = b0; Bw[1 * m3 + tid] = b1; Bw[2 * m3 + tid] = b2;
      TBw[0 * m3 + tid] = rec[REC_INVCP] * b0;
      TBw[1 * m3 + tid] = b1;
      TBw[2 * m3 + tid] = rec[REC_TANP] * b0 + b2;
    }
    __syncthreads();
    // ---- Gram matrices GU = U'WU, GV = V'WV (U = dt[Bw; E/m], V = dt^2[T Bw; E/m]) ----
    for (int e = tid; e < m3 * m3; e += NT) {
      int i = e / m3, j = e % m3;
      double gu = c->w[6] * Bw[i] * Bw[j] + c->w[7] * Bw[m3 + i] * Bw[m3 + j] + c->w[8] * Bw[2 * m3 + i] * Bw[2 * m3 + j];
      double gv = c->w[0] * TBw[i] * TBw[j] + c->w[1] * TBw[m3 + i] * TBw[m3 + j] + c->w[2] * TBw[2 * m3 + i] * TBw[2 * m3 + j];
      if (i % 3 == j % 3) { gu += c->w[9 + i % 3] * c->inv_mass * c->inv_mass; gv += c->w[3 + i % 3] * c->inv_mass * c->inv_mass; }
      GU[e] = gu * dt * dt;
      GV[e] = gv * dt * dt * dt * dt;
    }
    // ---- linear term pieces: c1_k = U'(W e_k)_{omega,v}, c2_k = V'(W e_k)_{rpy,pos}, k = a+1 ----
    if (tid < n) {
      int a = tid / m3, i = tid % m3;
      double kd = (double)(a + 1) * dt;
      const double *om = &rec[REC_OMEGA], *vb = &rec[REC_VBODY], *cm = &rec[REC_CMD];
      // free response minus reference at step k
      double e_r = rec[REC_ROLL] + kd * rec[REC_INVCP] * om[0];
      double e_p = rec[REC_PITCH] + kd * om[1];
      double e_y = kd * (rec[REC_TANP] * om[0] + om[2]) - kd * cm[2];
      double e_x = kd * vb[0] - kd * cm[0];
      double e_yy = kd * vb[1] - kd * cm[1];
      double e_z = rec[REC_COMZ] + kd * vb[2] - 0.5 * kd * kd * c->g - c->body_height;
      double e_w0 = om[0], e_w1 = om[1], e_w2 = om[2] - cm[2];
      double e_v0 = vb[0] - cm[0], e_v1 = vb[1] - cm[1], e_v2 = vb[2] - kd * c->g;
      int d = i % 3;
      double ev = (d == 0) ? c->w[9] * e_v0 : (d == 1 ? c->w[10] * e_v1 : c->w[11] * e_v2);
      double ep = (d == 0) ? c->w[3] * e_x : (d == 1 ? c->w[4] * e_yy : c->w[5] * e_z);
      c1[tid] = dt * (Bw[i] * c->w[6] * e_w0 + Bw[m3 + i] * c->w[7] * e_w1 + Bw[2 * m3 + i] * c->w[8] * e_w2 + c->inv_mass * ev);
      c2[tid] = dt * dt * (TBw[i] * c->w[0] * e_r + TBw[m3 + i] * c->w[1] * e_p + TBw[2 * m3 + i] * c->w[2] * e_y + c->inv_mass * ep);
    }
    __syncthreads();
    // ---- assemble column i of (P + rho I) and q_i ----
    double qi = 0.0;
    if (tid < n) {
      int a = tid / m3, i = tid % m3;
      for (int kk = a; kk < H; kk++) qi += c1[kk * m3 + i] + ((double)(kk - a) + 0.5) * c2[kk * m3 + i];
      qi *= 2.0;
      for (int bb = 0; bb < H; bb++) {
        double nab = 2.0 * Nt[a * H + bb], sab = 2.0 * St[a * H + bb];
        for (int j = 0; j < m3; j++) {
          int col = bb * m3 + j;
          double v = nab * GU[j * m3 + i] + sab * GV[j * m3 + i];
          if (col == tid) v += c->alpha + rho;
          M[(size_t)col * LD + tid] = v;
        }
      }
    }
    __syncthreads();
    // ---- symmetric sweep: M <- -(P + rho I)^-1 ----
    for (int kp = 0; kp < n; kp++) {
      if (tid < n) piv[tid] = M[(size_t)kp * LD + tid];
      __syncthreads();
      if (tid < n) {
        double d = piv[kp], invd = 1.0 / d;
        if (tid != kp) {
          double cc = piv[tid] * invd;
          for (int j = 0; j < n; j++) {
            double cur = M[(size_t)j * LD + tid];
            M[(size_t)j * LD + tid] = (j == kp) ? cc : cur - cc * piv[j];
          }
        } else {
          for (int j = 0; j < n; j++) M[(size_t)j * LD + tid] = (j == kp) ? -invd : piv[j] * invd;
        }
      }
      __syncthreads();
    }
    // ---- over-relaxed ADMM:  u = Minv (rho (z - y) - q);  z = Proj_K(relax u + (1-relax) z + y) ----
    double z = (tid < n && (tid % 3) == 2) ? lo : 0.0, y = 0.0;
    const double atol = c->admm_abs_tol;
    const int chk = c->admm_check;
    double zchk = z;
    int it = 0, next_chk = chk;
    for (; it < c->admm_iters; it++) {
      if (tid < n) piv[tid] = rho * (z - y) - qi;
      __syncthreads();
      double u = 0.0;
      if (tid < n) {
        for (int j = 0; j < n; j++) u -= M[(size_t)j * LD + tid] * piv[j];
        double uh = relax * u + (1.0 - relax) * z;
        wv[tid] = uh + y;
      }
      __syncthreads();
      if (tid < n) {
        int blk = tid - tid % 3;
        double px, py, pz;
        proj_pyramid(wv[blk], wv[blk + 1], wv[blk + 2], mu, lo, hi, kA, kB, px, py, pz);
        int d = tid % 3;
        double zn = (d == 0) ? px : (d == 1 ? py : pz);
        y = wv[tid] - zn;
        z = zn;
      }
      if (atol > 0.0 && it + 1 == next_chk) {
        const int moving = tid < n && fabs(z - zchk) > atol;
        zchk = z;
        next_chk += chk;
        if (!__syncthreads_or(moving)) { it++; break; }
      }
    }
    if (tid == 0) { atomicAdd(&st.counts[5], it); atomicMax(&st.counts[6], it); }
    // ---- first-step forces (negated), torques, action row ----
    if (tid < m3) grf[3 * nth_leg(cmask, tid / 3) + tid % 3] = -z;
    __syncthreads();
    if (tid < 12) {
      int leg = tid / 3, j = tid % 3;
      const double *J = &rec[REC_JAC + 9 * leg];
      double tau = (grf[3 * leg] * J[j] + grf[3 * leg + 1] * J[3 + j] + grf[3 * leg + 2] * J[6 + j]) * c->mdir[tid];
      grf[12 + tid] = tau;
      if (out.grf) out.grf[(size_t)b * 12 + tid] = (float)grf[tid];
      if (out.tau_stance) out.tau_stance[(size_t)b * 12 + tid] = (float)tau;
    }
    __syncthreads();
    if (tid < 60) {
      int j = tid / 5, f = tid % 5;
      int emit = ((int)rec[REC_EMIT] >> j) & 1;
      float v;
      if (emit) v = (f == 0) ? (float)rec[REC_SWINGQ + j] : (f == 1 ? (float)c->kp[j] : (f == 3 ? (float)c->kd[j] : 0.f));
      else v = (f == 4) ? (float)grf[12 + j] : 0.f;
      out.action[(size_t)b * 60 + tid] = v;
    }
  }
}

// ------------------------------------------------------------------------------------
// QP kernel, register-resident variant (the fast path).
// Thread t = (row r = t / SPLIT, part s = t % SPLIT) keeps C = N / SPLIT consecutive entries of
// row r of the symmetric matrix in VGPRs with compile-time indices.  The MI355X register file
// (512 KB per CU) is 3x its LDS, so this lifts the LDS-capacity occupancy limit of the
// LDS-resident kernel; LDS only carries the pivot row (sweep) / the rhs vector (ADMM) as
// broadcast reads.
//   sweep step kp:  pivot lanes publish row kp (with entry kp replaced by d-1 so that the
//                   unconditional FMA leaves cc = A_ik/d in column kp of every other row),
//                   pivot lanes scale their own row; its diagonal then holds +1 instead of
//                   -1/d, which is never read by another row and is undone in the mat-vec
//                   (see the branch-free form in the kernel body).
// ------------------------------------------------------------------------------------
template <int NC, int H, int SPLIT, int MINW>
__global__ void __launch_bounds__(((3 * NC * H * SPLIT + 63) / 64) * 64, MINW)
rg_qp_admm_reg_kernel(const DevCfg *__restrict__ c, DevState st, DevOut out, int B) {
  constexpr int m3 = 3 * NC;
  constexpr int N = m3 * H;
  constexpr int C = N / SPLIT;
  constexpr int NT = ((N * SPLIT + 63) / 64) * 64;
  static_assert(N % SPLIT == 0 && C % 2 == 0, "row split must give an even number of entries per lane");
  const int tid = threadIdx.x;
  const int r = tid / SPLIT, s = tid % SPLIT;
  const bool active = r < N;
  const int col0 = s * C;
  const int count = st.counts[NC];
  constexpr int NP = (N + 2 + 1) & ~1;  // pivot buffer stride (doubles), even for 16-B alignment
  double *pbuf = smem;               // 2 * NP   ping-pong pivot row; [N] = pivot value d
  double *vv = pbuf + 2 * NP;        // N  rhs vector
  double *wv = vv + N;               // N  projection input
  double *GU = wv + N;               // m3*m3
  double *GV = GU + m3 * m3;
  double *c1 = GV + m3 * m3;         // N
  double *c2 = c1 + N;               // N
  double *Bw = c2 + N;               // 3*m3
  double *TBw = Bw + 3 * m3;         // 3*m3
  double *rec = TBw + 3 * m3;        // RG_REC_N
  double *grf = rec + RG_REC_N;      // 24
  double *tabN = grf + 24;           // H*H
  double *tabS = tabN + H * H;       // H*H
  const double rho = c->rho, relax = c->relax, mu = c->mu, lo = c->fz_min, hi = c->fz_max, dt = c->dt;
  const double kA = 1.0 / (1.0 + 2.0 * mu * mu), kB = 1.0 / (1.0 + mu * mu);
  for (int e = tid; e < H * H; e += NT) { tabN[e] = 2.0 * c->Ntab[e]; tabS[e] = 2.0 * c->Stab[e]; }

  // Static round-robin over the bin.  (A dynamic atomic work queue was measured 15 % slower here:
  // under load the CU is throughput-bound, so keeping every slot busy in the tail only adds contention.)
  for (int work = blockIdx.x; work < count; work += gridDim.x) {
    const int b = st.bins[(size_t)NC * B + work];
    __syncthreads();
    for (int e = tid; e < RG_REC_N; e += NT) rec[e] = st.rec[(size_t)b * RG_REC_N + e];
    if (tid < 24) grf[tid] = 0.0;
    __syncthreads();
    const int cmask = (int)rec[REC_CONTACT];
    if (tid < m3) {
      int l = nth_leg(cmask, tid / 3), d = tid % 3;
      const double *rr = &rec[REC_FEETW + 3 * l];
      double s0 = (d == 0) ? 0.0 : (d == 1 ? -rr[2] : rr[1]);
      double s1 = (d == 0) ? rr[2] : (d == 1 ? 0.0 : -rr[0]);
      double s2 = (d == 0) ? -rr[1] : (d == 1 ? rr[0] : 0.0);
      const double *Iw = &rec[REC_IWINV];
      double b0 = Iw[0] * s0 + Iw[1] * s1 + Iw[2] * s2;
      double b1 = Iw[3] * s0 + Iw[4] * s1 + Iw[5] * s2;
      double b2 = Iw[6] * s0 + Iw[7] * s1 + Iw[8] * s2;
      Bw[tid] = b0; Bw[m3 + tid] = b1; Bw[2 * m3 + tid] = b2;
      TBw[tid] = rec[REC_INVCP] * b0; TBw[m3 + tid] = b1; TBw[2 * m3 + tid] = rec[REC_TANP] * b0 + b2;
    }
    __syncthreads();
    for (int e = tid; e < m3 * m3; e += NT) {
      int i = e / m3, j = e % m3;
      double gu = c->w[6] * Bw[i] * Bw[j] + c->w[7] * Bw[m3 + i] * Bw[m3 + j] + c->w[8] * Bw[2 * m3 + i] * Bw[2 * m3 + j];
      double gv = c->w[0] * TBw[i] * TBw[j] + c->w[1] * TBw[m3 + i] * TBw[m3 + j] + c->w[2] * TBw[2 * m3 + i] * TBw[2 * m3 + j];
      if (i % 3 == j % 3) { gu += c->w[9 + i % 3] * c->inv_mass * c->inv_mass; gv += c->w[3 + i % 3] * c->inv_mass * c->inv_mass; }
      GU[e] = gu * dt * dt;
      GV[e] = gv * dt * dt * dt * dt;
    }
    if (tid < N) {
      int a = tid / m3, i = tid % m3;
      double kd = (double)(a + 1) * dt;
      const double *om = &rec[REC_OMEGA], *vb = &rec[REC_VBODY], *cm = &rec[REC_CMD];
      double e_r = rec[REC_ROLL] + kd * rec[REC_INVCP] * om[0];
      double e_p = rec[REC_PITCH] + kd * om[1];
      double e_y = kd * (rec[REC_TANP] * om[0] + om[2]) - kd * cm[2];
      double e_x = kd * vb[0] - kd * cm[0];
      double e_yy = kd * vb[1] - kd * cm[1];
      double e_z = rec[REC_COMZ] + kd * vb[2] - 0.5 * kd * kd * c->g - c->body_height;
      double e_w0 = om[0], e_w1 = om[1], e_w2 = om[2] - cm[2];
      double e_v0 = vb[0] - cm[0], e_v1 = vb[1] - cm[1], e_v2 = vb[2] - kd * c->g;
      int d = i % 3;
      double ev = (d == 0) ? c->w[9] * e_v0 : (d == 1 ? c->w[10] * e_v1 : c->w[11] * e_v2);
      double ep = (d == 0) ? c->w[3] * e_x : (d == 1 ? c->w[4] * e_yy : c->w[5] * e_z);
      c1[tid] = dt * (Bw[i] * c->w[6] * e_w0 + Bw[m3 + i] * c->w[7] * e_w1 + Bw[2 * m3 + i] * c->w[8] * e_w2 + c->inv_mass * ev);
      c2[tid] = dt * dt * (TBw[i] * c->w[0] * e_r + TBw[m3 + i] * c->w[1] * e_p + TBw[2 * m3 + i] * c->w[2] * e_y + c->inv_mass * ep);
    }
    __syncthreads();
    // ---- my C entries of row r of (P + rho I), and q_r ----
    double row[C];
    double qi = 0.0;
    {
      // Opaque copies: without them LICM hoists ~4*C LDS addresses out of the persistent robot
      // loop and keeps them live through the sweep/ADMM (measured: 200 VGPRs, or spills to HBM).
      int rv = active ? r : 0, col0v = col0;
      asm volatile("" : "+v"(rv), "+v"(col0v));
      const int a = rv / m3, i = rv % m3;
      for (int kq = a; kq < H; kq++) qi += c1[kq * m3 + i] + ((double)(kq - a) + 0.5) * c2[kq * m3 + i];
      qi *= 2.0;
      const double *tN = tabN + a * H, *tS = tabS + a * H, *gu = GU + i, *gv = GV + i;
#pragma unroll
      for (int jj = 0; jj < C; jj++) {
        const int col = col0v + jj;
        const int bb = col / m3, j = col - bb * m3;
        double v = tN[bb] * gu[j * m3] + tS[bb] * gv[j * m3];
        if (col == rv) v += c->alpha + rho;
        row[jj] = v;
      }
    }
    // ---- symmetric sweep, rows in registers, branch-free ----
    // Every lane applies row += ncc * pivot_row' with pivot_row'[kp] = d - 1:
    //   other rows: ncc = -A_rk/d      -> column kp becomes A_rk/d, the rest A_rj - A_rk A_kj/d
    //   pivot row : ncc = 1/d - 1      -> row/d, and its diagonal becomes 2 - 1/d instead of -1/d.
    // The pivot row's diagonal is never read by another row, so the constant +2 is undone in
    // the mat-vec (part -= 2 rhs on the lane part that owns the diagonal).
    const int my_diag_part = r / C;  // which part of row r holds the diagonal
    if constexpr (C <= 32) {
    // The pivot loop is unrolled over the C positions inside a lane part (static register index of the
      // pivot element), so the pivot lane publishes the patched row and d without reading anything back.
      for (int sp = 0; sp < SPLIT; sp++) {
#pragma unroll
        for (int pj = 0; pj < C; pj++) {
          const int kp = sp * C + pj;
          double *pb = pbuf + (pj & 1) * NP;
          if (active && r == kp) {
#pragma unroll
            for (int jj = 0; jj < C; jj += 2) {
              double v0 = row[jj], v1 = row[jj + 1];
              if (jj == pj) v0 = (s == sp) ? v0 - 1.0 : v0;
              if (jj + 1 == pj) v1 = (s == sp) ? v1 - 1.0 : v1;
              *reinterpret_cast<double2 *>(&pb[col0 + jj]) = make_double2(v0, v1);
            }
            if (s == sp) pb[N] = row[pj];
          }
          __syncthreads();
          if (active) {
            const double invd = fast_rcp(pb[N]);
            const double ncc = (r == kp) ? invd - 1.0 : -pb[r] * invd;
#pragma unroll
            for (int jj = 0; jj < C; jj += 2) {
              double2 p2 = *reinterpret_cast<const double2 *>(&pb[col0 + jj]);
              row[jj] = fma(ncc, p2.x, row[jj]);
              row[jj + 1] = fma(ncc, p2.y, row[jj + 1]);
            }
          }
          pin_array<C>(row);  // stop hipcc from turning the unrolled pivots into a register-hungry look-ahead schedule
        }
      }
    } else {
      // wide rows: a fully unrolled pivot loop would not fit the register budget; the pivot lane reads
      // its own diagonal entry back from LDS instead (dynamic position inside the row).
      for (int kp = 0; kp < N; kp++) {
        double *pb = pbuf + (kp & 1) * NP;
        if (active && r == kp) {
#pragma unroll
          for (int jj = 0; jj < C; jj += 2) *reinterpret_cast<double2 *>(&pb[col0 + jj]) = make_double2(row[jj], row[jj + 1]);
          if (s == kp / C) { double d = pb[kp]; pb[kp] = d - 1.0; pb[N] = d; }
        }
        __syncthreads();
        if (active) {
          const double invd = fast_rcp(pb[N]);
          const double ncc = (r == kp) ? invd - 1.0 : -pb[r] * invd;
#pragma unroll
          for (int jj = 0; jj < C; jj += 2) {
            double2 p2 = *reinterpret_cast<const double2 *>(&pb[col0 + jj]);
            row[jj] = fma(ncc, p2.x, row[jj]);
            row[jj + 1] = fma(ncc, p2.y, row[jj + 1]);
          }
        }
      }
    }
    // row now holds -(P + rho I)^-1 entries (diagonal offset by +2)
    // ---- over-relaxed ADMM ----
    double z = (active && (r % 3) == 2) ? lo : 0.0, y = 0.0;
    const int blk = active ? r - r % 3 : 0, dax = r % 3;
    const double atol = c->admm_abs_tol;
    const int chk = c->admm_check;
    double zchk = z;
    int it = 0, next_chk = chk;
    for (; it < c->admm_iters; it++) {
      const double rhs = rho * (z - y) - qi;
      if (active && s == 0) vv[r] = rhs;
      __syncthreads();
      double part = 0.0;
      if (active) {
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
        for (int jj = 0; jj + 3 < C; jj += 4) {
          double2 va = *reinterpret_cast<const double2 *>(&vv[col0 + jj]);
          double2 vb2 = *reinterpret_cast<const double2 *>(&vv[col0 + jj + 2]);
          p0 = fma(row[jj], va.x, p0);
          p1 = fma(row[jj + 1], va.y, p1);
          p2 = fma(row[jj + 2], vb2.x, p2);
          p3 = fma(row[jj + 3], vb2.y, p3);
        }
        if constexpr (C % 4 != 0) {
          double2 va = *reinterpret_cast<const double2 *>(&vv[col0 + C - 2]);
          p0 = fma(row[C - 2], va.x, p0);
          p1 = fma(row[C - 1], va.y, p1);
        }
        part = (p0 + p1) + (p2 + p3);
        if (s == my_diag_part) part -= 2.0 * rhs;
      }
#pragma unroll
      for (int o = 1; o < SPLIT; o <<= 1) part += __shfl_xor(part, o);
      const double u = -part;
      const double w = relax * u + (1.0 - relax) * z + y;
      if (active && s == 0) wv[r] = w;
      __syncthreads();
      if (active) {
        double px, py, pz;
        proj_pyramid(wv[blk], wv[blk + 1], wv[blk + 2], mu, lo, hi, kA, kB, px, py, pz);
        const double zn = (dax == 0) ? px : (dax == 1 ? py : pz);
        y = w - zn;
        z = zn;
      }
      if (atol > 0.0 && it + 1 == next_chk) {
        const int moving = active && fabs(z - zchk) > atol;
        zchk = z;
        next_chk += chk;
        if (!__syncthreads_or(moving)) { it++; break; }
      }
    }
    if (tid == 0) { atomicAdd(&st.counts[5], it); atomicMax(&st.counts[6], it); }
    if (active && s == 0 && r < m3) grf[3 * nth_leg(cmask, r / 3) + r % 3] = -z;
    __syncthreads();
    if (tid < 12) {
      int leg = tid / 3, j = tid % 3;
      const double *J = &rec[REC_JAC + 9 * leg];
      double tau = (grf[3 * leg] * J[j] + grf[3 * leg + 1] * J[3 + j] + grf[3 * leg + 2] * J[6 + j]) * c->mdir[tid];
      grf[12 + tid] = tau;
      if (out.grf) out.grf[(size_t)b * 12 + tid] = (float)grf[tid];
      if (out.tau_stance) out.tau_stance[(size_t)b * 12 + tid] = (float)tau;
    }
    __syncthreads();
    if (tid < 60) {
      int j = tid / 5, f = tid % 5;
      int emit = ((int)rec[REC_EMIT] >> j) & 1;
      float v;
      if (emit) v = (f == 0) ? (float)rec[REC_SWINGQ + j] : (f == 1 ? (float)c->kp[j] : (f == 3 ? (float)c->kd[j] : 0.f));
      else v = (f == 4) ? (float)grf[12 + j] : 0.f;
      out.action[(size_t)b * 60 + tid] = v;
    }
  }
}

template <int NC, int H, int SPLIT>
static size_t qp_reg_lds_bytes() {
  constexpr int m3 = 3 * NC, N = m3 * H, NP = (N + 2 + 1) & ~1;
  return sizeof(double) * (size_t)(2 * NP + 2 * N + 2 * m3 * m3 + 2 * N + 6 * m3 + RG_REC_N + 24 + 2 * H * H);
}

template <int NC, int H, int SPLIT, int MINW>
static hipError_t launch_qp_reg(const DevCfg *dcfg, const DevState &st, const DevOut &dout, int B, int cu_count, hipStream_t s) {
  constexpr int NT = ((3 * NC * H * SPLIT + 63) / 64) * 64;
  int grid = cu_count * 8;
  if (grid > B) grid = B;
  const size_t lds = qp_reg_lds_bytes<NC, H, SPLIT>();
  rg_qp_admm_reg_kernel<NC, H, SPLIT, MINW><<<dim3(grid), dim3(NT), lds, s>>>(dcfg, st, dout, B);
  return hipGetLastError();
}

// returns true if a register-resident instantiation exists for (nc, H)
static bool launch_qp_reg_dispatch(int variant, int nc, int H, const DevCfg *dcfg, const DevState &st, const DevOut &dout, int B, int cu, hipStream_t s, hipError_t *err) {
  *err = hipSuccess;
  if (H == 10 && variant == 1) {  // A/B: alternative row splits of the row-per-lane kernel
    switch (nc) {
      case 2: *err = launch_qp_reg<2, 10, 1, 2>(dcfg, st, dout, B, cu, s); return true;
      case 4: *err = launch_qp_reg<4, 10, 4, 3>(dcfg, st, dout, B, cu, s); return true;
    }
  }
  if (H == 10) {
    switch (nc) {
      case 1: *err = launch_qp_reg<1, 10, 1, 4>(dcfg, st, dout, B, cu, s); return true;
      case 2: *err = launch_qp_reg<2, 10, 2, 3>(dcfg, st, dout, B, cu, s); return true;
      case 3: *err = launch_qp_reg<3, 10, 1, 1>(dcfg, st, dout, B, cu, s); return true;
      case 4: *err = launch_qp_reg<4, 10, 2, 2>(dcfg, st, dout, B, cu, s); return true;
    }
  } else if (H == 20) {
    switch (nc) {
      case 1: *err = launch_qp_reg<1, 20, 1, 2>(dcfg, st, dout, B, cu, s); return true;
      case 2: *err = launch_qp_reg<2, 20, 2, 2>(dcfg, st, dout, B, cu, s); return true;
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------
// QP kernel, 2-D register-tiled variant.
// The row-per-lane kernel above is bound by the LDS instruction pipe (rocprof:
// SQ_ACTIVE_INST_LDS ~ 88 % of kernel time): every f64 FMA needs half a 16-B broadcast read.
// Here the lanes of a robot form an LC x LC grid and lane (lr, lc) keeps the T x T tile
// rows lr*T.., cols lc*T.. of the (padded, NP = T*LC) symmetric matrix in VGPRs, so every value
// read from LDS feeds T FMAs:
//   sweep step kp: 2T values (pivot-row entries of my columns and, by symmetry, of my rows)
//                  for T*T FMAs; the LC lanes of lane-row kp/T publish the row in parallel.
//   ADMM mat-vec : T values of the rhs for T*T FMAs, then a reduce-scatter over the LC lanes
//                  of a lane-row (cross-lane, no LDS data) leaves one finished entry per lane.
// ------------------------------------------------------------------------------------
// Opaque "use + redefine" of one tile row: no instruction is emitted, but the optimiser can no longer
// defer this row's updates past this point.  (Left alone, hipcc turns the unrolled pivot steps into
// a look-ahead schedule that keeps every step's pivot-row values live: > 380 VGPRs, spills in the loop.)
template <int T>
__device__ __forceinline__ void pin_row(double (&t)[T]) {
  if constexpr (T == 8) asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
  else if constexpr (T == 6) asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]));
  else if constexpr (T == 4) asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
  else {
#pragma unroll
    for (int i = 0; i < T; i++) asm volatile("" : "+v"(t[i]));
  }
}

template <int LT>
__device__ __forceinline__ int bitrev_lt(int x) {
  int r = 0;
#pragma unroll
  for (int i = 0; i < LT; i++) r |= ((x >> i) & 1) << (LT - 1 - i);
  return r;
}

template <int NC, int H, int T, int LG, int MINW, bool AS>
__global__ void __launch_bounds__((1 << LG) * (1 << LG), AS ? 1 : MINW)
rg_qp_admm_tile_kernel(const DevCfg *__restrict__ c, DevState st, DevOut out, int B, int retry_pass) {
  constexpr int m3 = 3 * NC;
  constexpr int N = m3 * H;          // real QP variables
  constexpr int LC = 1 << LG;        // lanes per side
  constexpr int NP = T * LC;         // padded size
  constexpr int NT = LC * LC;
  constexpr bool TPOW2 = (T & (T - 1)) == 0;
  constexpr int LT = (T == 2) ? 1 : (T == 4) ? 2 : (T == 8) ? 3 : (T == 16) ? 4 : 0;
  static_assert(NP >= N && T % 2 == 0 && T <= LC, "tile/grid must cover the problem");
  const int tid = threadIdx.x;
  const int lr = tid >> LG, lc = tid & (LC - 1);
  // retry_pass: work list = robots the ADMM pass could not converge (RG_SOLVER_AUTO)
  const int list = retry_pass ? 5 + NC : NC;
  const int count = st.counts[retry_pass ? 8 + NC : NC];
  // LDS vectors read as T-wide groups (one group per lane-column) use a padded group stride TS so the
  // 16-B reads of different groups never share a bank (T = 8: 64-B groups are 2-way conflicting,
  // rocprof SQ_LDS_BANK_CONFLICT = 30-55 % of LDS cycles; stride 80 B is conflict-free for 8 and 16 groups).
  constexpr int TS = (T == 8) ? 10 : T;
  constexpr int NPAD = TS * LC;
  constexpr int NPB = NPAD + 2;
  double *pbuf = smem;               // 2 * NPB  ping-pong pivot row (padded); [NPAD] = pivot d
  double *vv = pbuf + 2 * NPB;       // NPAD rhs vector (padded)
  double *wv = vv + NPAD;            // NP projection input (unpadded, scalar accesses)
  double *GU = wv + NP;              // m3*m3
  double *GV = GU + m3 * m3;
  double *c1 = GV + m3 * m3;         // N
  double *c2 = c1 + N;               // N
  double *Bw = c2 + N;               // 3*m3
  double *TBw = Bw + 3 * m3;         // 3*m3
  double *rec = TBw + 3 * m3;        // RG_REC_N
  double *grf = rec + RG_REC_N;      // 24
  double *tabN = grf + 24;           // H*H
  double *tabS = tabN + H * H;       // H*H
  // active-set solver storage (only carved when the launch reserved it)
  constexpr int QMAX = N;            // at most N linearly independent active constraints
  double *as_x = tabS + H * H;       // NP   primal iterate
  double *as_g = as_x + NP;          // NP   g = G c_p
  double *as_z = as_g + NP;          // NP   step direction
  double *as_pg = as_z + NP;         // 2*NPAD published rows of G (padded groups)
  double *as_lam = as_pg + 2 * NPAD; // QMAX multipliers
  double *as_sv = as_lam + QMAX;     // QMAX
  double *as_r = as_sv + QMAX;       // QMAX
  double *as_v0 = as_r + QMAX;       // QMAX constraint coefficients
  double *as_v1 = as_v0 + QMAX;      // QMAX
  double *as_T = as_v1 + QMAX;       // QMAX*(QMAX+1)/2  packed symmetric (C_A G C_A')^-1
  int *as_i0 = reinterpret_cast<int *>(as_T + QMAX * (QMAX + 1) / 2);  // QMAX
  int *as_i1 = as_i0 + QMAX;         // QMAX
  int *as_id = as_i1 + QMAX;         // QMAX constraint ids
  int *as_am = as_id + QMAX;         // N/3 active-type mask per block (6 bits)
  double *as_sc = reinterpret_cast<double *>(as_am + ((N / 3 + 1) & ~1));  // 8 scalars
  constexpr bool use_as = AS;   // separate instantiation: the active-set path must not cost the ADMM path registers
  const double rho = use_as ? 0.0 : c->rho, relax = c->relax, mu = c->mu, lo = c->fz_min, hi = c->fz_max, dt = c->dt;
  const double kA = 1.0 / (1.0 + 2.0 * mu * mu), kB = 1.0 / (1.0 + mu * mu);
  for (int e = tid; e < H * H; e += NT) { tabN[e] = 2.0 * c->Ntab[e]; tabS[e] = 2.0 * c->Stab[e]; }
  // the one matrix row whose scalar ADMM state this lane owns after the reduce-scatter
  const bool owner = lc < T;
  // T == 8: the reduce-scatter below pairs lanes as (7-i), (i^1), (i^2) -> kept row 4*b2 + 2*b0 + b1
  const int own_a = (T == 8) ? (4 * ((lc >> 2) & 1) + 2 * (lc & 1) + ((lc >> 1) & 1)) : (TPOW2 ? bitrev_lt<LT>(lc & (T - 1)) : (lc < T ? lc : 0));
  const int io = lr * T + own_a;
  const bool own_real = owner && io < N;
  const int iov_pad = lr * TS + own_a;   // position of row io in the padded vectors

  // Static round-robin over the bin.  (A dynamic atomic work queue was measured 15 % slower here:
  // under load the CU is throughput-bound, so keeping every slot busy in the tail only adds contention.)
  for (int work = blockIdx.x; work < count; work += gridDim.x) {
    const int b = st.bins[(size_t)list * B + work];
    __syncthreads();
    for (int e = tid; e < RG_REC_N; e += NT) rec[e] = st.rec[(size_t)b * RG_REC_N + e];
    if (tid < 24) grf[tid] = 0.0;
    __syncthreads();
    const int cmask = (int)rec[REC_CONTACT];
    if (tid < m3) {
      int l = nth_leg(cmask, tid / 3), d = tid % 3;
      const double *rr = &rec[REC_FEETW + 3 * l];
      double s0 = (d == 0) ? 0.0 : (d == 1 ? -rr[2] : rr[1]);
      double s1 = (d == 0) ? rr[2] : (d == 1 ? 0.0 : -rr[0]);
      double s2 = (d == 0) ? -rr[1] : (d == 1 ? rr[0] : 0.0);
      const double *Iw = &rec[REC_IWINV];
      double b0 = Iw[0] * s0 + Iw[1] * s1 + Iw[2] * s2;
      double b1 = Iw[3] * s0 + Iw[4] * s1 + Iw[5] * s2;
      double b2 = Iw[6] * s0 + Iw[7] * s1 + Iw[8] * s2;
      Bw[tid] = b0; Bw[m3 + tid] = b1; Bw[2 * m3 + tid] = b2;
      TBw[tid] = rec[REC_INVCP] * b0; TBw[m3 + tid] = b1; TBw[2 * m3 + tid] = rec[REC_TANP] * b0 + b2;
    }
    __syncthreads();
    for (int e = tid; e < m3 * m3; e += NT) {
      int i = e / m3, j = e % m3;
      double gu = c->w[6] * Bw[i] * Bw[j] + c->w[7] * Bw[m3 + i] * Bw[m3 + j] + c->w[8] * Bw[2 * m3 + i] * Bw[2 * m3 + j];
      double gv = c->w[0] * TBw[i] * TBw[j] + c->w[1] * TBw[m3 + i] * TBw[m3 + j] + c->w[2] * TBw[2 * m3 + i] * TBw[2 * m3 + j];
      if (i % 3 == j % 3) { gu += c->w[9 + i % 3] * c->inv_mass * c->inv_mass; gv += c->w[3 + i % 3] * c->inv_mass * c->inv_mass; }
      GU[e] = gu * dt * dt;
      GV[e] = gv * dt * dt * dt * dt;
    }
    for (int e = tid; e < N; e += NT) {
      int a = e / m3, i = e % m3;
      double kd = (double)(a + 1) * dt;
      const double *om = &rec[REC_OMEGA], *vb = &rec[REC_VBODY], *cm = &rec[REC_CMD];
      double e_r = rec[REC_ROLL] + kd * rec[REC_INVCP] * om[0];
      double e_p = rec[REC_PITCH] + kd * om[1];
      double e_y = kd * (rec[REC_TANP] * om[0] + om[2]) - kd * cm[2];
      double e_x = kd * vb[0] - kd * cm[0];
      double e_yy = kd * vb[1] - kd * cm[1];
      double e_z = rec[REC_COMZ] + kd * vb[2] - 0.5 * kd * kd * c->g - c->body_height;
      double e_w0 = om[0], e_w1 = om[1], e_w2 = om[2] - cm[2];
      double e_v0 = vb[0] - cm[0], e_v1 = vb[1] - cm[1], e_v2 = vb[2] - kd * c->g;
      int d = i % 3;
      double ev = (d == 0) ? c->w[9] * e_v0 : (d == 1 ? c->w[10] * e_v1 : c->w[11] * e_v2);
      double ep = (d == 0) ? c->w[3] * e_x : (d == 1 ? c->w[4] * e_yy : c->w[5] * e_z);
      c1[e] = dt * (Bw[i] * c->w[6] * e_w0 + Bw[m3 + i] * c->w[7] * e_w1 + Bw[2 * m3 + i] * c->w[8] * e_w2 + c->inv_mass * ev);
      c2[e] = dt * dt * (TBw[i] * c->w[0] * e_r + TBw[m3 + i] * c->w[1] * e_p + TBw[2 * m3 + i] * c->w[2] * e_y + c->inv_mass * ep);
    }
    __syncthreads();
    // ---- my T x T tile of (P + rho I) (identity in the padding), and q of the row I own ----
    double tile[T][T];
    double qi = 0.0;
    {
      int lrv = lr, lcv = lc, iov = io;   // opaque copies defeat LICM of ~4 T^2 LDS addresses
      asm volatile("" : "+v"(lrv), "+v"(lcv), "+v"(iov));
      if (own_real) {
        const int a = iov / m3, i = iov - a * m3;
        for (int kq = a; kq < H; kq++) qi += c1[kq * m3 + i] + ((double)(kq - a) + 0.5) * c2[kq * m3 + i];
        qi *= 2.0;
      }
#pragma unroll
      for (int ta = 0; ta < T; ta++) {
        const int row = lrv * T + ta;
        const int a = row / m3, i = row - a * m3;
        const bool rreal = row < N;
        const double *tN = tabN + (rreal ? a : 0) * H, *tS = tabS + (rreal ? a : 0) * H, *gu = GU + (rreal ? i : 0), *gv = GV + (rreal ? i : 0);
#pragma unroll
        for (int tb = 0; tb < T; tb++) {
          const int col = lcv * T + tb;
          const int bb = col / m3, j = col - bb * m3;
          double v;
          if (rreal && col < N) {
            v = tN[bb] * gu[j * m3] + tS[bb] * gv[j * m3];
            if (col == row) v += c->alpha + rho;
          } else v = (col == row) ? 1.0 : 0.0;
          tile[ta][tb] = v;
        }
        __builtin_amdgcn_sched_barrier(0);  // one tile row at a time: bounds the LDS loads in flight (VGPR pressure)
      }
    }
    // ---- symmetric sweep: tile <- entries of -(P + rho I)^-1, pivot-row diagonals offset by +2 ----
    // Look-ahead: inside step kp the tile row that holds pivot row kp+1 is updated FIRST and published
    // immediately, so its LDS write -> read latency hides behind the other T-1 row updates.
    auto publish = [&](int tr, int kb, double *pb) {   // lanes of lane-row kb publish tile row tr (tr static after unrolling)
      if (lr == kb) {
        const bool diag = (lc == kb);
#pragma unroll
        for (int tb = 0; tb < T; tb += 2) {
          double v0 = tile[tr][tb], v1 = tile[tr][tb + 1];
          if (tb == tr) v0 = diag ? v0 - 1.0 : v0;
          if (tb + 1 == tr) v1 = diag ? v1 - 1.0 : v1;
          *reinterpret_cast<double2 *>(&pb[lc * TS + tb]) = make_double2(v0, v1);
        }
        if (diag) pb[NPAD] = tile[tr][tr];
      }
    };
    publish(0, 0, pbuf);
    __syncthreads();
    for (int kb = 0; kb < LC; kb++) {
#pragma unroll
      for (int tr = 0; tr < T; tr++) {
        const int kp = kb * T + tr;
        double *pb = pbuf + (kp & 1) * NPB;
        const double invd = fast_rcp(pb[NPAD]);
        double prow[T], pcol[T];
#pragma unroll
        for (int t2 = 0; t2 < T; t2 += 2) {
          double2 a2 = *reinterpret_cast<const double2 *>(&pb[lr * TS + t2]);
          double2 b2 = *reinterpret_cast<const double2 *>(&pb[lc * TS + t2]);
          prow[t2] = a2.x; prow[t2 + 1] = a2.y; pcol[t2] = b2.x; pcol[t2 + 1] = b2.y;
        }
        constexpr int dummy = 0; (void)dummy;
        const int tn = (tr + 1) % T;               // tile row of the next pivot (static)
        const int kbn = (tr + 1 < T) ? kb : kb + 1; // its lane-row
        {
          double ncc = -prow[tn] * invd;
          if (tn == tr) ncc = (lr == kb) ? invd - 1.0 : ncc;   // only when T == 1
#pragma unroll
          for (int tb = 0; tb < T; tb++) tile[tn][tb] = fma(ncc, pcol[tb], tile[tn][tb]);
        }
        if (kp + 1 < NP) publish(tn, kbn, pbuf + ((kp + 1) & 1) * NPB);
#pragma unroll
        for (int ta = 0; ta < T; ta++) {
          if (ta == tn) continue;
          double ncc = -prow[ta] * invd;
          if (ta == tr) ncc = (lr == kb) ? invd - 1.0 : ncc;
#pragma unroll
          for (int tb = 0; tb < T; tb++) tile[ta][tb] = fma(ncc, pcol[tb], tile[ta][tb]);
        }
#pragma unroll
        for (int ta = 0; ta < T; ta++) pin_row<T>(tile[ta]);
        __syncthreads();
      }
    }
    double z = 0.0;
    int it = 0;
    if constexpr (use_as) {
      // ================= exact dual active-set (range-space form) =================
      // tile holds -G + 2 I on pivot diagonals, G = P^-1.  x = x0 - G C_A' lam with
      // (C_A G C_A') lam = ..., kept through T = (C_A G C_A')^-1 (packed symmetric, LDS).
      // Constraint id = 6*block + type:  0: -fx+mu fz>=0  1: fx+mu fz>=0  2: -fy+mu fz>=0
      //                                  3:  fy+mu fz>=0  4: fz-lo>=0     5: hi-fz>=0
      constexpr int NB = N / 3;
      auto tile_matvec = [&](const double *vin_pad) -> double {   // returns (G v)_io on owner lanes
        double acc[T];
        double vloc[T];
#pragma unroll
        for (int t2 = 0; t2 < T; t2 += 2) {
          double2 v2 = *reinterpret_cast<const double2 *>(&vin_pad[lc * TS + t2]);
          vloc[t2] = v2.x; vloc[t2 + 1] = v2.y;
        }
#pragma unroll
        for (int ta = 0; ta < T; ta++) {
          double a0 = 0.0;
#pragma unroll
          for (int tb = 0; tb < T; tb++) a0 = fma(tile[ta][tb], vloc[tb], a0);
          acc[ta] = (lr == lc) ? a0 - 2.0 * vloc[ta] : a0;
        }
        double tot;
        if constexpr (T == 8) {
          {
            const bool up = (lc >> 2) & 1;
#pragma unroll
            for (int h2 = 0; h2 < 4; h2++) { double keep = up ? acc[4 + h2] : acc[h2]; double send = up ? acc[h2] : acc[4 + h2]; acc[h2] = keep + dpp_f64<0x141>(send); }
          }
          {
            const bool up = lc & 1;
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) { double keep = up ? acc[2 + h2] : acc[h2]; double send = up ? acc[h2] : acc[2 + h2]; acc[h2] = keep + dpp_f64<0xB1>(send); }
          }
          { const bool up = (lc >> 1) & 1; double keep = up ? acc[1] : acc[0]; double send = up ? acc[0] : acc[1]; tot = keep + dpp_f64<0x4E>(send); }
          if constexpr (LG >= 4) tot += dpp_f64<0x128>(tot);
#pragma unroll
          for (int kx = 4; kx < LG; kx++) tot += __shfl_xor(tot, 1 << kx);
        } else {
#pragma unroll
          for (int ta = 0; ta < T; ta++) {
#pragma unroll
            for (int kx = 0; kx < LG; kx++) acc[ta] += __shfl_xor(acc[ta], 1 << kx);
          }
          tot = acc[0];
#pragma unroll
          for (int ta = 1; ta < T; ta++) tot = (own_a == ta) ? acc[ta] : tot;
        }
        return -tot;
      };
      // --- state machine with ONE mat-vec site: pass 0 computes x0 = -G q, later passes one step each ---
      double *wpad = vv;   // padded mat-vec input (reuses the ADMM rhs buffer)
      if (tid < NB) as_am[tid] = 0;
      double x = 0.0, s_p = 0.0, lam_p = 0.0, sigma = 0.0;
      int q = 0, pid = -1, pi0 = 0, pi1 = 0, pblk = 0, pty = 0;
      double pv0 = 0.0, pv1 = 0.0;
      const double vtol = 1e-9 * (1.0 + hi * 1e-3);
      const int it_cap = 8 * N + 80;
      bool init = true, failed = false;
      int passes = 0;
      for (;; passes++) {
        if (passes > it_cap) { failed = true; break; }
        if (init) {
          if (owner) wpad[iov_pad] = own_real ? qi : 0.0;
        } else {
          // S4: sv = C_A g ;  S5: r = T sv (packed symmetric T, row k on lane k)
          if (tid < q) as_sv[tid] = as_v0[tid] * as_g[as_i0[tid]] + as_v1[tid] * as_g[as_i1[tid]];
          __syncthreads();
          if (tid < q) {
            double rk = 0.0;
            const int base = tid * (tid + 1) / 2;
            for (int j = 0; j <= tid; j++) rk = fma(as_T[base + j], as_sv[j], rk);
            for (int j = tid + 1; j < q; j++) rk = fma(as_T[j * (j + 1) / 2 + tid], as_sv[j], rk);
            as_r[tid] = rk;
          }
          // S6: w = c_p - C_A' r  (dense, padded)
          if (owner) wpad[iov_pad] = (io == pi0 ? pv0 : 0.0) + (io == pi1 ? pv1 : 0.0);
          __syncthreads();
          if (tid < q) {
            const double rk = as_r[tid];
            const int a0 = as_i0[tid], a1 = as_i1[tid];
            atomicAdd(&wpad[(a0 / T) * TS + a0 % T], -rk * as_v0[tid]);
            if (as_v1[tid] != 0.0) atomicAdd(&wpad[(a1 / T) * TS + a1 % T], -rk * as_v1[tid]);
          }
        }
        __syncthreads();
        double zz = tile_matvec(wpad);   // (G w)_io on owner lanes
        if (!own_real) zz = 0.0;
        bool need_search = false;
        if (init) {
          x = -zz;
          if (owner) as_x[io] = x;
          init = false;
          need_search = true;
          __syncthreads();   // x0 must be visible to the search on wave 0
        } else {
          if (owner) as_z[io] = zz;
          // S8: step lengths (dual bound t1 on wave 0)
          if (tid < 64) {
            double t1 = INFINITY; int lsel = -1;
            for (int kq = tid; kq < q; kq += 64) {
              const double rk = as_r[kq];
              if (rk > 0.0) { const double cand = as_lam[kq] / rk; if (cand < t1) { t1 = cand; lsel = kq; } }
            }
            {
              const double tmin = wave_min_f64(t1);
              const unsigned long long hit = __ballot(t1 == tmin && lsel >= 0);
              lsel = hit ? __builtin_amdgcn_readlane(lsel, __ffsll((long long)hit) - 1) : -1;
              t1 = tmin;
            }
            if (tid == 0) { as_sc[2] = t1; as_sc[3] = (double)lsel; }
          }
          __syncthreads();
          const double dz = pv0 * as_z[pi0] + pv1 * as_z[pi1];   // c_p' z = sigma - sv' r  (>= 0)
          const double t1 = as_sc[2];
          const int lsel = (int)as_sc[3];
          const bool have_z = dz > 1e-13 * (1.0 + fabs(sigma));
          const double t2 = have_z ? -s_p / dz : INFINITY;
          const double tt = fmin(t1, t2);
          if (!(tt < INFINITY)) { failed = true; break; }
          // S9: take the step
          if (have_z) { x = fma(tt, zz, x); if (owner) as_x[io] = x; s_p = fma(tt, dz, s_p); }
          if (tid < q) as_lam[tid] -= tt * as_r[tid];
          lam_p += tt;
          const bool full = have_z && (t2 <= t1);
          __syncthreads();
          if (full) {
            // add p: T <- [[T + r r'/dz, -r/dz], [-r'/dz, 1/dz]]   (row i of the packed triangle on lane i)
            const double idz = 1.0 / dz;
            for (int i2 = tid; i2 < q; i2 += NT) {
              const double ri = as_r[i2] * idz;
              double *Trow = as_T + i2 * (i2 + 1) / 2;
              int j2 = 0;
              for (; j2 + 3 <= i2; j2 += 4) {   // four independent read-modify-writes in flight
                const double t0 = Trow[j2], t1_ = Trow[j2 + 1], t2_ = Trow[j2 + 2], t3 = Trow[j2 + 3];
                const double r0 = as_r[j2], r1 = as_r[j2 + 1], r2 = as_r[j2 + 2], r3 = as_r[j2 + 3];
                Trow[j2] = fma(ri, r0, t0); Trow[j2 + 1] = fma(ri, r1, t1_); Trow[j2 + 2] = fma(ri, r2, t2_); Trow[j2 + 3] = fma(ri, r3, t3);
              }
              for (; j2 <= i2; j2++) Trow[j2] = fma(ri, as_r[j2], Trow[j2]);
            }
            if (tid < q) as_T[q * (q + 1) / 2 + tid] = -as_r[tid] * idz;
            if (tid == 0) {
              as_T[q * (q + 1) / 2 + q] = idz;
              as_i0[q] = pi0; as_i1[q] = pi1; as_v0[q] = pv0; as_v1[q] = pv1; as_id[q] = pid; as_lam[q] = lam_p;
              as_am[pblk] |= 1 << pty;
            }
            q++;
            it++;
            need_search = true;
            __syncthreads();
          } else {
            // partial step: drop constraint l = lsel (its multiplier reached 0), keep working on p
            const int l = lsel, last = q - 1;
            const double itau = 1.0 / as_T[l * (l + 1) / 2 + l];
            if (tid < q) as_sv[tid] = (tid <= l) ? as_T[l * (l + 1) / 2 + tid] : as_T[tid * (tid + 1) / 2 + l];   // column l
            __syncthreads();
            for (int i2 = tid; i2 < q; i2 += NT) {
              if (i2 == l) continue;
              const double ci = -as_sv[i2] * itau;
              double *Trow = as_T + i2 * (i2 + 1) / 2;
              int j2 = 0;
              for (; j2 + 3 <= i2; j2 += 4) {   // entries in row/column l are dead after the drop: updating them is harmless
                const double t0 = Trow[j2], t1_ = Trow[j2 + 1], t2_ = Trow[j2 + 2], t3 = Trow[j2 + 3];
                const double c0 = as_sv[j2], c1_ = as_sv[j2 + 1], c2_ = as_sv[j2 + 2], c3 = as_sv[j2 + 3];
                Trow[j2] = fma(ci, c0, t0); Trow[j2 + 1] = fma(ci, c1_, t1_); Trow[j2 + 2] = fma(ci, c2_, t2_); Trow[j2 + 3] = fma(ci, c3, t3);
              }
              for (; j2 <= i2; j2++) Trow[j2] = fma(ci, as_sv[j2], Trow[j2]);
            }
            __syncthreads();
            if (l != last) {   // move the last active constraint into slot l
              if (tid < last && tid != l) {
                const double v = as_T[last * (last + 1) / 2 + tid];
                if (tid < l) as_T[l * (l + 1) / 2 + tid] = v; else as_T[tid * (tid + 1) / 2 + l] = v;
              }
              if (tid == 0) as_T[l * (l + 1) / 2 + l] = as_T[last * (last + 1) / 2 + last];
            }
            if (tid == 0) {
              const int did = as_id[l];
              as_am[did / 6] &= ~(1 << (did % 6));
              if (l != last) { as_i0[l] = as_i0[last]; as_i1[l] = as_i1[last]; as_v0[l] = as_v0[last]; as_v1[l] = as_v1[last]; as_id[l] = as_id[last]; as_lam[l] = as_lam[last]; }
            }
            q--;
            __syncthreads();
          }
        }
        if (need_search) {
          // --- S1: most violated inactive constraint (wave 0) ---
          if (tid < 64) {
            double best = 0.0; int bid = -1;
            for (int blk2 = tid; blk2 < NB; blk2 += 64) {
              const double fx = as_x[3 * blk2], fy = as_x[3 * blk2 + 1], fz = as_x[3 * blk2 + 2];
              const int am = as_am[blk2];
              const double sv6[6] = {-fx + mu * fz, fx + mu * fz, -fy + mu * fz, fy + mu * fz, fz - lo, hi - fz};
#pragma unroll
              for (int ty = 0; ty < 6; ty++) if (!((am >> ty) & 1) && sv6[ty] < best) { best = sv6[ty]; bid = 6 * blk2 + ty; }
            }
            {
              const double bmin = wave_min_f64(best);
              const unsigned long long hit = __ballot(best == bmin && bid >= 0);
              bid = hit ? __builtin_amdgcn_readlane(bid, __ffsll((long long)hit) - 1) : -1;
              best = bmin;
            }
            if (tid == 0) { as_sc[0] = best; as_sc[1] = (double)bid; }
          }
          __syncthreads();
          s_p = as_sc[0];
          pid = (int)as_sc[1];
          if (pid < 0 || s_p >= -vtol) break;
          pblk = pid / 6; pty = pid % 6;
          pi0 = (pty < 2) ? 3 * pblk : (pty < 4 ? 3 * pblk + 1 : 3 * pblk + 2);
          pi1 = 3 * pblk + 2;
          pv0 = (pty == 0 || pty == 2 || pty == 5) ? -1.0 : 1.0;
          pv1 = (pty < 4) ? mu : 0.0;
          lam_p = 0.0;
          // --- S2: publish rows pi0 (and pi1) of G; S3: g = G c_p ---
          {
            const int r0l = pi0 / T, r0a = pi0 % T, r1l = pi1 / T, r1a = pi1 % T;
#pragma unroll
            for (int ta = 0; ta < T; ta++) {
              if (lr == r0l && ta == r0a) {
#pragma unroll
                for (int tb = 0; tb < T; tb++) as_pg[lc * TS + tb] = -(tile[ta][tb] - ((lc == lr && tb == ta) ? 2.0 : 0.0));
              }
              if (pv1 != 0.0 && lr == r1l && ta == r1a) {
#pragma unroll
                for (int tb = 0; tb < T; tb++) as_pg[NPAD + lc * TS + tb] = -(tile[ta][tb] - ((lc == lr && tb == ta) ? 2.0 : 0.0));
              }
            }
          }
          __syncthreads();
          if (owner) {
            const int ipad = (io / T) * TS + io % T;
            double gi = pv0 * as_pg[ipad];
            if (pv1 != 0.0) gi += pv1 * as_pg[NPAD + ipad];
            as_g[io] = own_real ? gi : 0.0;
          }
          __syncthreads();
          sigma = pv0 * as_g[pi0] + pv1 * as_g[pi1];
        }
      }
      if (failed && tid == 0) atomicAdd(&st.counts[7], 1);
      z = x;
    } else {
    // ---- over-relaxed ADMM; scalar state lives on the owner lane of each row ----
    // look-ahead extension: a (step, leg) block whose leg is not in contact at that step is pinned to 0
    bool enabled = true;
    if (c->lookahead && own_real) enabled = (((int)rec[REC_SCHED + (io % m3) / 3]) >> (io / m3)) & 1;
    z = (own_real && enabled && (io % 3) == 2) ? lo : 0.0;
    double y = 0.0;
    const int blk = own_real ? io - io % 3 : 0, dax = io % 3;
    if (owner) vv[iov_pad] = own_real ? rho * (z - y) - qi : 0.0;
    __syncthreads();
    const double atol = c->admm_abs_tol;
    const int chk = c->admm_check;
    double zchk = z;
    int next_chk = chk;
    it = 0;
    bool converged = false;
    for (; it < c->admm_iters; it++) {
      double acc[T];
      {
        double vloc[T];
#pragma unroll
        for (int t2 = 0; t2 < T; t2 += 2) {
          double2 v2 = *reinterpret_cast<const double2 *>(&vv[lc * TS + t2]);
          vloc[t2] = v2.x; vloc[t2 + 1] = v2.y;
        }
#pragma unroll
        for (int ta = 0; ta < T; ta++) {
          double a0 = 0.0;
#pragma unroll
          for (int tb = 0; tb < T; tb++) a0 = fma(tile[ta][tb], vloc[tb], a0);
          acc[ta] = (lr == lc) ? a0 - 2.0 * vloc[ta] : a0;
        }
      }
      // reduce over the LC lanes of this lane-row
      double tot;
      if constexpr (T == 8) {
        // reduce-scatter over the 8 lanes of a half-row, all in DPP (no LDS round trips):
        // step 1 pairs i <-> 7-i (row_half_mirror) and splits by bit 2, step 2 pairs i^1 / bit 0,
        // step 3 pairs i^2 / bit 1.  Lane-rows of 16/32 lanes finish with all-reduce steps.
        {
          const bool up = (lc >> 2) & 1;
#pragma unroll
          for (int h2 = 0; h2 < 4; h2++) {
            double keep = up ? acc[4 + h2] : acc[h2];
            double send = up ? acc[h2] : acc[4 + h2];
            acc[h2] = keep + dpp_f64<0x141>(send);
          }
        }
        {
          const bool up = lc & 1;
#pragma unroll
          for (int h2 = 0; h2 < 2; h2++) {
            double keep = up ? acc[2 + h2] : acc[h2];
            double send = up ? acc[h2] : acc[2 + h2];
            acc[h2] = keep + dpp_f64<0xB1>(send);
          }
        }
        {
          const bool up = (lc >> 1) & 1;
          double keep = up ? acc[1] : acc[0];
          double send = up ? acc[0] : acc[1];
          tot = keep + dpp_f64<0x4E>(send);
        }
        if constexpr (LG >= 4) tot += dpp_f64<0x128>(tot);
#pragma unroll
        for (int k = 4; k < LG; k++) tot += __shfl_xor(tot, 1 << k);
      } else if constexpr (TPOW2) {
        // reduce-scatter: after step k (xor 2^k) a lane keeps the half selected by bit k of lc
#pragma unroll
        for (int k = 0; k < LT; k++) {
          const int half = T >> (k + 1);
          const bool up = (lc >> k) & 1;
#pragma unroll
          for (int h2 = 0; h2 < half; h2++) {
            double keep = up ? acc[half + h2] : acc[h2];
            double send = up ? acc[h2] : acc[half + h2];
            acc[h2] = keep + __shfl_xor(send, 1 << k);
          }
        }
        tot = acc[0];
#pragma unroll
        for (int k = LT; k < LG; k++) tot += __shfl_xor(tot, 1 << k);
      } else {
#pragma unroll
        for (int ta = 0; ta < T; ta++) {
#pragma unroll
          for (int k = 0; k < LG; k++) acc[ta] += __shfl_xor(acc[ta], 1 << k);
        }
        tot = acc[0];
#pragma unroll
        for (int ta = 1; ta < T; ta++) tot = (own_a == ta) ? acc[ta] : tot;
      }
      const double u = -tot;
      const double w = relax * u + (1.0 - relax) * z + y;
      if (own_real) wv[io] = w;
      __syncthreads();
      if (own_real) {
        double px, py, pz;
        proj_pyramid(wv[blk], wv[blk + 1], wv[blk + 2], mu, lo, hi, kA, kB, px, py, pz);
        const double zn = enabled ? ((dax == 0) ? px : (dax == 1 ? py : pz)) : 0.0;
        y = w - zn;
        z = zn;
        vv[iov_pad] = rho * (z - y) - qi;
      }
      if (atol > 0.0 && it + 1 == next_chk) {
        // one robot per workgroup: a data-dependent exit costs no divergence, only this vote
        const int moving = own_real && fabs(z - zchk) > atol;
        zchk = z;
        next_chk += chk;
        if (!__syncthreads_or(moving)) { it++; converged = true; break; }
      } else __syncthreads();
    }
    if (c->solver == RG_SOLVER_AUTO && atol > 0.0 && !converged && tid == 0) {   // hand the robot to the exact solver
      const int slot = atomicAdd(&st.counts[8 + NC], 1);
      st.bins[(size_t)(5 + NC) * B + slot] = b;
    }
    }
    if (tid == 0) { atomicAdd(&st.counts[5], it); atomicMax(&st.counts[6], it); }
    if (own_real && io < m3) grf[3 * nth_leg(cmask, io / 3) + io % 3] = -z;
    __syncthreads();
    if (tid < 12) {
      int leg = tid / 3, j = tid % 3;
      const double *J = &rec[REC_JAC + 9 * leg];
      double tau = (grf[3 * leg] * J[j] + grf[3 * leg + 1] * J[3 + j] + grf[3 * leg + 2] * J[6 + j]) * c->mdir[tid];
      grf[12 + tid] = tau;
      if (out.grf) out.grf[(size_t)b * 12 + tid] = (float)grf[tid];
      if (out.tau_stance) out.tau_stance[(size_t)b * 12 + tid] = (float)tau;
    }
    __syncthreads();
    if (tid < 60) {
      int j = tid / 5, f = tid % 5;
      int emit = ((int)rec[REC_EMIT] >> j) & 1;
      float v;
      if (emit) v = (f == 0) ? (float)rec[REC_SWINGQ + j] : (f == 1 ? (float)c->kp[j] : (f == 3 ? (float)c->kd[j] : 0.f));
      else v = (f == 4) ? (float)grf[12 + j] : 0.f;
      out.action[(size_t)b * 60 + tid] = v;
    }
  }
}

template <int NC, int H, int T, int LG, int MINW, bool AS>
static hipError_t launch_qp_tile_impl(const DevCfg *dcfg, const DevState &st, const DevOut &dout, int B, int cu_count, hipStream_t s, bool retry) {
  constexpr int m3 = 3 * NC, N = m3 * H, LC = 1 << LG, NP = T * LC, NT = LC * LC;
  constexpr int TS = (T == 8) ? 10 : T, NPAD = TS * LC;
  size_t lds = sizeof(double) * (size_t)(2 * (NPAD + 2) + NPAD + NP + 2 * m3 * m3 + 2 * N + 6 * m3 + RG_REC_N + 24 + 2 * H * H);
  if (AS) lds += sizeof(double) * (size_t)(3 * NP + 2 * NPAD + 5 * N + N * (N + 1) / 2 + 8) + sizeof(int) * (size_t)(3 * N + ((N / 3 + 1) & ~1));
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static size_t attr_set = 0;
  if (lds > attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)rg_qp_admm_tile_kernel<NC, H, T, LG, MINW, AS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = lds;
  }
  int grid = cu_count * 8;
  if (retry) grid = cu_count / 2;   // retry lists are short; a full grid of large-LDS workgroups takes tens of us just to drain
  if (grid > B) grid = B;
  if (grid < 1) grid = 1;
  rg_qp_admm_tile_kernel<NC, H, T, LG, MINW, AS><<<dim3(grid), dim3(NT), lds, s>>>(dcfg, st, dout, B, retry ? 1 : 0);
  return hipGetLastError();
}

template <int NC, int H, int T, int LG, int MINW>
static hipError_t launch_qp_tile(const DevCfg *dcfg, const DevState &st, const DevOut &dout, int B, int cu_count, hipStream_t s, int active_set) {
  // active_set: 0 = ADMM kernel, 1 = active-set kernel on the main list, 2 = active-set kernel on the retry list
  if (active_set) {
    if constexpr (H == 10) return launch_qp_tile_impl<NC, H, T, LG, MINW, true>(dcfg, st, dout, B, cu_count, s, active_set == 2);
    else return hipErrorInvalidValue;
  }
  return launch_qp_tile_impl<NC, H, T, LG, MINW, false>(dcfg, st, dout, B, cu_count, s, false);
}

static bool launch_qp_tile_dispatch(int nc, int H, const DevCfg *dcfg, const DevState &st, const DevOut &dout, int B, int cu, hipStream_t s, hipError_t *err, int active_set) {
  *err = hipSuccess;
  if (H == 10) {
    switch (nc) {
      case 1: *err = launch_qp_tile<1, 10, 4, 3, 2>(dcfg, st, dout, B, cu, s, active_set); return true;   // 30 -> 32
      case 2: *err = launch_qp_tile<2, 10, 8, 3, 2>(dcfg, st, dout, B, cu, s, active_set); return true;   // 60 -> 64, one wave
      case 3: *err = launch_qp_tile<3, 10, 6, 4, 2>(dcfg, st, dout, B, cu, s, active_set); return true;   // 90 -> 96, four waves
      case 4: *err = launch_qp_tile<4, 10, 8, 4, 2>(dcfg, st, dout, B, cu, s, active_set); return true;   // 120 -> 128, four waves
    }
  } else if (H == 20) {
    switch (nc) {
      case 1: *err = launch_qp_tile<1, 20, 8, 3, 2>(dcfg, st, dout, B, cu, s, active_set); return true;   // 60 -> 64
      case 2: *err = launch_qp_tile<2, 20, 8, 4, 2>(dcfg, st, dout, B, cu, s, active_set); return true;   // 120 -> 128
      case 3: *err = launch_qp_tile<3, 20, 6, 5, 1>(dcfg, st, dout, B, cu, s, active_set); return true;   // 180 -> 192, sixteen waves
      case 4: *err = launch_qp_tile<4, 20, 8, 5, 1>(dcfg, st, dout, B, cu, s, active_set); return true;   // 240 -> 256, sixteen waves
    }
  }
  return false;
}

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
  for (int l = 0; l < 4; l++) ld |= (c->init_state[l] & 1) << l;
  st.last_desired[b] = ld;
  st.ring_len[b] = 0; st.ring_head[b] = 0;
  for (int a = 0; a < 3; a++) { st.fsum[a * B + b] = 0.0; st.fcorr[a * B + b] = 0.0; }
  st.swing_valid[b] = 0;
}

// RobotMotorModel.convert_to_torque HYBRID (reference model/robots/simple_motor.py:128-140)
__global__ void rg_hybrid_to_torque_kernel(const float *__restrict__ action, const float *__restrict__ q,
                                           const float *__restrict__ qd, float *__restrict__ tau, int B) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * 12) return;
  int b = e / 12, j = e % 12;
  const float *a = action + (size_t)b * 60 + 5 * j;
  double qs = a[0], kp = a[1], qds = a[2], kd = a[3], ff = a[4];
  double t = -1.0 * (kp * ((double)q[j * B + b] - qs)) - kd * ((double)qd[j * B + b] - qds) + ff;
  tau[e] = (float)t;
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
  size_t lds_bytes[5] = {0, 0, 0, 0, 0};
  std::vector<void *> allocs;
  std::string err;
  // optional per-kernel event timing
  std::vector<hipEvent_t> ev;   // RG_PROF_EV events per profiled step
  int prof_max = 0, prof_n = 0;
  bool force_lds_kernel = false;
  int qp_variant = 0;
  bool auto_retry = false;          // RG_SOLVER_AUTO with an active-set instantiation available
  bool concurrent_bins = false;     // opt-in: run the per-stance-count QP launches on forked streams
  hipStream_t aux[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_front = nullptr, ev_done[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

static thread_local std::string g_create_err;

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
  if (c->horizon < 1 || c->horizon > RG_MAXH) { err = "horizon out of range [1,20]"; return RG_MPC_ERR_INVALID; }
  if (c->window < 1 || c->window > 64) { err = "window out of range [1,64]"; return RG_MPC_ERR_INVALID; }
  if (!(c->mu[0] == c->mu[1] && c->mu[1] == c->mu[2] && c->mu[2] == c->mu[3]) || !(c->mu[0] > 0)) { err = "friction coefficients must be equal and positive"; return RG_MPC_ERR_INVALID; }
  if (!(c->mass > 0) || !(c->dt_plan > 0) || !(c->alpha > 0)) { err = "mass, dt_plan and alpha must be positive"; return RG_MPC_ERR_INVALID; }
  if (c->kin_mode != 0 && c->kin_mode != 1) { err = "kin_mode must be 0 or 1"; return RG_MPC_ERR_INVALID; }
  if (c->solver != RG_SOLVER_ADMM && c->solver != RG_SOLVER_ACTIVE_SET && c->solver != RG_SOLVER_AUTO) { err = "unsupported solver"; return RG_MPC_ERR_INVALID; }
  if (c->solver == RG_SOLVER_ACTIVE_SET && ((c->reserved0 & 7) != 0 || c->horizon != 10 || c->contact_lookahead)) { err = "the active-set solver needs the tiled QP kernel, horizon 10 and no contact look-ahead"; return RG_MPC_ERR_INVALID; }
  if (c->contact_lookahead && ((c->reserved0 & 7) != 0 || (c->horizon != 10 && c->horizon != 20))) { err = "contact_lookahead needs the tiled QP kernel (reserved0 bits 0-2 clear) and horizon 10 or 20"; return RG_MPC_ERR_INVALID; }
  if (!(c->admm_rho > 0) || c->admm_iters < 1 || !(c->admm_relax > 0 && c->admm_relax < 2) || !(c->admm_tol >= 0) || (c->admm_tol > 0 && c->admm_check < 1)) { err = "bad ADMM parameters"; return RG_MPC_ERR_INVALID; }
  for (int i = 0; i < 12; i++) if (!(c->motor_dir[i] == 1.0 || c->motor_dir[i] == -1.0)) { err = "motor_dir must be +-1"; return RG_MPC_ERR_INVALID; }
  for (int i = 0; i < 4; i++) {
    if (!(c->duty_factor[i] > 0 && c->duty_factor[i] <= 1) || !(c->stance_duration[i] > 0)) { err = "bad gait timing"; return RG_MPC_ERR_INVALID; }
    if (c->init_state[i] != RG_LEG_SWING && c->init_state[i] != RG_LEG_STANCE) { err = "init_state must be SWING or STANCE"; return RG_MPC_ERR_INVALID; }
  }
  d->H = c->horizon; d->window = c->window; d->kin_mode = c->kin_mode; d->ik_iters = c->ik_iters; d->admm_iters = c->admm_iters;
  d->dt = c->dt_plan; d->mass = c->mass; d->inv_mass = 1.0 / c->mass; d->body_height = c->body_height; d->alpha = c->alpha;
  d->mu = c->mu[0]; d->g = c->gravity;
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
  d->rho = c->admm_rho; d->relax = c->admm_relax;
  d->admm_abs_tol = c->admm_tol * c->mass * c->gravity; d->admm_check = c->admm_check; d->lookahead = c->contact_lookahead ? 1 : 0; d->solver = c->solver;
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

static size_t qp_lds_bytes(int nc, int H) {
  size_t m3 = 3 * nc, n = m3 * H, LD = n | 1;
  size_t dbl = n * LD + 2 * n + 2 * m3 * m3 + 2 * H * m3 + 6 * m3 + RG_REC_N + 24;
  return dbl * sizeof(double);
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
const char *rg_mpc_kernel_names(void) { return "rg_front_kernel,rg_qp_admm_tile_kernel,rg_qp_admm_reg_kernel,rg_qp_admm_kernel"; }

const char *rg_mpc_last_error(const rg_mpc_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int rg_mpc_create(const rg_mpc_config *cfg, int32_t batch, int32_t device, rg_mpc_handle **out) {
  if (!cfg || !out || batch < 1) { g_create_err = "null config/out or batch < 1"; return RG_MPC_ERR_INVALID; }
  *out = nullptr;
  rg_mpc_handle *h = new rg_mpc_handle();
  h->cfg = *cfg; h->B = batch; h->device = device;
  h->force_lds_kernel = (cfg->reserved0 & 1) != 0;
  h->auto_retry = cfg->solver == RG_SOLVER_AUTO && (cfg->reserved0 & 7) == 0 && cfg->horizon == 10 && !cfg->contact_lookahead;
  h->qp_variant = (cfg->reserved0 >> 1) & 3;
  h->concurrent_bins = ((cfg->reserved0 >> 3) & 1) != 0;  // bit3: fork the QP launches onto internal streams (measured slower: the bins compete for the same LDS/VALU)         // bits1-2: register-kernel tiling variant (tuning A/B)  // bit0: use the LDS-resident QP kernel (A/B and generic-H path)
  int rc = build_devcfg(cfg, &h->hcfg, h->err);
  if (rc) { g_create_err = h->err; delete h; return rc; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_err = "no HIP device available"; delete h; return RG_MPC_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; delete h; return RG_MPC_ERR_INVALID; }
#define CR(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_err = std::string(#call " failed: ") + hipGetErrorString(e_); rg_mpc_destroy(h); return RG_MPC_ERR_HIP; } } while (0)
#define AL(p, n) do { int r_ = dev_alloc(h, &(p), (n)); if (r_) { g_create_err = h->err; rg_mpc_destroy(h); return r_; } } while (0)
  CR(hipSetDevice(device));
  hipDeviceProp_t prop;
  CR(hipGetDeviceProperties(&prop, device));
  h->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const size_t B = batch, W = cfg->window;
  AL(h->dcfg, 1);
  CR(hipMemcpy(h->dcfg, &h->hcfg, sizeof(DevCfg), hipMemcpyHostToDevice));
  AL(h->st.reset_time, B); AL(h->st.flags, B); AL(h->st.last_desired, B);
  AL(h->st.ring, 3 * W * B); AL(h->st.ring_len, B); AL(h->st.ring_head, B);
  AL(h->st.fsum, 3 * B); AL(h->st.fcorr, 3 * B);
  AL(h->st.latched, 12 * B); AL(h->st.swing_q, 12 * B); AL(h->st.swing_valid, B);
  AL(h->st.cmd, 3 * B); AL(h->st.rec, B * RG_REC_N); AL(h->st.bins, 10 * B); AL(h->st.counts, 16);
  AL(h->idx_dev, B); AL(h->t0_dev, B);
  CR(hipEventCreateWithFlags(&h->ev_front, hipEventDisableTiming));
  for (int nc = 1; nc <= 4; nc++) {
    CR(hipStreamCreateWithFlags(&h->aux[nc], hipStreamNonBlocking));
    CR(hipEventCreateWithFlags(&h->ev_done[nc], hipEventDisableTiming));
  }
  for (int nc = 1; nc <= 4; nc++) {
    size_t bytes = qp_lds_bytes(nc, cfg->horizon);
    h->lds_bytes[nc] = bytes;
    if (bytes > 160 * 1024) { h->lds_bytes[nc] = 0; continue; } // unsupported size: rejected at step time if it occurs
    if (3 * nc * cfg->horizon <= 64) CR(hipFuncSetAttribute((const void *)rg_qp_admm_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    else if (3 * nc * cfg->horizon <= 128) CR(hipFuncSetAttribute((const void *)rg_qp_admm_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    else CR(hipFuncSetAttribute((const void *)rg_qp_admm_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
#undef CR
#undef AL
  *out = h;
  int r = rg_mpc_reset(h, nullptr, batch, 0.0, nullptr);
  if (r) { g_create_err = h->err; rg_mpc_destroy(h); *out = nullptr; return r; }
  hipDeviceSynchronize();
  return RG_MPC_OK;
}

void rg_mpc_destroy(rg_mpc_handle *h) {
  if (!h) return;
  hipSetDevice(h->device);
  for (void *p : h->allocs) (void)hipFree(p);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  if (h->ev_front) (void)hipEventDestroy(h->ev_front);
  for (int nc = 1; nc <= 4; nc++) {
    if (h->ev_done[nc]) (void)hipEventDestroy(h->ev_done[nc]);
    if (h->aux[nc]) (void)hipStreamDestroy(h->aux[nc]);
  }
  delete h;
}

static int reset_impl(rg_mpc_handle *h, const int32_t *idx_host, const double *t0_host, int32_t n, double t0, void *stream) {
  if (!h) return RG_MPC_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
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
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(h->st.cmd, cmd, sizeof(float) * 3 * h->B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return RG_MPC_OK;
}

int rg_mpc_step(rg_mpc_handle *h, double t, const rg_mpc_state_ptrs *in, const rg_mpc_out_ptrs *out, void *stream) {
  if (!h || !in || !out) { if (h) h->err = "step: null argument"; return RG_MPC_ERR_INVALID; }
  if (!in->rpy || !in->rpy_rate || !in->v_world || !in->quat || !in->q || !in->contact) { h->err = "step: missing required state pointer"; return RG_MPC_ERR_INVALID; }
  if (h->cfg.kin_mode == 0 && (!in->foot_pos || !in->jac)) { h->err = "step: kin_mode 0 needs foot_pos and jac"; return RG_MPC_ERR_INVALID; }
  if (!out->action) { h->err = "step: action output required"; return RG_MPC_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  const int B = h->B, H = h->cfg.horizon;
  DevIn di{in->rpy, in->rpy_rate, in->v_world, in->quat, in->q, in->foot_pos, in->jac, in->cmd, in->contact};
  DevOut dout{out->action, out->grf, out->tau_stance, out->phase, out->foot_target, out->v_body, out->leg_state, out->desired_state};
  HIPCHK(h, hipMemsetAsync(h->st.counts, 0, sizeof(int) * 16, s));
  hipEvent_t *pev = (h->prof_n < h->prof_max) ? &h->ev[(size_t)h->prof_n * RG_PROF_EV] : nullptr;
  if (pev) HIPCHK(h, hipEventRecord(pev[0], s));
  hipLaunchKernelGGL(rg_front_kernel, dim3((4 * B + 255) / 256), dim3(256), 0, s, h->dcfg, h->st, di, dout, t, B);
  HIPCHK(h, hipGetLastError());
  if (pev) HIPCHK(h, hipEventRecord(pev[1], s));
  // Robots with different stance-leg counts are independent; optionally (reserved0 bit3) the four QP
  // launches are forked onto internal streams and joined back into the caller's stream.  Measured
  // on MI355X this is ~8 % slower than back-to-back launches, so it is off by default.
  const bool fork = h->concurrent_bins;
  if (fork) HIPCHK(h, hipEventRecord(h->ev_front, s));
  const int order[4] = {4, 2, 3, 1};  // longest first
  for (int oi = 0; oi < 4; oi++) {
    const int nc = fork ? order[oi] : oi + 1;
    const int n = 3 * nc * H;
    hipStream_t qs = s;
    if (fork) { qs = h->aux[nc]; HIPCHK(h, hipStreamWaitEvent(qs, h->ev_front, 0)); }
    if (pev) HIPCHK(h, hipEventRecord(pev[2 * nc], qs));
    bool launched = false;
    if (!h->force_lds_kernel && h->qp_variant != 2 && h->qp_variant != 1) {
      hipError_t lerr;
      if (launch_qp_tile_dispatch(nc, H, h->dcfg, h->st, dout, B, h->cu_count, qs, &lerr, h->cfg.solver == RG_SOLVER_ACTIVE_SET ? 1 : 0)) {
        HIPCHK(h, lerr);
        launched = true;
        if (h->cfg.solver == RG_SOLVER_AUTO && h->auto_retry) {   // exact re-solve of the robots ADMM left unconverged
          launch_qp_tile_dispatch(nc, H, h->dcfg, h->st, dout, B, h->cu_count, qs, &lerr, 2);
          HIPCHK(h, lerr);
        }
      }
    }
    if (!launched && !h->force_lds_kernel) {
      hipError_t lerr;
      if (launch_qp_reg_dispatch(h->qp_variant, nc, H, h->dcfg, h->st, dout, B, h->cu_count, qs, &lerr)) {
        HIPCHK(h, lerr);
        launched = true;
      }
    }
    size_t lds = h->lds_bytes[nc];
    if (!launched && lds != 0) {
      int per_cu = (int)((160 * 1024) / lds);
      if (per_cu < 1) per_cu = 1;
      if (per_cu > 8) per_cu = 8;
      int grid = h->cu_count * per_cu;
      if (grid > B) grid = B;
      if (n <= 64) hipLaunchKernelGGL(rg_qp_admm_kernel<64>, dim3(grid), dim3(64), lds, qs, h->dcfg, h->st, dout, nc, B);
      else if (n <= 128) hipLaunchKernelGGL(rg_qp_admm_kernel<128>, dim3(grid), dim3(128), lds, qs, h->dcfg, h->st, dout, nc, B);
      else hipLaunchKernelGGL(rg_qp_admm_kernel<256>, dim3(grid), dim3(256), lds, qs, h->dcfg, h->st, dout, nc, B);
      HIPCHK(h, hipGetLastError());
    }
    // (no instantiation and no LDS fit: H = 20 with 3-4 stance legs -- robots of that bin keep their previous action)
    if (pev) HIPCHK(h, hipEventRecord(pev[2 * nc + 1], qs));
    if (fork) { HIPCHK(h, hipEventRecord(h->ev_done[nc], qs)); HIPCHK(h, hipStreamWaitEvent(s, h->ev_done[nc], 0)); }
  }
  if (pev) HIPCHK(h, hipEventRecord(pev[10], s));
  if (pev) h->prof_n++;
  return RG_MPC_OK;
}

int rg_mpc_profile_begin(rg_mpc_handle *h, int32_t max_steps) {
  if (!h || max_steps < 1 || max_steps > 100000) { if (h) h->err = "profile_begin: bad max_steps"; return RG_MPC_ERR_INVALID; }
  HIPCHK(h, hipSetDevice(h->device));
  while ((int)h->ev.size() < max_steps * RG_PROF_EV) {
    hipEvent_t e;
    HIPCHK(h, hipEventCreate(&e));
    h->ev.push_back(e);
  }
  h->prof_max = max_steps; h->prof_n = 0;
  return RG_MPC_OK;
}

int rg_mpc_profile_end(rg_mpc_handle *h, float *avg_ms6, int32_t *robots5, void *stream) {
  if (!h || !avg_ms6) { if (h) h->err = "profile_end: null output"; return RG_MPC_ERR_INVALID; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  int n = h->prof_n;
  h->prof_max = 0;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < n; k++) {
    hipEvent_t *e = &h->ev[(size_t)k * RG_PROF_EV];
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, e[0], e[1])); acc[0] += ms;
    for (int nc = 1; nc <= 4; nc++) { HIPCHK(h, hipEventElapsedTime(&ms, e[2 * nc], e[2 * nc + 1])); acc[nc] += ms; }
    HIPCHK(h, hipEventElapsedTime(&ms, e[0], e[10])); acc[5] += ms;
  }
  for (int j = 0; j < 6; j++) avg_ms6[j] = n > 0 ? (float)(acc[j] / n) : 0.f;
  if (robots5) HIPCHK(h, hipMemcpy(robots5, h->st.counts, sizeof(int) * 5, hipMemcpyDeviceToHost));
  return n;
}

int rg_mpc_hybrid_to_torque(rg_mpc_handle *h, const float *action, const float *q, const float *qd, float *tau, void *stream) {
  if (!h || !action || !q || !qd || !tau) { if (h) h->err = "hybrid_to_torque: null pointer"; return RG_MPC_ERR_INVALID; }
  HIPCHK(h, hipSetDevice(h->device));
  int total = h->B * 12;
  hipLaunchKernelGGL(rg_hybrid_to_torque_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, action, q, qd, tau, h->B);
  HIPCHK(h, hipGetLastError());
  return RG_MPC_OK;
}

int rg_mpc_last_solver_stats(rg_mpc_handle *h, int64_t *iters_sum, int32_t *iters_max, int32_t *qp_robots,
                             int32_t *retried, int32_t *failures, void *stream) {
  if (!h) return RG_MPC_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  int cnt[16];
  HIPCHK(h, hipMemcpy(cnt, h->st.counts, sizeof(cnt), hipMemcpyDeviceToHost));
  if (iters_sum) *iters_sum = cnt[5];
  if (iters_max) *iters_max = cnt[6];
  if (qp_robots) *qp_robots = cnt[1] + cnt[2] + cnt[3] + cnt[4];
  if (retried) *retried = cnt[9] + cnt[10] + cnt[11] + cnt[12];
  if (failures) *failures = cnt[7];
  return RG_MPC_OK;
}

int rg_mpc_last_bin_counts(rg_mpc_handle *h, int32_t *out5, void *stream) {
  if (!h || !out5) return RG_MPC_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(h, hipMemcpy(out5, h->st.counts, sizeof(int) * 5, hipMemcpyDeviceToHost));
  return RG_MPC_OK;
}

} // extern "C"
