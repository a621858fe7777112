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

__global__ void __launch_bounds__(64)
rg_front_kernel(const DevCfg *__restrict__ c, DevState st, DevIn in, DevOut out, double t_now, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int W = c->window;
  // ---- inputs (component-major, coalesced over robots) ----
  double rpy[3], rate[3], vw[3], quat[4], q[12], cmd[3];
  int contact[4];
#pragma unroll
  for (int i = 0; i < 3; i++) { rpy[i] = in.rpy[i * B + b]; rate[i] = in.rpy_rate[i * B + b]; vw[i] = in.v_world[i * B + b]; }
#pragma unroll
  for (int i = 0; i < 4; i++) { quat[i] = in.quat[i * B + b]; contact[i] = in.contact[i * B + b]; }
#pragma unroll
  for (int i = 0; i < 12; i++) q[i] = in.q[i * B + b];
  const float *cmdp = in.cmd ? in.cmd : st.cmd;
#pragma unroll
  for (int i = 0; i < 3; i++) cmd[i] = cmdp[i * B + b];
  double foot[12], jac[36];
  if (c->kin_mode == 1) {
    for (int leg = 0; leg < 4; leg++) leg_fk(c, leg, &q[3 * leg], &foot[3 * leg], &jac[9 * leg]);
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) foot[i] = in.foot_pos[i * B + b];
#pragma unroll
    for (int i = 0; i < 36; i++) jac[i] = in.jac[i * B + b];
  }
  int flags = st.flags[b];
  if (flags & 1) {
#pragma unroll
    for (int i = 0; i < 12; i++) st.latched[i * B + b] = foot[i];
  }
  // ---- LocomotionController.update(): gait ----
  double t = t_now - st.reset_time[b];
  int desired[4], lstate[4];
  double phase[4];
#pragma unroll
  for (int leg = 0; leg < 4; leg++) gait_leg(c, leg, t, contact[leg], desired[leg], lstate[leg], phase[leg]);
  // ---- velocity estimator: moving window (Neumaier), divide by window size always ----
  int rlen = st.ring_len[b], rhead = st.ring_head[b];
  double vf[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double s = st.fsum[a * B + b], cr = st.fcorr[a * B + b];
    size_t slot = ((size_t)a * W + rhead) * B + b;
    if (rlen >= W) neumaier_add(s, cr, -(double)st.ring[slot]);
    neumaier_add(s, cr, vw[a]);
    st.ring[slot] = (float)vw[a];
    st.fsum[a * B + b] = s; st.fcorr[a * B + b] = cr;
    vf[a] = (s + cr) / (double)W;
  }
  st.ring_head[b] = (rhead + 1) % W;
  if (rlen < W) st.ring_len[b] = rlen + 1;
  double vb[3];
  {
    double x = -quat[0], y = -quat[1], z = -quat[2], w = quat[3];
    double tx = 2 * (y * vf[2] - z * vf[1]), ty = 2 * (z * vf[0] - x * vf[2]), tz = 2 * (x * vf[1] - y * vf[0]);
    vb[0] = vf[0] + w * tx + (y * tz - z * ty);
    vb[1] = vf[1] + w * ty + (z * tx - x * tz);
    vb[2] = vf[2] + w * tz + (x * ty - y * tx);
  }
  // ---- swing update: latch at desired STANCE->SWING (skipped on the first update after reset) ----
  int last = st.last_desired[b];
  if (!(flags & 2)) {
#pragma unroll
    for (int leg = 0; leg < 4; leg++)
      if (desired[leg] == RG_LEG_SWING && ((last >> leg) & 1) != RG_LEG_SWING) {
#pragma unroll
        for (int a = 0; a < 3; a++) st.latched[(3 * leg + a) * B + b] = foot[3 * leg + a];
      }
  }
  int nl = 0;
#pragma unroll
  for (int leg = 0; leg < 4; leg++) nl |= (desired[leg] & 1) << leg;
  st.last_desired[b] = nl;
  st.flags[b] = 0;
  // ---- swing get_action ----
  int valid = st.swing_valid[b];
  double swq[12];
#pragma unroll
  for (int i = 0; i < 12; i++) swq[i] = st.swing_q[i * B + b];
  double ftarget[12];
#pragma unroll
  for (int i = 0; i < 12; i++) ftarget[i] = 0.0;
  for (int leg = 0; leg < 4; leg++) {
    if (lstate[leg] == RG_LEG_STANCE || lstate[leg] == RG_LEG_EARLY_CONTACT) continue;
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
    double ip = phase[leg], ph;
    if (ip <= 0.5) ph = 0.8 * sin(ip * M_PI); else ph = 0.8 + (ip - 0.5) * 0.4;
    double fp[3];
    fp[0] = (1 - ph) * start[0] + ph * target[0];
    fp[1] = (1 - ph) * start[1] + ph * target[1];
    {
      double mid = fmax(target[2], start[2]) + c->max_clearance;
      double d1 = mid - start[2], d2 = target[2] - start[2], d3 = 0.25 - 0.5;
      double ca = (d1 - d2 * 0.5) / d3, cb = (d2 * 0.25 - d1) / d3;
      fp[2] = ca * ph * ph + cb * ph + start[2];
    }
    double qo[3];
    leg_ik(c, leg, fp, &q[3 * leg], qo);
#pragma unroll
    for (int j = 0; j < 3; j++) { swq[3 * leg + j] = qo[j]; st.swing_q[(3 * leg + j) * B + b] = qo[j]; ftarget[3 * leg + j] = fp[j]; }
    valid |= 7 << (3 * leg);
  }
  st.swing_valid[b] = valid;
  int emit = 0;
#pragma unroll
  for (int j = 0; j < 12; j++) if (((valid >> j) & 1) && desired[j / 3] == RG_LEG_SWING) emit |= 1 << j;
  // ---- stance record ----
  int cmask = 0, nc = 0;
#pragma unroll
  for (int leg = 0; leg < 4; leg++) if (desired[leg] == RG_LEG_STANCE || desired[leg] == RG_LEG_EARLY_CONTACT) { cmask |= 1 << leg; nc++; }
  double sr, cr_, sp, cp;
  sincos(rpy[0], &sr, &cr_);
  sincos(rpy[1], &sp, &cp);
  // feet -> world-aligned frame with Rx(roll) Ry(pitch)   (yaw zeroed)
  double Rf[9] = {cp, 0, sp, sr * sp, cr_, -sr * cp, -cr_ * sp, sr, cr_ * cp};
  // body rotation for the inertia: Ry(pitch) Rx(roll)
  double Rb[9] = {cp, sp * sr, sp * cr_, 0, cr_, -sr, -sp, cp * sr, cp * cr_};
  double *rec = st.rec + (size_t)b * RG_REC_N;
  rec[REC_ROLL] = rpy[0]; rec[REC_PITCH] = rpy[1];
  double hz = 0;
#pragma unroll
  for (int leg = 0; leg < 4; leg++) {
    double fw[3];
    m3vec(Rf, &foot[3 * leg], fw);
    rec[REC_FEETW + 3 * leg] = fw[0]; rec[REC_FEETW + 3 * leg + 1] = fw[1]; rec[REC_FEETW + 3 * leg + 2] = fw[2];
    if ((cmask >> leg) & 1) hz += fw[2];
  }
  rec[REC_COMZ] = nc > 0 ? fabs(hz / nc) : 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++) { rec[REC_OMEGA + i] = rate[i]; rec[REC_VBODY + i] = vb[i]; rec[REC_CMD + i] = cmd[i]; }
  {
    double T1[9], Rt[9], Iw[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Rt[3 * i + j] = Rb[3 * j + i];
    m3mul(Rb, c->Iinv, T1);
    m3mul(T1, Rt, Iw);
#pragma unroll
    for (int i = 0; i < 9; i++) rec[REC_IWINV + i] = Iw[i];
  }
  rec[REC_INVCP] = 1.0 / cp;
  rec[REC_TANP] = sp / cp;
#pragma unroll
  for (int i = 0; i < 36; i++) rec[REC_JAC + i] = jac[i];
#pragma unroll
  for (int i = 0; i < 12; i++) rec[REC_SWINGQ + i] = swq[i];
  rec[REC_EMIT] = (double)emit;
  rec[REC_CONTACT] = (double)cmask;
  // ---- bin by number of stance legs ----
  int slot = atomicAdd(&st.counts[nc], 1);
  st.bins[(size_t)nc * B + slot] = b;
  // ---- optional outputs ----
#pragma unroll
  for (int leg = 0; leg < 4; leg++) {
    if (out.leg_state) out.leg_state[b * 4 + leg] = lstate[leg];
    if (out.desired_state) out.desired_state[b * 4 + leg] = desired[leg];
    if (out.phase) out.phase[b * 4 + leg] = (float)phase[leg];
  }
  if (out.foot_target)
#pragma unroll
    for (int i = 0; i < 12; i++) out.foot_target[b * 12 + i] = (float)ftarget[i];
  if (out.v_body)
#pragma unroll
    for (int i = 0; i < 3; i++) out.v_body[b * 3 + i] = (float)vb[i];
  if (nc == 0) {
    // no stance leg: forces are zero, the action row is complete here
    for (int j = 0; j < 12; j++) {
      float *a = out.action + (size_t)b * 60 + 5 * j;
      if ((emit >> j) & 1) { a[0] = (float)swq[j]; a[1] = (float)c->kp[j]; a[2] = 0.f; a[3] = (float)c->kd[j]; a[4] = 0.f; }
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
    int legs[4], k = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) if ((cmask >> l) & 1) legs[k++] = l;
    // ---- Bw = Iw^-1 [r_l]x ; TBw = T Bw,  T = [[1/cp,0,0],[0,1,0],[tan p,0,1]] ----
    if (tid < m3) {
      int l = legs[tid / 3], d = tid % 3;
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
    for (int it = 0; it < c->admm_iters; it++) {
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
    }
    // ---- first-step forces (negated), torques, action row ----
    if (tid < m3) grf[3 * legs[tid / 3] + tid % 3] = -z;
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

  for (int work = blockIdx.x; work < count; work += gridDim.x) {
    const int b = st.bins[(size_t)NC * B + work];
    __syncthreads();
    for (int e = tid; e < RG_REC_N; e += NT) rec[e] = st.rec[(size_t)b * RG_REC_N + e];
    if (tid < 24) grf[tid] = 0.0;
    __syncthreads();
    const int cmask = (int)rec[REC_CONTACT];
    int legs[4], kk = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) if ((cmask >> l) & 1) legs[kk++] = l;
    if (tid < m3) {
      int l = legs[tid / 3], d = tid % 3;
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
      const int a = active ? r / m3 : 0, i = active ? r % m3 : 0;
      for (int kq = a; kq < H; kq++) qi += c1[kq * m3 + i] + ((double)(kq - a) + 0.5) * c2[kq * m3 + i];
      qi *= 2.0;
#pragma unroll
      for (int jj = 0; jj < C; jj++) {
        const int bb = (col0 + jj) / m3, j = (col0 + jj) % m3;
        double v = tabN[a * H + bb] * GU[j * m3 + i] + tabS[a * H + bb] * GV[j * m3 + i];
        if (col0 + jj == r) v += c->alpha + rho;
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
    for (int kp = 0; kp < N; kp++) {
      double *pb = pbuf + (kp & 1) * NP;
      if (active && r == kp) {
#pragma unroll
        for (int jj = 0; jj < C; jj += 2) *reinterpret_cast<double2 *>(&pb[col0 + jj]) = make_double2(row[jj], row[jj + 1]);
        if (s == kp / C) { double d = pb[kp]; pb[kp] = d - 1.0; pb[N] = 1.0 / d; }
      }
      __syncthreads();
      if (active) {
        const double invd = pb[N];
        const double ncc = (r == kp) ? invd - 1.0 : -pb[r] * invd;
#pragma unroll
        for (int jj = 0; jj < C; jj += 2) {
          double2 p2 = *reinterpret_cast<const double2 *>(&pb[col0 + jj]);
          row[jj] = fma(ncc, p2.x, row[jj]);
          row[jj + 1] = fma(ncc, p2.y, row[jj + 1]);
        }
      }
    }
    // row now holds -(P + rho I)^-1 entries (diagonal offset by +2)
    // ---- over-relaxed ADMM ----
    double z = (active && (r % 3) == 2) ? lo : 0.0, y = 0.0;
    const int blk = active ? r - r % 3 : 0, dax = r % 3;
    for (int it = 0; it < c->admm_iters; it++) {
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
    }
    if (active && s == 0 && r < m3) grf[3 * legs[r / 3] + r % 3] = -z;
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
static bool launch_qp_reg_dispatch(int nc, int H, const DevCfg *dcfg, const DevState &st, const DevOut &dout, int B, int cu, hipStream_t s, hipError_t *err) {
  *err = hipSuccess;
  if (H == 10) {
    switch (nc) {
      case 1: *err = launch_qp_reg<1, 10, 1, 4>(dcfg, st, dout, B, cu, s); return true;
      case 2: *err = launch_qp_reg<2, 10, 2, 4>(dcfg, st, dout, B, cu, s); return true;
      case 3: *err = launch_qp_reg<3, 10, 1, 1>(dcfg, st, dout, B, cu, s); return true;
      case 4: *err = launch_qp_reg<4, 10, 4, 4>(dcfg, st, dout, B, cu, s); return true;
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
  std::vector<hipEvent_t> ev;   // 6 events per profiled step
  int prof_max = 0, prof_n = 0;
  bool force_lds_kernel = false;
};

static thread_local std::string g_create_err;

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
  if (c->solver != RG_SOLVER_ADMM) { err = "unsupported solver"; return RG_MPC_ERR_INVALID; }
  if (!(c->admm_rho > 0) || c->admm_iters < 1 || !(c->admm_relax > 0 && c->admm_relax < 2)) { err = "bad ADMM parameters"; return RG_MPC_ERR_INVALID; }
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
const char *rg_mpc_kernel_names(void) { return "rg_front_kernel,rg_qp_admm_reg_kernel,rg_qp_admm_kernel"; }

const char *rg_mpc_last_error(const rg_mpc_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int rg_mpc_create(const rg_mpc_config *cfg, int32_t batch, int32_t device, rg_mpc_handle **out) {
  if (!cfg || !out || batch < 1) { g_create_err = "null config/out or batch < 1"; return RG_MPC_ERR_INVALID; }
  *out = nullptr;
  rg_mpc_handle *h = new rg_mpc_handle();
  h->cfg = *cfg; h->B = batch; h->device = device;
  h->force_lds_kernel = (cfg->reserved0 & 1) != 0;  // bit0: use the LDS-resident QP kernel (A/B and generic-H path)
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
  AL(h->st.cmd, 3 * B); AL(h->st.rec, B * RG_REC_N); AL(h->st.bins, 5 * B); AL(h->st.counts, 8);
  AL(h->idx_dev, B); AL(h->t0_dev, B);
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
  HIPCHK(h, hipMemsetAsync(h->st.counts, 0, sizeof(int) * 8, s));
  hipEvent_t *pev = (h->prof_n < h->prof_max) ? &h->ev[(size_t)h->prof_n * 6] : nullptr;
  if (pev) HIPCHK(h, hipEventRecord(pev[0], s));
  hipLaunchKernelGGL(rg_front_kernel, dim3((B + 63) / 64), dim3(64), 0, s, h->dcfg, h->st, di, dout, t, B);
  HIPCHK(h, hipGetLastError());
  if (pev) HIPCHK(h, hipEventRecord(pev[1], s));
  for (int nc = 1; nc <= 4; nc++) {
    const int n = 3 * nc * H;
    if (!h->force_lds_kernel) {
      hipError_t lerr;
      if (launch_qp_reg_dispatch(nc, H, h->dcfg, h->st, dout, B, h->cu_count, s, &lerr)) {
        HIPCHK(h, lerr);
        if (pev) HIPCHK(h, hipEventRecord(pev[1 + nc], s));
        continue;
      }
    }
    size_t lds = h->lds_bytes[nc];
    if (lds == 0) { if (pev) HIPCHK(h, hipEventRecord(pev[1 + nc], s)); continue; } // TODO(round 2): out-of-LDS variant for H=20 with 3-4 stance legs
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    int grid = h->cu_count * per_cu;
    if (grid > B) grid = B;
    if (n <= 64) hipLaunchKernelGGL(rg_qp_admm_kernel<64>, dim3(grid), dim3(64), lds, s, h->dcfg, h->st, dout, nc, B);
    else if (n <= 128) hipLaunchKernelGGL(rg_qp_admm_kernel<128>, dim3(grid), dim3(128), lds, s, h->dcfg, h->st, dout, nc, B);
    else hipLaunchKernelGGL(rg_qp_admm_kernel<256>, dim3(grid), dim3(256), lds, s, h->dcfg, h->st, dout, nc, B);
    HIPCHK(h, hipGetLastError());
    if (pev) HIPCHK(h, hipEventRecord(pev[1 + nc], s));
  }
  if (pev) h->prof_n++;
  return RG_MPC_OK;
}

int rg_mpc_profile_begin(rg_mpc_handle *h, int32_t max_steps) {
  if (!h || max_steps < 1 || max_steps > 100000) { if (h) h->err = "profile_begin: bad max_steps"; return RG_MPC_ERR_INVALID; }
  HIPCHK(h, hipSetDevice(h->device));
  while ((int)h->ev.size() < max_steps * 6) {
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
    hipEvent_t *e = &h->ev[(size_t)k * 6];
    for (int j = 0; j < 5; j++) { float ms = 0; HIPCHK(h, hipEventElapsedTime(&ms, e[j], e[j + 1])); acc[j] += ms; }
    float ms = 0; HIPCHK(h, hipEventElapsedTime(&ms, e[0], e[5])); acc[5] += ms;
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

int rg_mpc_last_bin_counts(rg_mpc_handle *h, int32_t *out5, void *stream) {
  if (!h || !out5) return RG_MPC_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(h, hipMemcpy(out5, h->st.counts, sizeof(int) * 5, hipMemcpyDeviceToHost));
  return RG_MPC_OK;
}

} // extern "C"
