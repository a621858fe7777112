// tests/studies/sweep_bench.hip -- micro-benchmark and check of the QP bodies' matrix inversion, stand-alone:
//   variant 0: build_tile_kron6 + tile_sweep<8, 3, 60> + diagonal fix + permute_tile_rows_for_reduce   (both triangles)
//   variant 1: sym6_build_kron6 + sym6_sweep<10, 10, 64> + sym6_to_tile8                                (every block once)
// Each workgroup (one wave) inverts M = tabN (x) U + tabS (x) V + alpha I for its robot's 6 x 6 U, V (60 x 60, cond ~ 1e5),
// applies the result to a vector with tile8_matvec and writes the 60 entries; the host checks them against a dense CPU
// solve.  Timed alone (256 workgroups: one wave per CU) and under load (2 waves per SIMD through the LDS size, the product
// launch's shape) from in-kernel wall-clock stamps around the inversion.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I robot_gym_amd/csrc -o studies_bin/sweep_bench tests/studies/sweep_bench.hip
#include "rg_mpc_dev.h"
#include "../../include/rg_mpc.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "rg_qp_common.inc"
#include "rg_qp_tile_kernel.inc"
#include "rg_qp_wrench_kernel.inc"
#include "rg_qp_sym6.inc"
#include "sweep_variants.inc"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int H = 10, N = 60;

// tile_sweep without its LDS stores (ablation: wrong results, what is left is the cost of everything but the publish)
template <int T, int LG, int NREAL>
__device__ __forceinline__ void tile_sweep_nostore(double (&tile)[T][T], double *pbuf, const int lr, const int lc) {
  constexpr int TS = (T == 8) ? 10 : T, LC = 1 << LG, NPAD = TS * LC, NPB = NPAD + 2;
  for (int kb = 0; kb < LC; kb++) {
#pragma unroll
    for (int tr = 0; tr < T; tr++) {
      const int kp = kb * T + tr;
      if (kp >= NREAL) continue;
      double *pb = pbuf + (kp & 1) * NPB;
      const double invd = fast_rcp(pb[NPAD]);
      double prow[T], pcol[T];
#pragma unroll
      for (int t2 = 0; t2 < T; t2 += 2) {
        double2 a2 = *reinterpret_cast<const double2 *>(&pb[lr * TS + t2]);
        double2 b2 = *reinterpret_cast<const double2 *>(&pb[lc * TS + t2]);
        prow[t2] = a2.x; prow[t2 + 1] = a2.y; pcol[t2] = b2.x; pcol[t2 + 1] = b2.y;
      }
#pragma unroll
      for (int ta = 0; ta < T; ta++) {
        const double ncc = -prow[ta] * invd;
#pragma unroll
        for (int tb = 0; tb < T; tb++) tile[ta][tb] = fma(ncc, pcol[tb], tile[ta][tb]);
      }
#pragma unroll
      for (int ta = 0; ta < T; ta++) pin_row<T>(tile[ta]);
      __syncthreads();
    }
  }
}

// LDS map (doubles): tabN 100 | tabS 100 | U 36 | V 36 | vv 80 | pbuf 2 * 82 | stg (variant 0: EU | EV there)
// VAR 0: tile_sweep; 1..4: sym6 with exchange mode VAR - 1; 5: tile_sweep without stores
template <int VAR>
__global__ void __launch_bounds__(64, 2)
sweep_kernel(const double *__restrict__ UV, const double *__restrict__ tabs, const double *__restrict__ vin, double *__restrict__ out, unsigned *__restrict__ clk, const double alpha, const int reps) {
  double *tabN = smem, *tabS = tabN + 100, *U = tabS + 100, *V = U + 36, *vv = V + 36, *pbuf = vv + 80, *stg = pbuf /* the pivot buffers are dead when the blocks are staged */, *EU = pbuf + 164, *EV = EU + RG_E6_DOUBLES;
  const int b = blockIdx.x, tid = wg_lane<64>();
  for (int e = tid; e < 100; e += 64) { tabN[e] = tabs[e]; tabS[e] = tabs[100 + e]; }
  if (tid < 36) {
    const double u = UV[(size_t)b * 72 + tid], v = UV[(size_t)b * 72 + 36 + tid];
    U[tid] = u; V[tid] = v;
    if constexpr (VAR == 0 || VAR == 5) { put_periodic6(EU, tid / 6, tid % 6, u); put_periodic6(EV, tid / 6, tid % 6, v); }
  }
  for (int e = tid; e < 64; e += 64) vv[(e >> 3) * 10 + (e & 7)] = e < N ? vin[(size_t)b * 64 + e] : 0.0;
  __syncthreads();
  double tile[8][8];
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < reps; r++) {
    if constexpr (VAR == 0 || VAR == 5) {
      int lr = tid >> 3, lc = tid & 7;
      asm volatile("" : "+v"(lr), "+v"(lc));
      build_tile_kron6<H>(tile, tabN, tabS, EU, EV, lr, lc, N, alpha);
      if constexpr (VAR == 0) tile_sweep<8, 3, N>(tile, pbuf, lr, lc); else tile_sweep_nostore<8, 3, N>(tile, pbuf, lr, lc);
      if (lr == lc) {
#pragma unroll
        for (int ta = 0; ta < 8; ta++) tile[ta][ta] -= 2.0;
      }
      permute_tile_rows_for_reduce(tile, lc);
    } else {
      double A[6][6];
      int t2 = tid;
      asm volatile("" : "+v"(t2));
      int br, bc; bool on;
      if constexpr (VAR == 11) sym6_lane_quad<10>(t2, br, bc, on); else sym6_lane<10>(t2, br, bc, on);
      sym6_build_kron6<H>(A, tabN, tabS, U, V, br, bc, on, alpha);
      if constexpr (VAR == 15) sym6_sweep<10, 10, 64>(A, pbuf, br, bc, on); else if constexpr (VAR == 11) sym6_sweep_quad<10, 10>(A, pbuf, br, bc, on); else if constexpr (VAR >= 12) sym6_sweep_tn_abl<10, 10, 64, VAR - 8>(A, pbuf, br, bc, on); else if constexpr (VAR >= 7) sym6_sweep_tn_abl<10, 10, 64, VAR - 7>(A, pbuf, br, bc, on); else if constexpr (VAR == 6) sym6_sweep2<10, 10, 64>(A, pbuf, br, bc, on); else sym6_sweep_mode<10, 10, 64, VAR - 1>(A, pbuf, br, bc, on);
      sym6_to_tile8<10, 10, 3, 64>(A, br, bc, on, tile, stg, t2);
    }
    __syncthreads();
  }
  const unsigned long long t1 = wall_clock64();
  const int lr = tid >> 3, lc = tid & 7;
  const double t = tile8_matvec<3>(tile, vv, lr, lc);   // = (M^-1 v)_io on the owner lanes
  const int io = lr * 8 + TileShape<8>::own_a(lc);
  if (io < 64) out[(size_t)b * 64 + io] = t;
  if (tid == 0) { clk[2 * b] = (unsigned)(t0 & 0xFFFFFFFFu); clk[2 * b + 1] = (unsigned)((t1 - t0) & 0xFFFFFFFFu); }
}

static void cpu_solve(const double *M, const double *v, double *x, int n) {   // Gaussian elimination with partial pivoting (long double)
  std::vector<long double> a((size_t)n * (n + 1));
  for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) a[(size_t)i * (n + 1) + j] = M[(size_t)i * n + j]; a[(size_t)i * (n + 1) + n] = v[i]; }
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int i = k + 1; i < n; i++) if (fabsl(a[(size_t)i * (n + 1) + k]) > fabsl(a[(size_t)p * (n + 1) + k])) p = i;
    if (p != k) for (int j = 0; j <= n; j++) std::swap(a[(size_t)k * (n + 1) + j], a[(size_t)p * (n + 1) + j]);
    for (int i = k + 1; i < n; i++) {
      const long double f = a[(size_t)i * (n + 1) + k] / a[(size_t)k * (n + 1) + k];
      for (int j = k; j <= n; j++) a[(size_t)i * (n + 1) + j] -= f * a[(size_t)k * (n + 1) + j];
    }
  }
  for (int i = n - 1; i >= 0; i--) {
    long double s = a[(size_t)i * (n + 1) + n];
    for (int j = i + 1; j < n; j++) s -= a[(size_t)i * (n + 1) + j] * x[j];
    x[i] = (double)(s / a[(size_t)i * (n + 1) + i]);
  }
}

template <int VAR>
static void run(const char *name, int grid, int reps, size_t lds, const double *dUV, const double *dtabs, const double *dv, double *dout, unsigned *dclk,
                const std::vector<double> &UV, const std::vector<double> &tabs, const std::vector<double> &vin, double alpha, int ncheck) {
  CHECK(hipFuncSetAttribute((const void *)sweep_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  sweep_kernel<VAR><<<grid, 64, lds>>>(dUV, dtabs, dv, dout, dclk, alpha, reps);   // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int launches = 5;
  for (int l = 0; l < launches; l++) sweep_kernel<VAR><<<grid, 64, lds>>>(dUV, dtabs, dv, dout, dclk, alpha, reps);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> clk(2 * (size_t)grid);
  std::vector<double> out((size_t)grid * 64);
  CHECK(hipMemcpy(clk.data(), dclk, clk.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(out.data(), dout, out.size() * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<double> durs(grid);
  for (int b = 0; b < grid; b++) durs[b] = clk[2 * b + 1] * 0.01 / reps;   // 100 MHz wall clock -> us per inversion
  std::sort(durs.begin(), durs.end());
  double mean = 0.0;
  for (double d : durs) mean += d;
  mean /= grid;
  // check
  double worst = 0.0;
  for (int b = 0; b < ncheck && b < grid; b++) {
    std::vector<double> M((size_t)N * N), x(N);
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) {
        const int a = i / 6, ia = i % 6, bb = j / 6, ja = j % 6;
        M[(size_t)i * N + j] = tabs[a * H + bb] * UV[(size_t)b * 72 + ia * 6 + ja] + tabs[100 + a * H + bb] * UV[(size_t)b * 72 + 36 + ia * 6 + ja] + (i == j ? alpha : 0.0);
      }
    cpu_solve(M.data(), &vin[(size_t)b * 64], x.data(), N);
    double nx = 0.0, ne = 0.0;
    for (int i = 0; i < N; i++) { nx = fmax(nx, fabs(x[i])); ne = fmax(ne, fabs(x[i] - out[(size_t)b * 64 + i])); }
    worst = fmax(worst, ne / nx);
  }
  printf("%-34s grid %6d reps %d: launch %8.1f us | per inversion: mean %6.2f p50 %6.2f p99 %6.2f max %6.2f us | worst rel err vs CPU %.2e\n", name, grid, reps,
         ms * 1000.0 / launches, mean, durs[grid / 2], durs[(size_t)(grid * 0.99)], durs[grid - 1], worst);
}

int main(int argc, char **argv) {
  const int maxgrid = 8192;
  const double alpha = 1e-5;
  std::vector<double> tabs(200), UV((size_t)maxgrid * 72), vin((size_t)maxgrid * 64);
  for (int a = 0; a < H; a++)
    for (int b = 0; b < H; b++) {
      const int m = a > b ? a : b;
      double s = 0.0;
      for (int k = m + 1; k <= H; k++) s += (k - a - 0.5) * (k - b - 0.5);
      tabs[a * H + b] = 2.0 * (H - m);
      tabs[100 + a * H + b] = 2.0 * s;
    }
  srand(1);
  auto rnd = []() { return rand() / (double)RAND_MAX * 2.0 - 1.0; };
  for (int b = 0; b < maxgrid; b++) {
    // U = R R' dt^2-ish, V = Q Q' dt^4-ish (SPD, scaled like G_U, G_V of a trot robot)
    double R[36], Q[36];
    for (int e = 0; e < 36; e++) { R[e] = rnd(); Q[e] = rnd(); }
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double u = 0.0, v = 0.0;
        for (int k = 0; k < 6; k++) { u += R[i * 6 + k] * R[j * 6 + k]; v += Q[i * 6 + k] * Q[j * 6 + k]; }
        UV[(size_t)b * 72 + i * 6 + j] = u * 6.25e-4 * 0.05;
        UV[(size_t)b * 72 + 36 + i * 6 + j] = v * 3.9e-7 * 5.0;
      }
    for (int e = 0; e < 64; e++) vin[(size_t)b * 64 + e] = rnd();
  }
  double *dUV, *dtabs, *dv, *dout; unsigned *dclk;
  CHECK(hipMalloc(&dUV, UV.size() * 8)); CHECK(hipMalloc(&dtabs, tabs.size() * 8)); CHECK(hipMalloc(&dv, vin.size() * 8));
  CHECK(hipMalloc(&dout, (size_t)maxgrid * 64 * 8)); CHECK(hipMalloc(&dclk, (size_t)maxgrid * 2 * 4));
  CHECK(hipMemcpy(dUV, UV.data(), UV.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dtabs, tabs.data(), tabs.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dv, vin.data(), vin.size() * 8, hipMemcpyHostToDevice));
  const size_t lds_load = 20 * 1024 - 256;   // 8 workgroups per CU: two waves per SIMD, the product launch's shape
  const size_t lds_alone = 80 * 1024;        // one or two workgroups per CU
  CHECK(hipMemset(dout, 0, (size_t)maxgrid * 64 * 8));
#define RUNALL(tag, grid, reps, lds) \
  run<0>("tile_sweep " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64); \
  run<1>("sym6 lds " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64); \
  run<2>("sym6 lds-select " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64); \
  run<3>("sym6 bpermute " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64); \
  run<4>("sym6 NOSTORE(abl) " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 0); \
  run<5>("tile_sweep NOSTORE(abl) " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 0); \
  run<6>("sym6 two-pivot " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64); \
  run<7>("sym6 turn-over " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64); \
  run<8>("turn-over ABL read-other-buf " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 0); \
  run<9>("turn-over ABL one-store " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 0); \
  run<10>("turn-over b64 stores " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64); \
  run<11>("sym6 quad-spread " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64); \
  run<12>("turn-over ABL no LDS " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 0); \
  run<13>("turn-over ABL reads only " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 0); \
  run<14>("turn-over ABL stores only " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 0); \
  run<15>("sym6_sweep (shipped) " tag, grid, reps, lds, dUV, dtabs, dv, dout, dclk, UV, tabs, vin, alpha, 64);
  for (int pass = 0; pass < 2; pass++) {
    RUNALL("alone", 256, 4, lds_alone)
    RUNALL("1 round", 2048, 4, lds_load)
    RUNALL("4 rounds", 8192, 1, lds_load)
  }
  return 0;
}
