// Study (GPU): rate of v_mfma_f64_16x16x4_f64 on gfx950 and whether it overlaps f64 VALU work of the same wave / of the other
// wave on the SIMD.  hipcc -O3 --offload-arch=gfx950 tests/studies/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NF>
__global__ __launch_bounds__(64, 2) void k(double *out, int iters, int role_split) {
  __shared__ double pad[2400];   // ~19 KB: eight workgroups per CU = two waves per SIMD
  const int lane = threadIdx.x;
  pad[lane] = lane;
  constexpr int NA = NM > 0 ? 16 : 1;
  d4 acc[NA];
  for (int i = 0; i < NA; i++) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  constexpr int NFR = NF > 0 ? 32 : 1;   // (no FMA registers in the MFMA-only kernel: it must fit two waves per SIMD)
  double f[NFR];
  for (int i = 0; i < NFR; i++) f[i] = 1.0 + i * 1e-3 + lane * 1e-6;
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  const bool do_m = role_split == 0 || (blockIdx.x & 1) == 0, do_f = role_split == 0 || (blockIdx.x & 1) == 1;
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    if (do_m) {
#pragma unroll
      for (int i = 0; i < NM; i++) acc[i % NA] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % NA], 0, 0, 0);
    }
    if (do_f) {
#pragma unroll
      for (int i = 0; i < NF; i++) f[i % NFR] = fma(f[i % NFR], a, b);
    }
  }
  long long t1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < NA; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < NFR; i++) s += f[i];
  out[blockIdx.x * 64 + lane] = s + pad[(lane * 7) & 63] + (double)(t1 - t0);
  if (lane == 0) out[blockIdx.x * 64] = (double)(t1 - t0);
}

template <int NM, int NF>
void run(const char *what, int grid, int role_split) {
  double *out;
  hipMalloc(&out, sizeof(double) * grid * 64);
  const int iters = 2000;
  hipLaunchKernelGGL((k<NM, NF>), dim3(grid), dim3(64), 0, 0, out, 10, role_split);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NF>), dim3(grid), dim3(64), 0, 0, out, iters, role_split);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double *h = (double *)malloc(sizeof(double) * grid * 64);
  hipMemcpy(h, out, sizeof(double) * grid * 64, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int b = 0; b < grid; b++) mean += h[b * 64];
  mean /= grid;
  // wall_clock64: 100 MHz
  printf("%-44s grid %5d  kernel %.3f ms  per wave-iteration %.1f ns (in-kernel clock), %d MFMA + %d FMA per iteration\n", what, grid, ms, mean * 10.0 / iters, NM, NF);
  free(h); hipFree(out);
}

int main() {
  // one wave per SIMD (4 workgroups per CU), then two
  for (int wpc : {1, 4, 8}) {
    const int grid = 256 * wpc;
    printf("== %d workgroups (waves) per CU\n", wpc);
    run<16, 0>("16 MFMA", grid, 0);
    run<0, 128>("128 FMA", grid, 0);
    run<0, 256>("256 FMA", grid, 0);
    run<16, 128>("16 MFMA + 128 FMA, same wave", grid, 0);
    run<16, 256>("16 MFMA + 256 FMA, same wave", grid, 0);
    run<16, 256>("16 MFMA | 256 FMA, alternate workgroups", grid, 1);
    run<16, 64>("16 MFMA + 64 FMA, same wave", grid, 0);
  }
  return 0;
}
