// Stand-alone reproducer attempt for the round-2 diagnosis "__syncthreads_or gave one wave of a 256-lane workgroup a
// different vote result than the others once the launch's timing shifted" (rg_qp_common.inc, workgroup_any).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/sor tests/studies/syncthreads_or_repro.hip && /tmp/sor
// Each 256-lane workgroup votes in a loop the way the ADMM bodies do: per vote one lane (a different wave each time) says
// "still moving" until its own stopping iteration; every wave records the vote results it saw and the iteration at which it
// left the loop.  A correct workgroup reduction gives every wave the same record.  Between repetitions a second kernel with
// a large unrolled body evicts the instruction cache, every other repetition a third kernel leaves all LDS full of a non-zero
// pattern, and the workgroups do a data-dependent amount of LDS / VALU work between votes so that their waves arrive at the
// vote at different times.  Variant 0: the device library's
// __syncthreads_or.  Variant 1: ballot + LDS flags + two barriers (what the library uses now).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void evict_icache(double *out, int n) {   // ~70 KB of straight-line code
  double a = threadIdx.x * 1e-3, b = 1.0;
#pragma unroll 1
  for (int r = 0; r < n; r++) {
#pragma unroll
    for (int k = 0; k < 4096; k++) { a = fma(a, 1.0000001, b); b = fma(b, 0.9999999, a * 1e-9 + k); }
  }
  if (a == 12345.678) out[0] = a + b;
}

// LDS keeps its contents from kernel to kernel: leave every CU's LDS full of a non-zero pattern before the vote kernel, the
// way another QP kernel (or the test suite's poison kernel) does -- a workgroup reduction that expected its LDS accumulator to
// start at zero would now start from garbage.
extern __shared__ unsigned dyn_lds[];
__global__ void fill_lds(unsigned pattern, int nwords) {
  for (int e = threadIdx.x; e < nwords; e += blockDim.x) dyn_lds[e] = pattern;
  __syncthreads();
  if (dyn_lds[(threadIdx.x * 7) % nwords] == 1u) dyn_lds[0] = 2u;
}

__shared__ int slots[16];
template <int VARIANT>
__device__ __forceinline__ bool vote(int pred) {
  if (VARIANT == 0) return __syncthreads_or(pred) != 0;
  const int any = __ballot(pred) != 0ull;
  if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = any;
  __syncthreads();
  const int r = slots[0] | slots[1] | slots[2] | slots[3];
  __syncthreads();
  return r != 0;
}

template <int VARIANT>
__global__ void __launch_bounds__(256) vote_loop(int *exit_it, unsigned long long *seen, int max_votes, unsigned seed) {
  __shared__ double work[256];
  const int tid = threadIdx.x, wave = tid >> 6, wg = blockIdx.x;
  unsigned h = (wg * 2654435761u) ^ seed;
  const int stop_at = 3 + (h >> 8) % (max_votes - 4);          // the workgroup's true stopping vote
  unsigned long long rec = 0;
  int it = 0;
  double x = tid * 0.001;
  for (; it < max_votes; it++) {
    // uneven work per wave before the vote (a different wave is the slow one each time)
    const int spin = ((it + wave + (h & 3)) & 3) * 40 + 8;
    for (int k = 0; k < spin; k++) { work[tid] = x; x = fma(x, 1.0000001, work[(tid + 17) & 255] * 1e-9); }
    const int mover = (it * 37 + (h >> 4)) & 255;               // the one lane that still moves at this vote
    const int pred = (tid == mover) && (it < stop_at);
    const bool any = vote<VARIANT>(pred);
    rec |= (unsigned long long)(any ? 1 : 0) << (it & 63);
    if (!any) break;
  }
  if ((tid & 63) == 0) { exit_it[wg * 4 + wave] = it; seen[wg * 4 + wave] = rec; }
  if (x == 42.0) exit_it[0] = -1;
}

template <int VARIANT>
static long run(int reps, int wgs) {
  int *d_exit; unsigned long long *d_seen; double *d_out;
  hipMalloc(&d_exit, wgs * 4 * sizeof(int)); hipMalloc(&d_seen, wgs * 4 * sizeof(unsigned long long)); hipMalloc(&d_out, 8);
  std::vector<int> ex(wgs * 4); std::vector<unsigned long long> sn(wgs * 4);
  long bad = 0;
  for (int r = 0; r < reps; r++) {
    evict_icache<<<1024, 64>>>(d_out, 2);
    if (r & 1) fill_lds<<<2048, 256, 64 * 1024>>>((r & 2) ? 0xFFFFFFFFu : 0x7ff8deadu, 16 * 1024);   // odd repetitions: dirty LDS first
    vote_loop<VARIANT><<<wgs, 256>>>(d_exit, d_seen, 48, 1234567u * (r + 1));
    hipMemcpy(ex.data(), d_exit, ex.size() * sizeof(int), hipMemcpyDeviceToHost);
    hipMemcpy(sn.data(), d_seen, sn.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    for (int w = 0; w < wgs; w++)
      for (int k = 1; k < 4; k++)
        if (ex[w * 4 + k] != ex[w * 4] || sn[w * 4 + k] != sn[w * 4]) {
          if (bad < 5) printf("  variant %d rep %d workgroup %d: wave 0 left at vote %d (saw %llx), wave %d at vote %d (saw %llx)\n", VARIANT, r, w, ex[w * 4], sn[w * 4], k, ex[w * 4 + k], sn[w * 4 + k]);
          bad++;
        }
  }
  hipFree(d_exit); hipFree(d_seen); hipFree(d_out);
  return bad;
}

int main() {
  const int reps = 200, wgs = 4096;
  const long b0 = run<0>(reps, wgs), b1 = run<1>(reps, wgs);
  printf("__syncthreads_or       : %ld wave disagreements in %d launches x %d workgroups\n", b0, reps, wgs);
  printf("ballot + LDS + barriers: %ld wave disagreements in %d launches x %d workgroups\n", b1, reps, wgs);
  return 0;
}
