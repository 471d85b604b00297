// gridbar.hip -- what does a device-wide barrier cost on MI355X (8 XCDs)?  Sizing experiment for a persistent,
// cooperative small-rule march: G resident workgroups, N barriers of the sense-free counter type
// (one atomic add per workgroup, spin on the counter with a bounded loop: a wave that gives up sets an abort flag
// that every other spin loop checks, so the grid always drains).  Prints us per barrier, with and without a
// 64 KB write + __threadfence() per workgroup in front of it (the data a real phase would publish).
//   hipcc --offload-arch=gfx950 -O3 gridbar.hip -o gridbar && ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned int* counter, unsigned int* abort_flag, unsigned int target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();  // release: this workgroup's writes before its arrival
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();  // acquire
  }
  ok = __syncthreads_and(ok);
  return ok;
}

__global__ void __launch_bounds__(256) k_bar(unsigned int* counter, unsigned int* abort_flag, int nbar, double* buf, int work) {
  const unsigned G = gridDim.x;
  for (int b = 0; b < nbar; ++b) {
    if (work) {  // 64 KB per workgroup, written and (after the barrier) read by the neighbour
      double* mine = buf + (size_t)blockIdx.x * 8192;
      for (int q = threadIdx.x; q < 8192; q += 256) mine[q] = (double)(b + q);
    }
    if (!grid_barrier(counter, abort_flag, (unsigned)(b + 1) * G)) return;
    if (work) {
      const double* other = buf + (size_t)((blockIdx.x + 1) % G) * 8192;
      double s = 0;
      for (int q = threadIdx.x; q < 8192; q += 256) s += other[q];
      if (s == -1.0) buf[0] = s;
    }
  }
}

int main() {
  unsigned int* d;
  double* buf;
  hipMalloc(&d, 64);
  hipMalloc(&buf, sizeof(double) * 8192 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int work = 0; work < 2; ++work)
    for (unsigned G : {16u, 64u, 128u, 256u, 512u}) {
      for (int nbar : {1, 200}) {
        hipMemset(d, 0, 64);
        void* args[] = {&d, nullptr, &nbar, &buf, &work};
        unsigned int* ab = d + 8;
        args[1] = &ab;
        hipEventRecord(a, 0);
        hipError_t e = hipLaunchCooperativeKernel((const void*)k_bar, dim3(G), dim3(256), args, 0, 0);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned int h[16];
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("work %d  G %4u  barriers %4d: %8.2f us total%s%s\n", work, G, nbar, 1e3 * ms, e == hipSuccess ? "" : "  LAUNCH FAILED",
               h[8] ? "  ABORTED" : "");
      }
    }
  return 0;
}
