// PCIe by kernels instead of the SDMA engines: a copy kernel reading pinned host memory (H2D) / writing it (D2H),
// against hipMemcpyAsync on the same buffers.  build: hipcc --offload-arch=gfx950 -O3 pcie_kernel.hip -o pcie_kernel
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  const size_t MB = 1 << 20;
  char *h, *d; hipStream_t st;
  hipHostMalloc((void**)&h, 32 * MB, hipHostMallocDefault); hipMalloc((void**)&d, 32 * MB); hipStreamCreate(&st);
  char* hd = nullptr; hipHostGetDevicePointer((void**)&hd, h, 0);
  memset(h, 1, 32 * MB);
  for (int rep = 0; rep < 3; ++rep)
    for (size_t sz : {8 * MB, 16 * MB}) {
      double t0 = now(); hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); double t1 = now();
      hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); double t2 = now();
      printf("memcpy %2zu MB: H2D %.1f us (%.1f GB/s)  D2H %.1f us (%.1f GB/s)\n", sz / MB, t1 - t0, sz / (t1 - t0) / 1e3, t2 - t1, sz / (t2 - t1) / 1e3);
      for (int grid : {64, 256, 1024, 4096}) {
        t0 = now(); hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st, (const double2*)hd, (double2*)d, sz / 16); hipStreamSynchronize(st); t1 = now();
        hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st, (const double2*)d, (double2*)hd, sz / 16); hipStreamSynchronize(st); t2 = now();
        printf("kernel %2zu MB grid %4d: H2D %.1f us (%.1f GB/s)  D2H %.1f us (%.1f GB/s)\n", sz / MB, grid, t1 - t0, sz / (t1 - t0) / 1e3, t2 - t1, sz / (t2 - t1) / 1e3);
      }
    }
  return 0;
}
