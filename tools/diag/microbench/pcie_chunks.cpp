// D2H of 16 MB from pinned staging in halving chunks: one stream against chunks alternating between two streams
// (does the second stream hide the ~9 us of fixed cost per copy?).  build: hipcc -O2 pcie_chunks.cpp -o pcie_chunks
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t MB = 1 << 20, KB = 1 << 10;
  char *h, *d; hipStream_t s0, s1;
  hipHostMalloc((void**)&h, 32 * MB, hipHostMallocDefault); hipMalloc((void**)&d, 32 * MB);
  hipStreamCreate(&s0); hipStreamCreate(&s1);
  std::vector<std::vector<size_t>> plans = {
      {16 * MB}, {8 * MB, 4 * MB, 2 * MB, 2 * MB}, {8 * MB, 4 * MB, 2 * MB, 1 * MB, 512 * KB, 512 * KB},
      {4 * MB, 4 * MB, 4 * MB, 2 * MB, 1 * MB, 512 * KB, 256 * KB, 256 * KB}, {2 * MB, 2 * MB, 2 * MB, 2 * MB, 2 * MB, 2 * MB, 2 * MB, 1 * MB, 512 * KB, 256 * KB, 256 * KB}};
  for (int rep = 0; rep < 3; ++rep)
    for (auto& p : plans)
      for (int two = 0; two < 2; ++two) {
        hipDeviceSynchronize();
        double t0 = now();
        size_t off = 0; int c = 0;
        for (size_t len : p) { hipMemcpyAsync(h + off, d + off, len, hipMemcpyDeviceToHost, (two && (c & 1)) ? s1 : s0); off += len; ++c; }
        double ti = now();
        hipStreamSynchronize(s0); if (two) hipStreamSynchronize(s1);
        double t1 = now();
        printf("chunks %2zu (last %4zu KB) %s: %.1f us (issue %.1f)\n", p.size(), p.back() / KB, two ? "two streams" : "one stream ", t1 - t0, ti - t0);
      }
  return 0;
}
