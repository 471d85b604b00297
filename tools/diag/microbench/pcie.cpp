// PCIe copy microbenchmark: pinned H2D 8 MB / D2H 16 MB, whole and in 1 MB chunks; host memcpy with T threads.
#include <hip/hip_runtime.h>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t MB = 1 << 20;
  char *h, *d; hipStream_t st;
  hipHostMalloc((void**)&h, 32 * MB, hipHostMallocDefault); hipMalloc((void**)&d, 32 * MB); hipStreamCreate(&st);
  std::vector<char> pg(32 * MB, 1), pg2(32 * MB, 2);
  memset(h, 1, 32 * MB);
  for (int rep = 0; rep < 3; ++rep) {
    for (size_t sz : {8 * MB, 16 * MB}) {
      double t0 = now(); hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); double t1 = now();
      hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); double t2 = now();
      printf("whole %2zu MB: H2D %.1f us (%.1f GB/s)  D2H %.1f us (%.1f GB/s)\n", sz / MB, t1 - t0, sz / (t1 - t0) / 1e3, t2 - t1, sz / (t2 - t1) / 1e3);
      t0 = now(); for (size_t o = 0; o < sz; o += MB) hipMemcpyAsync(d + o, h + o, MB, hipMemcpyHostToDevice, st); double ti = now(); hipStreamSynchronize(st); t1 = now();
      for (size_t o = 0; o < sz; o += MB) hipMemcpyAsync(h + o, d + o, MB, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); t2 = now();
      printf("1MB chunks %2zu MB: H2D %.1f us (issue %.1f)  D2H %.1f us\n", sz / MB, t1 - t0, ti - t0, t2 - t1);
    }
    for (int T : {1, 2, 4, 8, 16}) {
      double t0 = now();
#pragma omp parallel num_threads(T)
      { int q = omp_get_thread_num(), nt = omp_get_num_threads(); size_t per = 16 * MB / nt; memcpy(h + q * per, pg.data() + q * per, per); }
      double t1 = now();
#pragma omp parallel num_threads(T)
      { int q = omp_get_thread_num(), nt = omp_get_num_threads(); size_t per = 16 * MB / nt; memcpy(pg2.data() + q * per, h + q * per, per); }
      double t2 = now();
      printf("memcpy 16 MB, %2d threads: to pinned %.1f us (%.1f GB/s), from pinned %.1f us (%.1f GB/s)\n", T, t1 - t0, 16 * MB / (t1 - t0) / 1e3, t2 - t1, 16 * MB / (t2 - t1) / 1e3);
    }
  }
  // event record / query overhead
  hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
  double t0 = now(); for (int i = 0; i < 100; ++i) hipEventRecord(e, st); double t1 = now(); hipStreamSynchronize(st);
  double t2 = now(); for (int i = 0; i < 100; ++i) hipEventQuery(e); double t3 = now();
  printf("hipEventRecord %.2f us, hipEventQuery %.2f us per call\n", (t1 - t0) / 100, (t3 - t2) / 100);
  hipPointerAttribute_t at; t0 = now(); for (int i = 0; i < 100; ++i) { if (hipPointerGetAttributes(&at, pg.data() + i) != hipSuccess) (void)hipGetLastError(); } t1 = now();
  for (int i = 0; i < 100; ++i) hipPointerGetAttributes(&at, h + i); t2 = now();
  printf("hipPointerGetAttributes pageable %.2f us, pinned %.2f us\n", (t1 - t0) / 100, (t2 - t1) / 100);
  return 0;
}
