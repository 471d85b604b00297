// micro-benchmark: throughput of ocml j0 / j1 (FP64) by argument range (diagnostics)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void kj(const double* x, double* y, int n, int reps, double scale, int which) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = 0.0, xv = x[i] * scale;
  for (int r = 0; r < reps; ++r) {
    a += which ? j1(xv) : j0(xv);
    xv += 1e-3;
  }
  y[i] = a;
}
int main() {
  const int n = 1 << 20, reps = 16;
  double *x, *y;
  hipMalloc(&x, n * 8); hipMalloc(&y, n * 8);
  double* h = (double*)malloc(n * 8);
  for (int i = 0; i < n; ++i) h[i] = (i + 0.5) / n;
  hipMemcpy(x, h, n * 8, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int which = 0; which < 2; ++which)
    for (double scale : {1.0, 8.0, 50.0, 200.0, 800.0, 5000.0, 1e5}) {
      kj<<<n / 256, 256>>>(x, y, n, reps, scale, which);
      hipDeviceSynchronize();
      hipEventRecord(a);
      kj<<<n / 256, 256>>>(x, y, n, reps, scale, which);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("j%d x<%g: %.1f us for %d evals -> %.1f G evals/s\n", which, scale, ms * 1e3, n * reps, n * (double)reps / ms / 1e6);
    }
  return 0;
}
