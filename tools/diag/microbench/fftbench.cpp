// micro-benchmark: rocFFT c2c inverse, length n, batch b, in-place vs out-of-place (diagnostics)
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <vector>
static double run(size_t n, size_t batch, bool inplace, rocfft_array_type = rocfft_array_type_complex_interleaved) {
  rocfft_plan plan = nullptr;
  size_t len[1] = {n};
  rocfft_plan_create(&plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace,
                     rocfft_transform_type_complex_inverse, rocfft_precision_double, 1, len, batch, nullptr);
  size_t wb = 0;
  rocfft_plan_get_work_buffer_size(plan, &wb);
  rocfft_execution_info info;
  rocfft_execution_info_create(&info);
  void* work = nullptr;
  if (wb) { hipMalloc(&work, wb); rocfft_execution_info_set_work_buffer(info, work, wb); }
  void *a, *b;
  hipMalloc(&a, n * batch * 16); hipMalloc(&b, n * batch * 16);
  hipMemset(a, 0, n * batch * 16);
  hipStream_t st; hipStreamCreate(&st);
  rocfft_execution_info_set_stream(info, st);
  void* in[1] = {a}; void* out[1] = {b};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) rocfft_execute(plan, in, inplace ? nullptr : out, info);
  hipStreamSynchronize(st);
  const int reps = 200;
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) rocfft_execute(plan, in, inplace ? nullptr : out, info);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("n=%zu batch=%zu %s work=%zu KB: %.2f us per transform call\n", n, batch, inplace ? "in-place " : "out-place", wb / 1024, ms * 1e3 / reps);
  return ms;
}
int main() {
  rocfft_setup();
  for (size_t n : {131072ul, 65536ul, 8192ul}) {
    run(n, 2, true); run(n, 2, false); run(n, 4, true); run(n, 4, false);
  }
  run(131072, 20, true); run(131072, 20, false);
  return 0;
}
