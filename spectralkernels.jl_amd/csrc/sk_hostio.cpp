// sk_hostio.cpp -- transfers between caller-owned host arrays and HBM for the host-array entries
// (sk_kernel_values & co: src/adaptive.jl:95-108 takes and returns plain host Vectors).
//
// A pageable hipMemcpy of the 24 MB of BASELINE config 2 (8 MB of lags in, 16 MB of values and error estimates
// out) costs ~3.5 ms on this stack (single-threaded staging inside the runtime), against 0.12 ms for the march
// itself.  Here every DMA runs from / to pinned staging buffers owned by the config, in chunks:
//   (a few chunks of growing / shrinking size, see make_chunks)
//   in:  T host threads copy chunk c into the pinned buffer (each its own slice) while the DMA engine moves chunk
//        c-1; the same pass checks that the lags are valid and strictly increasing (src/adaptive.jl:99,113), so the
//        march needs no speculative start on the device;
//   out: all chunk DMAs are enqueued at once, each followed by an event; the threads drain chunk c into the
//        caller's arrays while the engine moves chunk c+1.
// Arrays that already live in pinned memory (sk_host_alloc) are copied directly, without staging.
// No pageable pointer is ever handed to an asynchronous HIP copy.
#include <omp.h>
#include <immintrin.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "sk_config.h"

namespace sk {

static const size_t kChunk = (size_t)1 << 17;       // smallest DMA chunk, doubles (1 MB)
static const size_t kSerialMax = (size_t)1 << 14;   // below this many doubles one thread does the staging

static int copy_threads() {
  static int t = [] {
    int v = sk::host_threads(8);  // no more than the process may run at once (affinity, cgroup quota)
    if (const char* e = getenv("SK_COPY_THREADS")) v = atoi(e);
    return v < 1 ? 1 : (v > 32 ? 32 : v);
  }();
  return t;
}

// Is p inside a pinned (hipHostMalloc / hipHostRegister) allocation?
static bool is_pinned(const void* p) {
  hipPointerAttribute_t at;
  std::memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // plain pageable memory: clear the sticky error
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

static int ensure_stage(double** buf, size_t* have, size_t doubles) {
  if (doubles * sizeof(double) <= *have) return SK_OK;
  if (*buf) HIPCHK(hipHostFree(*buf));
  *buf = nullptr;
  *have = 0;
  const size_t bytes = doubles * sizeof(double) + doubles * sizeof(double) / 4;
  HIPCHK(hipHostMalloc((void**)buf, bytes, hipHostMallocDefault));
  *have = bytes;
  return SK_OK;
}

// Chunk boundaries of [0, total): sizes halve from total/2 down to kChunk (download: the last chunk, whose drain
// is exposed, is small; every DMA costs ~9 us of fixed overhead on this stack -- 16 MB in 1 MB chunks took 450 us
// against 305 us in one piece -- so there are only log2 many); `growing` reverses the order (upload: the first
// chunk, whose staging copy is exposed, is small).
static void make_chunks(size_t total, bool growing, std::vector<size_t>& edge) {
  std::vector<size_t> len;
  size_t rem = total;
  while (rem > 2 * kChunk) {
    size_t c = ((rem / 2 + kChunk - 1) / kChunk) * kChunk;
    len.push_back(c);
    rem -= c;
  }
  if (rem) len.push_back(rem);
  if (growing) std::reverse(len.begin(), len.end());
  edge.assign(1, 0);
  for (size_t c : len) edge.push_back(edge.back() + c);
}

// slice q of nt of [lo, hi)
static inline void slice_of(size_t lo, size_t hi, int q, int nt, size_t& a, size_t& b) {
  const size_t len = hi - lo, per = (len + nt - 1) / nt;
  a = lo + (size_t)q * per;
  b = a + per;
  if (a > hi) a = hi;
  if (b > hi) b = hi;
}

// order flags of src[a, b): bit 0 = not strictly increasing (incl. the step from src[a-1]), bit 1 = invalid lag
static inline int scan_order(const double* src, size_t a, size_t b) {
  int bad = 0, uns = 0;
  if (a > 0 && a < b) uns |= !(src[a - 1] < src[a]);
  for (size_t i = a; i + 1 < b; ++i) uns |= !(src[i] < src[i + 1]);
  for (size_t i = a; i < b; ++i) bad |= !(src[i] >= 0.0);
  return uns | (bad << 1);
}

// The chunked staging pass of h2d_doubles: `nthreads` host threads copy chunk c of src into `stage` (each its own slice)
// and look at the order of their slice; thread 0 hands every complete chunk [lo, hi) to `on_chunk` (the DMA).
// The order flags are taken from the CALLER's array -- the step across a slice boundary reads src[a-1], never the
// neighbouring thread's (possibly unfinished) slice of the staging buffer.  Returns the OR of scan_order's flags.
template <class OnChunk>
static int stage_chunks(double* stage, const double* src, const std::vector<size_t>& edge,
                        std::vector<std::atomic<int>>& chunk_done, bool want_order, int nthreads, OnChunk&& on_chunk) {
  const size_t nch = edge.size() - 1;
  std::atomic<int> flags{0};
#pragma omp parallel num_threads(nthreads)
  {
    const int nt = omp_get_num_threads(), q = omp_get_thread_num();
    int f = 0;
    for (size_t c = 0; c < nch; ++c) {
      const size_t lo = edge[c], hi = edge[c + 1];
      size_t a, b;
      slice_of(lo, hi, q, nt, a, b);
      if (b > a) {
        std::memcpy(stage + a, src + a, (b - a) * sizeof(double));
        if (want_order) f |= scan_order(src, a, b);
      }
      chunk_done[c].fetch_add(1, std::memory_order_release);
      if (q == 0) {
        while (chunk_done[c].load(std::memory_order_acquire) < nt) _mm_pause();
        on_chunk(lo, hi);
      }
    }
    if (f) flags.fetch_or(f);
  }
  return flags.load();
}

int h2d_doubles(sk_config* cfg, double* d_dst, const double* src, size_t cnt, hipStream_t st, int* order) {
  if (order) *order = SK_ORDER_UNKNOWN;
  if (cnt == 0) return SK_OK;
  if (is_pinned(src)) {
    HIPCHK(hipMemcpyAsync(d_dst, src, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    return SK_OK;
  }
  int rc = ensure_stage(&cfg->h_in, &cfg->hin_bytes, cnt);
  if (rc) return rc;
  double* stage = cfg->h_in;
  if (cnt <= kSerialMax) {
    std::memcpy(stage, src, cnt * sizeof(double));
    if (order) {
      const int f = scan_order(src, 0, cnt);
      *order = (f & 2) ? SK_ORDER_INVALID : ((f & 1) ? SK_ORDER_UNSORTED : SK_ORDER_SORTED);
    }
    HIPCHK(hipMemcpyAsync(d_dst, stage, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    return SK_OK;
  }
  std::vector<size_t>& edge = cfg->chunk_edge;
  make_chunks(cnt, true, edge);
  const size_t nch = edge.size() - 1;
  if (cfg->chunk_done.size() < nch) cfg->chunk_done = std::vector<std::atomic<int>>(nch + 8);
  for (size_t c = 0; c < nch; ++c) cfg->chunk_done[c].store(0, std::memory_order_relaxed);
  std::atomic<int> err{0};
  const int nthr = cfg->copy_threads > 0 ? cfg->copy_threads : copy_threads();
  const int f = stage_chunks(stage, src, edge, cfg->chunk_done, order != nullptr, nthr, [&](size_t lo, size_t hi) {
    if (hipMemcpyAsync(d_dst + lo, stage + lo, (hi - lo) * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
      err.store(1);
  });
  if (err.load()) return fail(SK_EHIP, "H2D copy failed");
  if (order) *order = (f & 2) ? SK_ORDER_INVALID : ((f & 1) ? SK_ORDER_UNSORTED : SK_ORDER_SORTED);
  return SK_OK;
}

// [d_src, d_src + cnt_each) -> dst_a and, if dst_b, [d_src + cnt_each, d_src + 2 cnt_each) -> dst_b.
// Returns after the data is in the caller's arrays (synchronises `st`).
int d2h_doubles2(sk_config* cfg, double* dst_a, double* dst_b, const double* d_src, size_t cnt_each, hipStream_t st) {
  if (cnt_each == 0) return SK_OK;
  const size_t total = dst_b ? 2 * cnt_each : cnt_each;
  if (is_pinned(dst_a) && (!dst_b || is_pinned(dst_b))) {
    HIPCHK(hipMemcpyAsync(dst_a, d_src, cnt_each * sizeof(double), hipMemcpyDeviceToHost, st));
    if (dst_b) HIPCHK(hipMemcpyAsync(dst_b, d_src + cnt_each, cnt_each * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SK_OK;
  }
  int rc = ensure_stage(&cfg->h_out, &cfg->hout_bytes, total);
  if (rc) return rc;
  double* stage = cfg->h_out;
  auto drain = [&](size_t a, size_t b) {  // flat range [a, b) of the staged data into the caller's arrays
    if (a < cnt_each) {
      const size_t e = b < cnt_each ? b : cnt_each;
      std::memcpy(dst_a + a, stage + a, (e - a) * sizeof(double));
    }
    if (b > cnt_each) {
      const size_t s = a > cnt_each ? a : cnt_each;
      std::memcpy(dst_b + (s - cnt_each), stage + s, (b - s) * sizeof(double));
    }
  };
  if (total <= kSerialMax) {
    HIPCHK(hipMemcpyAsync(stage, d_src, total * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    drain(0, total);
    return SK_OK;
  }
  std::vector<size_t>& edge = cfg->chunk_edge;
  make_chunks(total, false, edge);
  const size_t nch = edge.size() - 1;
  while (cfg->cp_events.size() < nch) {
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    cfg->cp_events.push_back(e);
  }
  for (size_t c = 0; c < nch; ++c) {
    const size_t lo = edge[c], hi = edge[c + 1];
    HIPCHK(hipMemcpyAsync(stage + lo, d_src + lo, (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(cfg->cp_events[c], st));
  }
  std::atomic<long long> ready{0};  // chunks landed in the staging buffer; -1: a copy failed
#pragma omp parallel num_threads(cfg->copy_threads > 0 ? cfg->copy_threads : copy_threads())
  {
    const int nt = omp_get_num_threads(), q = omp_get_thread_num();
    for (size_t c = 0; c < nch; ++c) {
      if (q == 0) {
        hipError_t e;
        while ((e = hipEventQuery(cfg->cp_events[c])) == hipErrorNotReady) _mm_pause();
        ready.store(e == hipSuccess ? (long long)c + 1 : -1, std::memory_order_release);
      } else {
        long long r;
        while ((r = ready.load(std::memory_order_acquire)) >= 0 && r < (long long)c + 1) _mm_pause();
      }
      if (ready.load(std::memory_order_acquire) < 0) break;
      const size_t lo = edge[c], hi = edge[c + 1];
      size_t a, b;
      slice_of(lo, hi, q, nt, a, b);
      if (b > a) drain(a, b);
    }
  }
  if (ready.load() < 0) {
    (void)hipStreamSynchronize(st);
    return fail(SK_EHIP, "D2H copy failed");
  }
  return SK_OK;
}

int d2h_doubles(sk_config* cfg, double* dst, const double* d_src, size_t cnt, hipStream_t st) {
  return d2h_doubles2(cfg, dst, nullptr, d_src, cnt, st);
}

}  // namespace sk

// ---------------------------------------------------------------------- memory helpers of the C ABI
extern "C" {

// Host-only hook (CPU test tier): the staging pass of the host-array upload on plain memory -- `stage` receives the
// copy, *order what the pass found out about the lags (1 sorted, 2 unsorted, 3 invalid).  threads <= 0: SK_COPY_THREADS.
int sk_host_stage_scan(const double* src, size_t cnt, double* stage, int threads, int* order) {
  if (!src || !stage || !order) return fail(SK_EINVAL, "null argument");
  int f = 0;
  if (cnt <= kSerialMax) {
    std::memcpy(stage, src, cnt * sizeof(double));
    f = scan_order(src, 0, cnt);
  } else {
    std::vector<size_t> edge;
    make_chunks(cnt, true, edge);
    std::vector<std::atomic<int>> done(edge.size() + 8);
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    f = stage_chunks(stage, src, edge, done, true, threads > 0 ? threads : copy_threads(), [](size_t, size_t) {});
  }
  *order = (f & 2) ? SK_ORDER_INVALID : ((f & 1) ? SK_ORDER_UNSORTED : SK_ORDER_SORTED);
  return SK_OK;
}

int sk_device_count(int* n) {
  if (!n) return fail(SK_EINVAL, "null argument");
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) {
    (void)hipGetLastError();
    c = 0;
  }
  *n = c;
  return SK_OK;
}

int sk_device_malloc(void** d_ptr, size_t bytes, int device) {
  if (!d_ptr) return fail(SK_EINVAL, "null argument");
  *d_ptr = nullptr;
  if (device >= 0) HIPCHK(hipSetDevice(device));
  HIPCHK(hipMalloc(d_ptr, bytes ? bytes : 8));
  return SK_OK;
}

int sk_device_free(void* d_ptr) {
  if (d_ptr) HIPCHK(hipFree(d_ptr));
  return SK_OK;
}

int sk_memcpy_h2d(void* d_dst, const void* src, size_t bytes) {
  if (bytes) HIPCHK(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
  return SK_OK;
}

int sk_memcpy_d2h(void* dst, const void* d_src, size_t bytes) {
  if (bytes) HIPCHK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return SK_OK;
}

int sk_device_synchronize(int device) {
  if (device >= 0) HIPCHK(hipSetDevice(device));
  HIPCHK(hipDeviceSynchronize());
  return SK_OK;
}

int sk_device_mem_info(size_t* free_bytes, size_t* total_bytes) {
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return SK_OK;
}

int sk_host_alloc(void** ptr, size_t bytes) {
  if (!ptr) return fail(SK_EINVAL, "null argument");
  *ptr = nullptr;
  HIPCHK(hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocDefault));
  return SK_OK;
}

int sk_host_free(void* ptr) {
  if (ptr) HIPCHK(hipHostFree(ptr));
  return SK_OK;
}

}  // extern "C"
