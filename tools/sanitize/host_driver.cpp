// host_driver.cpp -- drives the THREADED HOST CODE of the library under a sanitizer, without a device:
//   1. the staging pass of the host-array upload (sk_hostio.cpp: stage_chunks through sk_host_stage_scan; round 2's
//      medium-severity race lived here) with 1, 3 and 8 threads on arrays whose verdict differs exactly at slice boundaries;
//   2. the sharded entries (sk_shard.cpp) with the injected backend: ns in {2, 8}, blocked and strided split, sorted and
//      unsorted input with duplicates (parallel host merge sort + scatter), the pair-list entry (dense list and index pairs).
// Built and run by tools/sanitize/run_host_sanitizers.sh in the build container (-fsanitize=thread, then address,undefined);
// never on the GPU box.  Exit code 0 = every check held (the sanitizer reports on stderr and sets its own exit code).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../include/spectralkernels_hip.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      std::fprintf(stderr, "CHECK FAILED %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);                  \
      std::fprintf(stderr, "\n");                         \
      ++g_fail;                                           \
    }                                                     \
  } while (0)

static double truth(double x) { return std::exp(-x) * (1.0 + 0.25 * x); }

// injected shard backend: values of a smooth function, errors = shard index (so that the scatter can be checked)
static int backend(int shard, const double* xs, size_t n, double k0, int pd, double* vals, double* errs, void* ctx) {
  (void)pd;
  (void)ctx;
  for (size_t i = 0; i + 1 < n; ++i)
    if (!(xs[i] < xs[i + 1])) return SK_EINVAL;  // a shard always sees sorted unique lags
  for (size_t i = 0; i < n; ++i) {
    vals[i] = k0 * truth(xs[i]);
    errs[i] = (double)shard;
  }
  return SK_OK;
}

static void test_stage_scan() {
  const size_t n = (size_t)1 << 21;  // several chunks, slices of every thread count
  std::vector<double> a(n), b(n), stage(n);
  for (size_t i = 0; i < n; ++i) a[i] = 1e-6 * (double)(i + 1);
  for (int threads : {1, 3, 8}) {
    for (int rep = 0; rep < 6; ++rep) {
      b = a;
      int want = 1;
      if (rep % 3 == 1) {  // an equal pair exactly across a slice boundary of this thread count
        const size_t per = (n / 2 + threads - 1) / threads;
        const size_t q = std::min(n - 1, per * (size_t)(1 + rep % threads));
        b[q] = b[q - 1];
        want = 2;
      } else if (rep % 3 == 2) {
        b[n - 1 - (size_t)rep] = -1.0;
        want = 3;
      }
      int order = 0;
      const int rc = sk_host_stage_scan(b.data(), n, stage.data(), threads, &order);
      CHECK(rc == SK_OK, "sk_host_stage_scan rc %d", rc);
      CHECK(order == want, "threads %d rep %d: order %d, want %d", threads, rep, order, want);
      CHECK(std::equal(b.begin(), b.end(), stage.begin()), "threads %d rep %d: staged copy differs", threads, rep);
    }
  }
}

static void test_sharded() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(0.0, 5.0);
  const size_t n = 200000;
  std::vector<double> sorted(n), unsorted(n);
  for (size_t i = 0; i < n; ++i) sorted[i] = 1e-5 * (double)(i + 1);
  for (size_t i = 0; i < n; ++i) unsorted[i] = U(rng);
  for (size_t i = 0; i < n / 10; ++i) unsorted[n - 1 - i] = unsorted[i];  // duplicates
  unsorted[17] = 0.0;
  for (int ns : {2, 8}) {
    for (int strided = 0; strided < 2; ++strided) {
      for (int which = 0; which < 2; ++which) {
        const std::vector<double>& xs = which ? unsorted : sorted;
        sk_sharded* sh = nullptr;
        CHECK(sk_sharded_create_injected(&sh, ns, backend, nullptr) == SK_OK, "create");
        CHECK(sk_sharded_set_split(sh, strided) == SK_OK, "split");
        std::vector<double> v(n, -1.0), e(n, -1.0);
        const int rc = sk_kernel_values_sharded(sh, xs.data(), n, 2.0, 1, 0, v.data(), e.data());
        CHECK(rc == SK_OK, "sharded rc %d: %s", rc, sk_last_error());
        double worst = 0.0;
        for (size_t i = 0; i < n; ++i) worst = std::max(worst, std::fabs(v[i] - 2.0 * truth(xs[i])));
        CHECK(worst == 0.0, "ns %d strided %d input %d: worst %g", ns, strided, which, worst);
        int64_t covered = 0;
        for (int s = 0; s < ns; ++s) {
          sk_stats st;
          double ms;
          int64_t lo, hi;
          CHECK(sk_sharded_last(sh, s, &st, &ms, &lo, &hi) == SK_OK, "last");
          covered += st.n_unique;
        }
        CHECK(covered > 0 && covered <= (int64_t)n, "covered %lld", (long long)covered);
        sk_sharded_destroy(sh);
      }
    }
  }
  // pair lists: dense list of 600 points in the plane, and an explicit index list on the line
  const size_t npts = 600;
  std::vector<double> pts(2 * npts);
  for (double& p : pts) p = U(rng);
  for (int ns : {2, 8}) {
    sk_sharded* sh = nullptr;
    CHECK(sk_sharded_create_injected(&sh, ns, backend, nullptr) == SK_OK, "create");
    const size_t nd = 1 + npts * (npts - 1) / 2;
    std::vector<double> v(nd), e(nd);
    int rc = sk_kernel_values_pairs_sharded(sh, pts.data(), npts, 2, nullptr, 0, 1.0, 1, 0, v.data(), e.data());
    CHECK(rc == SK_OK, "pairs rc %d: %s", rc, sk_last_error());
    size_t q = 1;
    double worst = std::fabs(v[0] - truth(0.0));
    for (size_t i = 0; i < npts; ++i)
      for (size_t j = i + 1; j < npts; ++j, ++q) {
        const double dx = pts[2 * i] - pts[2 * j], dy = pts[2 * i + 1] - pts[2 * j + 1];
        worst = std::max(worst, std::fabs(v[q] - truth(std::sqrt(dx * dx + dy * dy))));
      }
    CHECK(worst == 0.0, "dense pairs ns %d: worst %g", ns, worst);
    std::vector<int64_t> pairs;
    for (size_t i = 0; i < 5000; ++i) {
      pairs.push_back((int64_t)(rng() % (2 * npts)));
      pairs.push_back((int64_t)(rng() % (2 * npts)));
    }
    std::vector<double> v1(5000), e1(5000);
    rc = sk_kernel_values_pairs_sharded(sh, pts.data(), 2 * npts, 1, pairs.data(), 5000, 1.0, 1, 0, v1.data(), e1.data());
    CHECK(rc == SK_OK, "pairs(1d) rc %d: %s", rc, sk_last_error());
    worst = 0.0;
    for (size_t i = 0; i < 5000; ++i)
      worst = std::max(worst, std::fabs(v1[i] - truth(std::fabs(pts[(size_t)pairs[2 * i]] - pts[(size_t)pairs[2 * i + 1]]))));
    CHECK(worst == 0.0, "index pairs ns %d: worst %g", ns, worst);
    sk_sharded_destroy(sh);
  }
  // error path: a failing shard is reported, nothing hangs
  {
    sk_sharded* sh = nullptr;
    sk_sharded_create_injected(&sh, 4, [](int shard, const double*, size_t, double, int, double*, double*, void*) {
      return shard == 2 ? (int)SK_EHIP : (int)SK_OK; }, nullptr);
    std::vector<double> v(n), e(n);
    const int rc = sk_kernel_values_sharded(sh, sorted.data(), n, 1.0, 1, 0, v.data(), e.data());
    CHECK(rc == SK_EHIP, "failing shard: rc %d", rc);
    sk_sharded_destroy(sh);
  }
}

int main() {
  test_stage_scan();
  test_sharded();
  std::printf("host_driver: %s\n", g_fail ? "FAILED" : "all checks held");
  return g_fail ? 1 : 0;
}
