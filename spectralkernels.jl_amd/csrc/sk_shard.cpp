// sk_shard.cpp -- one process, several devices: kernel_values on a lag set sharded over the GPUs of a node
// (SURVEY 8e "independent mode"; north_star: "sharding the r-target set", K(0) the only shared quantity).
//
// The sorted unique lag set is cut into contiguous blocks, one per device.  Every device holds a replica of
// the config (rules, NUFFT geometry, workspaces, its own stream) and runs its own march on its block: its panel
// widths follow its own largest lag (src/adaptive.jl:152), so there is no exchange of any kind between the
// devices after K(0) -- computed once on the host and handed to every shard.  One host thread per device drives
// its shard through the ordinary host-array entry (staged upload, march, staged download), reading its block of
// the caller's lags and writing its block of the caller's result arrays in place: no gather step.
// Unsorted input or duplicates (src/adaptive.jl:99-120): one sort + dedup (on the first device; a parallel host merge
// sort with the injected CPU-tier backend), shards on the unique set, scatter back through the inverse map.
// Split of the sorted unique set: `blocked` (contiguous blocks, the default: no copies, but the shards' lag ranges and
// hence their panel schedules differ -- the block with the smallest lags marches longest) or `strided` (shard s takes
// lags s, s + ns, ...: every shard sees the whole range, the same schedule, equal work; one gather / scatter per shard).
// Pair lists (sk_kernel_values_pairs_sharded: src/model.jl:64-65, src/utils.jl:41-64): the points are replicated, the
// pair list is cut into blocks, every device forms, sorts and deduplicates the lags of its block itself -- no lag ever
// exists on the host.
#include <omp.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>
#include "sk_config.h"

// The shards' host threads, kept for the life of the handle: worker s runs shard s of every call (shard 0 runs on the
// caller's thread).  Threads created per call would each build -- and tear down at their exit -- an OpenMP team for the
// staging copies inside sk_kernel_values: milliseconds per call, more than the march of a shard (measured on the
// overlapped marches of sk_api.cpp: 9 ms against 2).
struct ShardPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::function<void(int)> job;
  unsigned long long gen = 0;
  int active = 0;     // workers that take part in the current generation: shards 1 .. active
  int pending = 0;
  bool stop = false;
  void start(int nworkers) {
    for (int w = 1; w <= nworkers; ++w)
      th.emplace_back([this, w] {
        unsigned long long seen = 0;
        for (;;) {
          std::function<void(int)> f;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            if (w > active) continue;
            f = job;
          }
          f(w);
          {
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
          }
        }
      });
  }
  // runs work(0) here and work(1 .. ns-1) on the workers; returns when all are done
  void run(int ns, const std::function<void(int)>& work) {
    if (ns <= 1 || th.empty()) {
      for (int s = 0; s < ns; ++s) work(s);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      job = work;
      active = std::min(ns - 1, (int)th.size());
      pending = active;
      ++gen;
    }
    cv_go.notify_all();
    work(0);
    for (int s = active + 1; s < ns; ++s) work(s);  // (never: one worker per shard)
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~ShardPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_go.notify_all();
    for (auto& t : th) t.join();
  }
};

struct sk_sharded {
  std::vector<sk_config*> cfgs;   // one replica per device (empty with an injected backend)
  std::vector<int> devices;
  int nshard = 0;
  sk_shard_fn backend = nullptr;  // test hook: replaces the device call of a shard
  void* backend_ctx = nullptr;
  sk_config_args args;
  // last call
  std::vector<sk_stats> stats;
  std::vector<double> ms;
  std::vector<int64_t> lo, hi;    // block bounds in the sorted unique set
  int64_t n_unique = 0;
  std::vector<double> uxs, uvals, uerrs;  // unsorted input: unique set and its results
  std::vector<int64_t> inv;
  int split = 0;                  // 0 blocked, 1 strided
  ShardPool pool;                 // host threads of shards 1 .. nshard-1
};

namespace {

// contiguous block [lo, hi) of shard s (sizes differ by at most one)
void block_of(int64_t n, int s, int ns, int64_t& lo, int64_t& hi) {
  const int64_t base = n / ns, rem = n % ns;
  lo = s * base + std::min<int64_t>(s, rem);
  hi = lo + base + (s < rem ? 1 : 0);
}

// 0: strictly increasing and valid, 1: not strictly increasing, 2: a negative / NaN lag
int scan_lags(const double* xs, size_t n) {
  int uns = 0, bad = 0;
#pragma omp parallel for reduction(| : uns, bad) schedule(static) num_threads(sk::host_threads(8))
  for (long long i = 0; i < (long long)n; ++i) {
    bad |= !(xs[i] >= 0.0);
    if (i + 1 < (long long)n) uns |= !(xs[i] < xs[i + 1]);
  }
  return bad ? 2 : (uns ? 1 : 0);
}

// Sort + dedup of `n` lags on the host: T chunks sorted in parallel, then rounds of pairwise merges (the merges of a
// round in parallel).  uxs: sorted unique values; inv[i]: position of xs[i] in uxs.
void host_sort_unique(const double* xs, size_t n, std::vector<double>& uxs, std::vector<int64_t>& inv) {
  struct KV { double x; int64_t i; };
  std::vector<KV> a(n), b(n);
  int T = sk::host_threads(16);
  while (T > 1 && n / T < 4096) T /= 2;
  int P = 1;
  while (P * 2 <= T) P *= 2;  // power of two chunks
  std::vector<size_t> edge(P + 1);
  for (int q = 0; q <= P; ++q) edge[q] = n * (size_t)q / P;
  auto less = [](const KV& l, const KV& r) { return l.x < r.x || (l.x == r.x && l.i < r.i); };
#pragma omp parallel for schedule(static, 1) num_threads(P)
  for (int q = 0; q < P; ++q) {
    for (size_t i = edge[q]; i < edge[q + 1]; ++i) a[i] = KV{xs[i] == 0.0 ? 0.0 : xs[i], (int64_t)i};  // -0.0 == 0.0
    std::sort(a.begin() + edge[q], a.begin() + edge[q + 1], less);
  }
  KV* src = a.data();
  KV* dst = b.data();
  for (int w = 1; w < P; w *= 2) {
#pragma omp parallel for schedule(static, 1) num_threads(P / (2 * w))
    for (int q = 0; q < P; q += 2 * w)
      std::merge(src + edge[q], src + edge[q + w], src + edge[q + w], src + edge[q + 2 * w], dst + edge[q], less);
    std::swap(src, dst);
  }
  uxs.clear();
  uxs.reserve(n);
  inv.resize(n);
  for (size_t q = 0; q < n; ++q) {
    if (q == 0 || src[q].x != uxs.back()) uxs.push_back(src[q].x);
    inv[(size_t)src[q].i] = (int64_t)uxs.size() - 1;
  }
}

}  // namespace

namespace sk {
// sk_api.cpp: sort + dedup of host lags on the config's device (rocPRIM radix sort, as the single-device entry)
int sort_unique_on_device(sk_config* cfg, const double* xs, size_t n, std::vector<double>& uxs, std::vector<int64_t>& inv);
// sk_api.cpp: the dense pair list restricted to entries [e0, e0 + count) (sk_kernel_values_pairs otherwise)
int kernel_values_pairs_range(sk_config* cfg, const double* pts, size_t npts, int pdim, const int64_t* pairs, size_t e0,
                              size_t count, double k0, int param_derivative, double* vals_out, double* errs_out);
}  // namespace sk

extern "C" {

int sk_sharded_create(sk_sharded** out, const sk_config_args* args, const int* devices, int ndev) {
  if (!out || !args || ndev < 1) return fail(SK_EINVAL, "sk_sharded_create: bad arguments");
  *out = nullptr;
  int count = 0;
  sk_device_count(&count);
  sk_sharded* sh = new sk_sharded();
  sh->args = *args;
  sh->nshard = ndev;
  int warn = SK_OK;
  for (int s = 0; s < ndev; ++s) {
    const int dev = devices ? devices[s] : s;
    if (dev < 0 || dev >= count) {
      sk_sharded_destroy(sh);
      return fail(SK_ENODEVICE, "sk_sharded_create: device " + std::to_string(dev) + " is not visible (" +
                                    std::to_string(count) + " devices)");
    }
    sk_config_args a = *args;
    a.device = dev;
    sk_config* c = nullptr;
    const int rc = sk_config_create(&c, &a);
    if (rc != SK_OK && rc != SK_WARN_QUADSPEC) {
      sk_sharded_destroy(sh);
      return rc;
    }
    if (rc == SK_WARN_QUADSPEC) warn = rc;
    // the staging threads of all shards run at once: share the cores (8 per shard alone would be 64 on an 8-GPU node)
    c->copy_threads = std::max(1, std::min(8, sk::host_threads(0) / std::max(1, ndev)));
    sh->cfgs.push_back(c);
    sh->devices.push_back(dev);
  }
  sh->stats.assign(ndev, sk_stats{});
  sh->ms.assign(ndev, 0.0);
  sh->lo.assign(ndev, 0);
  sh->hi.assign(ndev, 0);
  sh->pool.start(ndev - 1);
  *out = sh;
  return warn;
}

// Test hook (CPU tier): `nshard` shards whose device call is replaced by `fn` -- no device is touched.
int sk_sharded_create_injected(sk_sharded** out, int nshard, sk_shard_fn fn, void* ctx) {
  if (!out || nshard < 1 || !fn) return fail(SK_EINVAL, "sk_sharded_create_injected: bad arguments");
  sk_sharded* sh = new sk_sharded();
  std::memset(&sh->args, 0, sizeof(sh->args));
  sh->nshard = nshard;
  sh->backend = fn;
  sh->backend_ctx = ctx;
  sh->stats.assign(nshard, sk_stats{});
  sh->ms.assign(nshard, 0.0);
  sh->lo.assign(nshard, 0);
  sh->hi.assign(nshard, 0);
  sh->pool.start(nshard - 1);
  *out = sh;
  return SK_OK;
}

void sk_sharded_destroy(sk_sharded* sh) {
  if (!sh) return;
  for (sk_config* c : sh->cfgs) sk_config_destroy(c);
  delete sh;
}

int sk_sharded_count(const sk_sharded* sh) { return sh ? sh->nshard : 0; }

int sk_sharded_set_split(sk_sharded* sh, int strided) {
  if (!sh) return fail(SK_EINVAL, "null handle");
  sh->split = strided ? 1 : 0;
  return SK_OK;
}

int sk_kernel_values_sharded(sk_sharded* sh, const double* xs, size_t n, double k0, int has_k0, int param_derivative,
                             double* vals_out, double* errs_out) {
  if (!sh) return fail(SK_EINVAL, "null handle");
  if (n == 0) return SK_OK;
  if (!has_k0) {  // the shared normaliser (src/adaptive.jl:97): once, on the host
    if (sh->cfgs.empty()) return fail(SK_EINVAL, "k0 is required with an injected backend");
    int rc = sk_compute_k0(sh->cfgs[0], &k0);
    if (rc) return rc;
  }
  const int order = scan_lags(xs, n);
  if (order == 2) return fail(SK_EINVAL, "lags must be non-negative and finite");
  const double* ux = xs;
  double *uv = vals_out, *ue = errs_out;
  int64_t nu = (int64_t)n;
  if (order == 1) {
    // src/adaptive.jl:99-120: unique + sort (first device; host merge sort without one), inverse map for the scatter
    if (!sh->cfgs.empty()) {
      int rc = sk::sort_unique_on_device(sh->cfgs[0], xs, n, sh->uxs, sh->inv);
      if (rc) return rc;
    } else {
      host_sort_unique(xs, n, sh->uxs, sh->inv);
    }
    nu = (int64_t)sh->uxs.size();
    sh->uvals.assign(nu, 0.0);
    sh->uerrs.assign(nu, 0.0);
    ux = sh->uxs.data();
    uv = sh->uvals.data();
    ue = sh->uerrs.data();
  }
  sh->n_unique = nu;
  const int ns = (int)std::min<int64_t>(sh->nshard, nu);  // never an empty shard
  std::vector<int> rcs(sh->nshard, SK_OK);
  std::vector<std::string> msgs(sh->nshard);
  const bool strided = sh->split == 1 && ns > 1;
  auto work = [&](int s) {
    int64_t lo, hi;
    block_of(nu, s, ns, lo, hi);
    const double* sx = ux + lo;
    double *sv = uv + lo, *se = ue + lo;
    size_t cnt = (size_t)(hi - lo);
    std::vector<double> gx, gv, ge;
    if (strided) {  // lags s, s + ns, ... of the sorted unique set: gathered here, scattered back below
      lo = s;
      hi = nu;
      cnt = (size_t)((nu - s + ns - 1) / ns);
      gx.resize(cnt);
      gv.resize(cnt);
      ge.resize(cnt);
      for (size_t q = 0; q < cnt; ++q) gx[q] = ux[s + (int64_t)q * ns];
      sx = gx.data();
      sv = gv.data();
      se = ge.data();
    }
    sh->lo[s] = lo;
    sh->hi[s] = hi;
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (sh->backend) {
      rc = sh->backend(s, sx, cnt, k0, param_derivative, sv, se, sh->backend_ctx);
      std::memset(&sh->stats[s], 0, sizeof(sk_stats));
      sh->stats[s].n_unique = (int64_t)cnt;
    } else {
      rc = sk_kernel_values(sh->cfgs[s], sx, cnt, k0, 1, param_derivative, sv, se);
      sk_last_stats(sh->cfgs[s], &sh->stats[s]);
    }
    if (strided && !rc)
      for (size_t q = 0; q < cnt; ++q) {
        uv[s + (int64_t)q * ns] = gv[q];
        ue[s + (int64_t)q * ns] = ge[q];
      }
    sh->ms[s] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rcs[s] = rc;
    if (rc) msgs[s] = sk_last_error();  // the error channel is thread-local: carry the message to the caller
  };
  for (int s = ns; s < sh->nshard; ++s) {
    sh->lo[s] = sh->hi[s] = nu;
    sh->ms[s] = 0.0;
    std::memset(&sh->stats[s], 0, sizeof(sk_stats));
  }
  sh->pool.run(ns, work);
  for (int s = 0; s < ns; ++s)
    if (rcs[s]) return fail(rcs[s], "shard " + std::to_string(s) + ": " + msgs[s]);
  if (order == 1) {
    const int64_t* inv = sh->inv.data();
#pragma omp parallel for schedule(static) num_threads(sk::host_threads(8))
    for (long long i = 0; i < (long long)n; ++i) {
      vals_out[i] = uv[inv[i]];
      errs_out[i] = ue[inv[i]];
    }
  }
  return SK_OK;
}

int sk_kernel_values_pairs_sharded(sk_sharded* sh, const double* pts, size_t npts, int pdim, const int64_t* pairs, size_t npairs,
                                   double k0, int has_k0, int param_derivative, double* vals_out, double* errs_out) {
  if (!sh) return fail(SK_EINVAL, "null handle");
  if (pdim < 1 || npts < 1) return fail(SK_EINVAL, "need at least one point and pdim >= 1");
  const size_t n = pairs ? npairs : 1 + npts * (npts - 1) / 2;
  if (n == 0) return SK_OK;
  if (!has_k0) {
    if (sh->cfgs.empty()) return fail(SK_EINVAL, "k0 is required with an injected backend");
    int rc = sk_compute_k0(sh->cfgs[0], &k0);
    if (rc) return rc;
  }
  sh->n_unique = 0;
  const int ns = (int)std::min<size_t>((size_t)sh->nshard, n);
  std::vector<int> rcs(sh->nshard, SK_OK);
  std::vector<std::string> msgs(sh->nshard);
  auto work = [&](int s) {
    int64_t lo, hi;
    block_of((int64_t)n, s, ns, lo, hi);  // block of the pair list = block of the outputs: no scatter
    sh->lo[s] = lo;
    sh->hi[s] = hi;
    const size_t cnt = (size_t)(hi - lo);
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (sh->backend) {
      // CPU-tier hook: the lags of the block formed here, as k_pair_lags forms them on the device
      std::vector<double> lag(cnt);
      for (size_t q = 0; q < cnt; ++q) {
        const size_t e = (size_t)lo + q;
        size_t i, j;
        if (pairs) {
          i = (size_t)pairs[2 * e];
          j = (size_t)pairs[2 * e + 1];
        } else if (e == 0) {
          lag[q] = 0.0;
          continue;
        } else {  // entry 1 + i npts - i (i + 1) / 2 + (j - i - 1) is |p_i - p_j|, i < j (src/utils.jl:43-46)
          size_t r = e - 1;
          i = 0;
          while (r >= npts - 1 - i) {
            r -= npts - 1 - i;
            ++i;
          }
          j = i + 1 + r;
        }
        double d2 = 0.0;
        for (int c = 0; c < pdim; ++c) {
          const double d = pts[i * pdim + c] - pts[j * pdim + c];
          d2 += d * d;
        }
        lag[q] = pdim == 1 ? std::fabs(pts[i] - pts[j]) : std::sqrt(d2);
      }
      // the backend takes sorted unique lags, like a device shard's march
      std::vector<double> ux;
      std::vector<int64_t> inv;
      host_sort_unique(lag.data(), cnt, ux, inv);
      std::vector<double> v(ux.size()), er(ux.size());
      rc = sh->backend(s, ux.data(), ux.size(), k0, param_derivative, v.data(), er.data(), sh->backend_ctx);
      for (size_t q = 0; q < cnt && !rc; ++q) {
        vals_out[lo + (int64_t)q] = v[(size_t)inv[q]];
        errs_out[lo + (int64_t)q] = er[(size_t)inv[q]];
      }
      std::memset(&sh->stats[s], 0, sizeof(sk_stats));
      sh->stats[s].n_unique = (int64_t)ux.size();
    } else {
      rc = sk::kernel_values_pairs_range(sh->cfgs[s], pts, npts, pdim, pairs, (size_t)lo, cnt, k0, param_derivative,
                                         vals_out + lo, errs_out + lo);
      sk_last_stats(sh->cfgs[s], &sh->stats[s]);
    }
    sh->ms[s] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rcs[s] = rc;
    if (rc) msgs[s] = sk_last_error();
  };
  for (int s = ns; s < sh->nshard; ++s) {
    sh->lo[s] = sh->hi[s] = (int64_t)n;
    sh->ms[s] = 0.0;
    std::memset(&sh->stats[s], 0, sizeof(sk_stats));
  }
  sh->pool.run(ns, work);
  for (int s = 0; s < ns; ++s)
    if (rcs[s]) return fail(rcs[s], "shard " + std::to_string(s) + ": " + msgs[s]);
  return SK_OK;
}

int sk_sharded_last(const sk_sharded* sh, int shard, sk_stats* st, double* ms, int64_t* lo, int64_t* hi) {
  if (!sh || shard < 0 || shard >= sh->nshard) return fail(SK_EINVAL, "bad shard index");
  if (st) *st = sh->stats[shard];
  if (ms) *ms = sh->ms[shard];
  if (lo) *lo = sh->lo[shard];
  if (hi) *hi = sh->hi[shard];
  return SK_OK;
}

}  // extern "C"
