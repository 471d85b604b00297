// sk_shard.cpp -- one process, several devices: kernel_values on a lag set sharded over the GPUs of a node
// (SURVEY 8e "independent mode"; north_star: "sharding the r-target set", K(0) the only shared quantity).
//
// The sorted unique lag set is cut into contiguous blocks, one per device.  Every device holds a replica of
// the config (rules, NUFFT geometry, workspaces, its own stream) and runs its own march on its block: its panel
// widths follow its own largest lag (src/adaptive.jl:152), so there is no exchange of any kind between the
// devices after K(0) -- computed once on the host and handed to every shard.  One host thread per device drives
// its shard through the ordinary host-array entry (staged upload, march, staged download), reading its block of
// the caller's lags and writing its block of the caller's result arrays in place: no gather step.
// Unsorted input or duplicates (src/adaptive.jl:99-120): one host sort + dedup, shards on the unique set,
// scatter back through the inverse map.
#include <omp.h>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <numeric>
#include <thread>
#include "sk_config.h"

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
#pragma omp parallel for reduction(| : uns, bad) schedule(static) num_threads(8)
  for (long long i = 0; i < (long long)n; ++i) {
    bad |= !(xs[i] >= 0.0);
    if (i + 1 < (long long)n) uns |= !(xs[i] < xs[i + 1]);
  }
  return bad ? 2 : (uns ? 1 : 0);
}

}  // namespace

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
    sh->cfgs.push_back(c);
    sh->devices.push_back(dev);
  }
  sh->stats.assign(ndev, sk_stats{});
  sh->ms.assign(ndev, 0.0);
  sh->lo.assign(ndev, 0);
  sh->hi.assign(ndev, 0);
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
  *out = sh;
  return SK_OK;
}

void sk_sharded_destroy(sk_sharded* sh) {
  if (!sh) return;
  for (sk_config* c : sh->cfgs) sk_config_destroy(c);
  delete sh;
}

int sk_sharded_count(const sk_sharded* sh) { return sh ? sh->nshard : 0; }

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
    // src/adaptive.jl:99-120: unique + sort on the host, inverse map for the scatter
    std::vector<int64_t> perm(n);
    std::iota(perm.begin(), perm.end(), (int64_t)0);
    std::sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return xs[a] < xs[b]; });
    sh->inv.resize(n);
    sh->uxs.clear();
    sh->uxs.reserve(n);
    for (size_t q = 0; q < n; ++q) {
      const double x = xs[perm[q]];
      if (q == 0 || x != sh->uxs.back()) sh->uxs.push_back(x);  // -0.0 == 0.0
      sh->inv[perm[q]] = (int64_t)sh->uxs.size() - 1;
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
  auto work = [&](int s) {
    int64_t lo, hi;
    block_of(nu, s, ns, lo, hi);
    sh->lo[s] = lo;
    sh->hi[s] = hi;
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (sh->backend) {
      rc = sh->backend(s, ux + lo, (size_t)(hi - lo), k0, param_derivative, uv + lo, ue + lo, sh->backend_ctx);
      std::memset(&sh->stats[s], 0, sizeof(sk_stats));
      sh->stats[s].n_unique = hi - lo;
    } else {
      rc = sk_kernel_values(sh->cfgs[s], ux + lo, (size_t)(hi - lo), k0, 1, param_derivative, uv + lo, ue + lo);
      sk_last_stats(sh->cfgs[s], &sh->stats[s]);
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
  if (ns == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int s = 1; s < ns; ++s) th.emplace_back(work, s);
    work(0);
    for (auto& t : th) t.join();
  }
  for (int s = 0; s < ns; ++s)
    if (rcs[s]) return fail(rcs[s], "shard " + std::to_string(s) + ": " + msgs[s]);
  if (order == 1) {
    const int64_t* inv = sh->inv.data();
#pragma omp parallel for schedule(static) num_threads(8)
    for (long long i = 0; i < (long long)n; ++i) {
      vals_out[i] = uv[inv[i]];
      errs_out[i] = ue[inv[i]];
    }
  }
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
