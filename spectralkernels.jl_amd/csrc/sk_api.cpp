// sk_api.cpp -- C ABI (include/spectralkernels_hip.h), config construction and the host driver
// of the device-resident panel march.  Host-side mirror of src/adaptive.jl:24-108.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "sk_config.h"

namespace sk {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

static std::once_flag g_fft_once;

static int64_t next_smooth_even(int64_t n) {
  if (n < 2) n = 2;
  for (int64_t v = n + (n & 1);; v += 2) {
    int64_t t = v;
    for (int p : {2, 3, 5, 7})
      while (t % p == 0) t /= p;
    if (t == 1) return v;
  }
}

}  // namespace sk

namespace sk {

template <class T>
static int upload(sk_config* cfg, const std::vector<T>& v, const T** out) {
  void* d = nullptr;
  size_t bytes = v.size() * sizeof(T);
  HIPCHK(hipMalloc(&d, bytes ? bytes : 8));
  cfg->owned.push_back(d);
  if (bytes) HIPCHK(hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice));
  *out = (const T*)d;
  return SK_OK;
}

template <class T>
static int dalloc(sk_config* cfg, size_t count, T** out) {
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, count * sizeof(T) ? count * sizeof(T) : 8));
  cfg->owned.push_back(d);
  *out = (T*)d;
  return SK_OK;
}

static int validate(const sk_config_args& a, int64_t& m, int64_t& k, int& warned) {
  if (a.criteria < 0 || a.criteria > 2)
    return fail(SK_EINVAL, "Argument convergence_criteria must be one of :panel, :tails, :both.");
  if (a.dim < 1) return fail(SK_EINVAL, "dim must be >= 1");
  if (a.alpha >= a.dim) return fail(SK_EINVAL, "alpha must be less than dim to be integrable.");
  if (a.m < 1 || a.k < 1) return fail(SK_EINVAL, "quadspec entries must be positive");
  if (!(a.tol > 0)) return fail(SK_EINVAL, "tol must be positive");
  if (a.batch_panels != 0 && a.batch_panels != 1 && a.batch_panels != 2 && a.batch_panels != 4 && a.batch_panels != 8)
    return fail(SK_EINVAL, "batch_panels must be 0 (auto), 1, 2, 4 or 8");
  m = a.m;
  k = a.k;
  warned = 0;
  if (a.tol < 1e-12 && m * k > 4096) {  // src/adaptive.jl:37-40
    warned = 1;
    m = 4096;
    k = 1;
  }
  double p = -a.alpha + (a.dim == 1 ? 0.0 : a.dim / 2.0) + (a.derivative ? 1.0 : 0.0);
  if (p != 0.0 && p <= -1.0) return fail(SK_EINVAL, "p needs to be in (-1.0, Inf) to be integrable");
  if (a.sdf_kind == SK_SDF_CALLBACK && !a.f) return fail(SK_EINVAL, "sdf callback is null");
  if (a.sdf_kind < 0 || a.sdf_kind > SK_SDF_GAUSSIAN) return fail(SK_EINVAL, "unknown sdf_kind");
  {
    const int np = a.sdf_kind == SK_SDF_MATERN ? 3 : (a.sdf_kind == SK_SDF_CALLBACK ? 0 : 1);
    if (a.sdf_dparam < 0 || a.sdf_dparam > np) return fail(SK_EINVAL, "sdf_dparam out of range for this sdf_kind");
    if (a.sdf_dparam && a.logw) return fail(SK_EUNSUPPORTED, "logw=true with a parameter-derivative integrand");
  }
  return SK_OK;
}

static int choose_width(const sk_config_args& a) {
  if (a.nufft_width) {
    int w = a.nufft_width;
    w = w < 4 ? 4 : (w > 16 ? 16 : w);
    return w + (w & 1);
  }
  // NUFFT error target = tol/1000 (the reference always asks FINUFFT for 1e-15); w = digits + 1
  double eps = std::fmax(a.tol * 1e-3, 1e-15);
  int w = (int)std::ceil(-std::log10(eps)) + 1;
  w += (w & 1);
  return w < 4 ? 4 : (w > 16 ? 16 : w);
}

// composite-rule source positions in grid units for one variant/rule, plus cell starts
static void build_positions(const sk_config* cfg, int variant, int rule, std::vector<double>& pos,
                            std::vector<int>& cellstart) {
  const int64_t m = cfg->m, k = cfg->k, n1 = cfg->dc.n1;
  const int64_t mr = rule ? 2 * m : m;
  const std::vector<double>& leg = cfg->leg[rule ? 2 : 0];
  const std::vector<double>& jac = cfg->jac[rule ? 2 : 0];
  pos.resize((size_t)(mr * k));
  const long double delta = 2.0L / (long double)(n1 - cfg->dc.w);
  for (int64_t sp = 0; sp < k; ++sp)
    for (int64_t q = 0; q < mr; ++q) {
      long double t = (variant == 1 && sp == 0) ? (long double)jac[q] : (long double)leg[q];
      long double u = ((2.0L * sp + 1.0L) / (long double)k - 1.0L) + t / (long double)k;
      pos[(size_t)(sp * mr + q)] = (double)(u / delta + 0.5L * (long double)n1);
    }
  cellstart.assign((size_t)n1 + 1, 0);
  std::vector<int> cnt((size_t)n1 + 1, 0);
  for (double P : pos) {
    int64_t cidx = (int64_t)std::floor(P);
    cidx = cidx < 0 ? 0 : (cidx >= n1 ? n1 - 1 : cidx);
    cnt[(size_t)cidx + 1]++;
  }
  for (int64_t i = 0; i < n1; ++i) cellstart[(size_t)i + 1] = cellstart[(size_t)i] + cnt[(size_t)i + 1];
}

// Spread work list of one variant: tiles of <= maxc cells holding <= ~cap touching sources where the node density
// allows it (Gauss nodes cluster like 1/m^2 at every sub-panel edge), heaviest first.
static void build_tiles_geom(const sk_config* cfg, const std::vector<int> cs[2], int64_t maxc, int64_t cap,
                             std::vector<int4>& tiles) {
  const int64_t n1 = cfg->dc.n1;
  const int W = cfg->dc.w;
  struct T { int key, start, len, cnt, jlo, jhi; };
  std::vector<T> all;
  for (int rule = 0; rule < 2; ++rule) {
    const std::vector<int>& c = cs[rule];
    auto touching = [&](int64_t a, int64_t b) {  // sources in cells [a - W/2, b + W/2)
      int64_t lo = a - W / 2 < 0 ? 0 : a - W / 2, hi = b + W / 2 > n1 ? n1 : b + W / 2;
      return c[(size_t)hi] - c[(size_t)lo];
    };
    int64_t start = 0;
    while (start < n1) {
      int64_t end = start + 4 > n1 ? n1 : start + 4;
      while (end < n1 && end - start < maxc && touching(start, end + 1) <= cap) ++end;
      const int64_t lo = start - W / 2 < 0 ? 0 : start - W / 2, hi = end + W / 2 > n1 ? n1 : end + W / 2;
      all.push_back({(rule << 30) | (int)start, (int)start, (int)(end - start), touching(start, end), c[(size_t)lo],
                     c[(size_t)hi]});
      start = end;
    }
  }
  std::stable_sort(all.begin(), all.end(), [](const T& a, const T& b) { return a.cnt > b.cnt; });
  tiles.clear();
  for (const T& t : all) tiles.push_back(make_int4(t.key, t.len, t.jlo, t.jhi));
}

// The tile geometry by the rule's size.  A tile is one workgroup's unit of latency: loads, the staging of its sources
// (one dependent FP64 chain per source: ES weights, strength), the gather over the sources touching each cell, the
// reduction and the write.  The default rule (n1 = 65 536) has 2 236 tiles of 64 cells / <= 256 sources for ~1 000
// resident workgroups: big tiles amortise that chain.  A small rule has few of them -- quadspec (256, 16): 128 tiles on
// 256 CUs, the dense ones (252 sources: every thread both the weights and the strength of a source, ten gather rounds)
// three times as long as the light ones, and the launch ends with the slowest.  Smaller tiles (<= 128 sources: two
// threads share a source; fewer cells: several lanes share a cell's gather) spread the same work over the idle CUs
// until the tile count reaches about 0.6 of the resident workgroups, beyond which the replicated state prologue and the
// halo (W cells re-staged per tile) cost more than the shorter chain saves.  Measured (tools/diag/tile_sweep.py, 10^5
// lags, ms per call, default -> chosen): (64,4) 0.585 -> 0.55, (128,8) 0.258 -> 0.236, (256,16) 0.183 -> 0.168 and, at
// tol 1e-12 (BASELINE config 5), 1.150 -> 1.02, (512,16) 0.163 -> 0.151, (1024,16) 0.176 -> 0.168, (2048,16) 0.208 ->
// 0.194, (4096,16) unchanged.
static void build_tiles(const sk_config* cfg, const std::vector<int> cs[2], int resident, std::vector<int4>& tiles) {
  const char* ec = getenv("SK_TILE_CELLS");  // diagnostics: a fixed geometry
  const char* es = getenv("SK_TILE_SRC");
  if (ec || es) {
    int64_t maxc = ec ? std::atoi(ec) : 64, cap = es ? std::atoi(es) : 256;
    build_tiles_geom(cfg, cs, maxc < 4 ? 4 : (maxc > 64 ? 64 : maxc), cap < 16 ? 16 : cap, tiles);
    return;
  }
  const size_t one_round = (size_t)(0.6 * resident), two_rounds = (size_t)(1.5 * resident);
  static const int ladder[][2] = {{64, 128}, {64, 96}, {48, 96}, {32, 96}, {32, 64}, {16, 64}};
  build_tiles_geom(cfg, cs, ladder[0][0], ladder[0][1], tiles);
  if (tiles.size() > two_rounds) {
    // large rules.  dim 1: every workgroup has several tiles anyway, big ones amortise best.  dim >= 2 (k_spread_multi:
    // one workgroup per tile, batches of 128 sources carrying every series term): a tile of <= 128 sources is one batch
    // -- 44 -> 28.5 us per launch on BASELINE config 4
    if (cfg->dc.dim == 1) build_tiles_geom(cfg, cs, 64, 256, tiles);
    return;
  }
  std::vector<int4> next;
  for (size_t q = 1; q < sizeof(ladder) / sizeof(ladder[0]); ++q) {
    build_tiles_geom(cfg, cs, ladder[q][0], ladder[q][1], next);
    if (next.size() > one_round) break;
    tiles.swap(next);
  }
}

static int ensure_ws(sk_config* cfg, size_t bytes) {
  if (bytes <= cfg->ws_bytes) return SK_OK;
  if (cfg->ws) HIPCHK(hipFree(cfg->ws));
  cfg->ws = nullptr;
  cfg->ws_bytes = 0;
  size_t want = bytes + bytes / 4;
  HIPCHK(hipMalloc(&cfg->ws, want));
  cfg->ws_bytes = want;
  return SK_OK;
}

static const char* status_message(int code) {
  switch (code) {
    case SK_ESUBDIVIDE:
      return "The sub-interval (a, b) has been split too many times (b - a < 1e-16). Exiting to avoid infinite splitting.";
    case SK_ENAN: return "NaN detected in panel integral...";
    case SK_ESTACK: return "splitting stack capacity (5000) exceeded";
    case SK_EHIP: return "internal error: inconsistent device state detected by a kernel guard";
    default: return "device-side failure";
  }
}

// host nodes of the current (sub)panel for callback SDFs (same arithmetic as k_gen)
static void host_nodes(const sk_config* cfg, const PanelParams& pp, std::vector<double>& w) {
  const int64_t m = cfg->m, k = cfg->k, M = cfg->M;
  w.resize((size_t)(3 * M));
  for (int rule = 0; rule < 2; ++rule) {
    const int64_t mr = rule ? 2 * m : m;
    const std::vector<double>& no = (rule ? cfg->leg[2] : cfg->leg[0]);
    const std::vector<double>& jno = (rule ? cfg->jac[2] : cfg->jac[0]);
    double* out = w.data() + (rule ? M : 0);
    for (int64_t sp = 0; sp < k; ++sp) {
      double sa = (sp == 0) ? pp.pa : pp.pa + (pp.pb - pp.pa) * ((double)sp / (double)k);
      double sb = (sp + 1 == k) ? pp.pb : pp.pa + (pp.pb - pp.pa) * ((double)(sp + 1) / (double)k);
      double bmad2 = 0.5 * (sb - sa), bpad2 = 0.5 * (sb + sa);
      const std::vector<double>& nn = (pp.jacobi && sp == 0) ? jno : no;
      for (int64_t q = 0; q < mr; ++q) out[sp * mr + q] = bmad2 * nn[q] + bpad2;
    }
  }
}

static hipEvent_t ev_get(sk_config* cfg) {
  if (cfg->ev_used == cfg->ev_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    cfg->ev_pool.push_back(e);
  }
  return cfg->ev_pool[cfg->ev_used++];
}
// SK_DEBUG_SYNC=1: name every launch group on stderr and synchronise after it, so that an asynchronous
// GPU fault can be attributed to a kernel (diagnostics only; serialises the pipeline).
static const bool g_debug_sync = getenv("SK_DEBUG_SYNC") != nullptr;
static const bool g_no_fuse = getenv("SK_NO_FUSE") != nullptr;  // diagnostics: separate k_gen and k_spread launches
static const bool g_debug_hk = getenv("SK_DEBUG_HK") != nullptr;    // diagnostics: the plan of every Hankel step
static const bool g_hk_serial = getenv("SK_HK_SERIAL") != nullptr;  // diagnostics: Hankel boxes on the march's stream
static const bool g_no_stream = getenv("SK_NO_STREAM") != nullptr;  // diagnostics: rounds with k_advance throughout
static const bool g_debug_batch = getenv("SK_DEBUG_BATCH") != nullptr;
static const bool g_no_hybrid = getenv("SK_NO_HYBRID") != nullptr;  // diagnostics: no one-panel launches inside a full-width march  // diagnostics: one step per round, every state printed
static const char* const k_names[SK_K_COUNT] = {"k_gen", "k_spread", "rocfft", "k_interp", "k_advance", "k_hk_near"};

// SK_ROCTX=1: roctx ranges named after the reference's @timeit sections (src/adaptive.jl:156-199,
// src/quadrature.jl:99-160) around the launches that do their work, for rocprofv3 --marker-trace.  The library is
// looked up at run time (librocprofiler-sdk-roctx, then libroctx64): no link-time dependency, no cost when off.
static const char* const k_roctx_names[SK_K_COUNT] = {
    "update quadrature buffers", "FINUFFT call / spread", "FINUFFT call / fft",
    "FINUFFT call / interpolate + add panels to integral + check convergence", "interval integral / step state",
    "NUFHT call / near field"};
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (!getenv("SK_ROCTX")) return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
      }
    }
    push = nullptr;
    pop = nullptr;
  }
};
static Roctx g_roctx;
struct RoctxRange {
  explicit RoctxRange(const char* name) { if (g_roctx.push) g_roctx.push(name); }
  ~RoctxRange() { if (g_roctx.pop) g_roctx.pop(); }
};

struct EvScope {
  sk_config* cfg;
  hipStream_t st;
  int kid;
  hipEvent_t a, b;
  EvScope(sk_config* c, hipStream_t s, int k) : cfg(c), st(s), kid(k) {
    if (g_roctx.push) g_roctx.push(k_roctx_names[kid]);
    if (g_debug_sync) fprintf(stderr, "[sk] launch %s\n", k_names[kid]);
    if (cfg->profile) {
      a = ev_get(cfg);
      b = ev_get(cfg);
      (void)hipEventRecord(a, st);
    }
  }
  ~EvScope() {
    if (g_roctx.pop) g_roctx.pop();
    if (g_debug_sync) {
      hipError_t e = hipStreamSynchronize(st);
      fprintf(stderr, "[sk]   %s -> %s\n", k_names[kid], hipGetErrorString(e));
    }
    if (cfg->profile) {
      (void)hipEventRecord(b, st);
      cfg->ev_recs.push_back({kid, a, b});
    }
  }
};

static int run_fft(sk_config* cfg, hipStream_t st) {
  FFTCHK(rocfft_execution_info_set_stream(cfg->info, st));
  void* in[1] = {cfg->db.fft};
  void* out[1] = {cfg->db.fft_out};
  FFTCHK(rocfft_execute(cfg->plan, in, out, cfg->info));
  return SK_OK;
}

// Hankel path: the batched transform of the first nt series terms (both rules), in place in db.hk_fft
static int run_fft_hk(sk_config* cfg, int nt, hipStream_t st) {
  const int full = hk_terms(cfg->dc.w);
  rocfft_plan plan = cfg->plan_hk;
  rocfft_execution_info info = cfg->info_hk;
  if (nt < full) {
    if (!cfg->plan_hk_nt[nt]) {
      size_t len2[1] = {(size_t)cfg->dc.n2};
      FFTCHK(rocfft_plan_create(&cfg->plan_hk_nt[nt], rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                                rocfft_precision_double, 1, len2, (size_t)2 * nt, nullptr));
      size_t wb = 0;
      FFTCHK(rocfft_plan_get_work_buffer_size(cfg->plan_hk_nt[nt], &wb));
      FFTCHK(rocfft_execution_info_create(&cfg->info_hk_nt[nt]));
      if (wb) {
        HIPCHK(hipMalloc(&cfg->fft_work_hk_nt[nt], wb));
        FFTCHK(rocfft_execution_info_set_work_buffer(cfg->info_hk_nt[nt], cfg->fft_work_hk_nt[nt], wb));
      }
    }
    plan = cfg->plan_hk_nt[nt];
    info = cfg->info_hk_nt[nt];
  }
  FFTCHK(rocfft_execution_info_set_stream(info, st));
  void* in[1] = {cfg->db.hk_fft};
  FFTCHK(rocfft_execute(plan, in, nullptr, info));
  return SK_OK;
}

static int run_fft_b(sk_config* cfg, hipStream_t st, bool one_panel = false) {
  const bool one = (one_panel || cfg->dc.K == 1) && cfg->plan_b1;
  rocfft_execution_info info = one ? cfg->info_b1 : cfg->info_b;
  FFTCHK(rocfft_execution_info_set_stream(info, st));
  void* in[1] = {cfg->db.bfft};
  void* out[1] = {cfg->db.bfft_out};
  FFTCHK(rocfft_execute(one ? cfg->plan_b1 : cfg->plan_b, in, out, info));
  return SK_OK;
}

// Waits until k_advance has published status block number `seq` (pinned host memory, polled).
static int wait_status(sk_config* cfg, unsigned int seq, hipStream_t st) {
  volatile unsigned int* p = &cfg->h_status->seq;
  unsigned long long spins = 0;
  while (*p != seq) {
    if ((++spins & 0x3fff) == 0) {
      hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) {
        if (*p == seq) break;
        return fail(SK_EHIP, "device finished without publishing its status block");
      }
      if (q != hipErrorNotReady) return fail(SK_EHIP, std::string("HIP error: ") + hipGetErrorString(q));
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  cfg->stats.n_syncs++;
  return SK_OK;
}

int joint_wait_status(sk_config* cfg, unsigned int seq, hipStream_t st) { return wait_status(cfg, seq, st); }
void joint_host_nodes(const sk_config* cfg, const PanelParams& pp, std::vector<double>& w) { host_nodes(cfg, pp, w); }

// One step of the Hankel path by boxes (sk_hankel_inner.inc) for Bessel order nu (a_nu: its asymptotic coefficients) at
// the state the device published in `hs`: one NUFFT triple for the lags above r_c = r_hi / 8^l, a directly summed near box
// for the few sources below om0 there, the inner box (all sources, lags up to r_c) by moments.  l from a cost model in
// Bessel evaluations; the lag counts per power of 8 come from k_advance.  boxes = false: the step does not fit the
// buffers (the caller falls back to the level split).  Shared by the single-integrand march (run_core) and the joint
// march of sk_joint.cpp.
int hk_plan_boxes(sk_config* cfg, DevBuffers& db, const DevState* hs, int nu, const double* a_nu, int raw, hipStream_t st,
                  HkInner& hin, bool& boxes) {
  const double r_hi = hs->pp.r_hi, pb = hs->pp.pb;
  int L = 0;
  while (L < kHkMaxLevels - 1 && kHkZ0 * std::pow(kHkBeta, L + 1) / (2.0 * M_PI * r_hi) < pb) ++L;
  std::memset(&hin, 0, sizeof(hin));
  boxes = false;
  if (cfg->hk_nb > 0) {
    auto n_of_z = [](double z) { return std::ceil(z + 9.0 * std::cbrt(z) + 14.0); };
    // series terms: the default count is sized for z0 = 24 (hk_terms); from a larger first z, the fewest terms with
    // the same first omitted term
    const int nt_full = hk_terms(cfg->dc.w);
    double ak[kHkTerms + 2];
    ak[0] = 1.0;
    for (int q = 1; q <= kHkTerms + 1; ++q) ak[q] = ak[q - 1] * (4.0 * nu * nu - (2.0 * q - 1.0) * (2.0 * q - 1.0)) / (8.0 * q);
    const double thr = std::fabs(ak[nt_full]) / std::pow(kHkZ0, nt_full);
    auto terms_for = [&](double z) {
      for (int K = 2; K < nt_full; ++K)
        if (std::fabs(ak[K]) / std::pow(z, K) <= thr) return K;
      return nt_full;
    };
    const double pa = hs->pp.pa;
    const long long nact = hs->hi - hs->ix1;
    hin.nu = nu;
    hk_series_fill(hin.ser, a_nu, nu);
    for (int K = 0; K <= kHkTerms; ++K)  // K terms suffice from z = (|a_K| / thr)^(1/K) on (terms_for, per workgroup of k_hk_far)
      hin.zterm[K] = (K >= 2 && K <= nt_full && ak[K] != 0.0) ? std::pow(std::fabs(ak[K]) / thr, 1.0 / K) : INFINITY;
    hin.pa = pa;
    hin.pb = pb;
    hin.r_hi = r_hi;
    hin.lag_lo = hs->ix1;
    hin.escale = (raw == SK_RAW_NONE && cfg->dc.dim > 2) ? cfg->dc.dim * 0.5 - 1.0 : 0.0;
    // bands of the inner box for a given r_c: the fewest (divisors of k) with at most ~400 moments each
    auto bands_for = [&](double r_c, int& nb, int& ngm) {
      nb = 0;
      for (int d = 1; d <= (int)cfg->k; ++d) {
        if (cfg->k % d) continue;
        const int ng = (int)n_of_z(M_PI * (pb - pa) / d * r_c) + 2;
        if (ng > cfg->hk_ngmax) continue;
        nb = d;
        ngm = ng;
        if (ng <= 400) break;
      }
      return nb > 0;
    };
    if (L == 0) {  // no far field: the whole panel is one smooth box
      int nb, ngm;
      const double qd = std::ceil(M_PI * pb * r_hi / 96.0);
      if (bands_for(r_hi, nb, ngm) && qd <= (double)kHkQCap) {
        hin.has_far = 0; hin.r_c = r_hi; hin.nb = nb; hin.ngm = ngm;
        hin.Q = qd < 1.0 ? 1 : (int)qd;
        hin.lag_cnt = nact;
        boxes = true;
      }
    } else {
      double best = INFINITY;
      for (int l = 1; l < kHkMaxLevels; ++l) {
        const double beta = std::pow(kHkBeta, l), r_c = r_hi / beta;
        double om0 = kHkZ0 * beta / (2.0 * M_PI * r_hi);
        if (om0 >= pb) break;  // nothing left for the series
        const long long icnt = cfg->h_status->lvl_cnt[l];
        // a panel that starts beyond om0 has no near sources; the series then runs from z >= 2 pi pa r_c > z0 on and
        // needs fewer terms for the same truncation error
        int nt = nt_full;
        if (om0 <= pa) {
          om0 = pa;
          nt = terms_for(2.0 * M_PI * pa * r_c);
        }
        const double fnear = om0 <= pa ? 0.0 : (om0 - pa) / (pb - pa);
        const double qn = fnear > 0.0 ? std::ceil(M_PI * om0 * (r_hi - r_c) / 96.0) : 0.0;
        double cost = 3.0 * (double)cfg->M * fnear * qn * kHkNodes;
        // spread + FFT: 8.6 us per term at 196 608 sources; taps: 7.4 us per term and 1e6 lags; 1 us ~ 7e4 evaluations
        cost += nt * (3.06 * 3.0 * (double)cfg->M + 0.52 * (double)(nact - icnt));
        int nb = 1, ngm = 0;
        double qi = 0.0;
        if (icnt > 0) {
          if (!bands_for(r_c, nb, ngm)) continue;
          qi = std::max(1.0, std::ceil(M_PI * pb * r_c / 96.0));
          const double zb = M_PI * (pb - pa) / nb * r_c;
          cost += qi * kHkNodes * nb * n_of_z(0.5 * zb) + 3.0 * (double)cfg->M * ngm / 22.0 + qi * 2.0 * nb * ngm * ngm / 110.0;
        }
        if (qn + qi > (double)kHkQCap) continue;
        if (cost < best) {
          best = cost;
          hin.has_far = 1; hin.r_c = r_c; hin.om0 = om0; hin.nb = nb; hin.ngm = ngm; hin.nt = nt;
          hin.Qn = (int)qn; hin.Q = (int)qi;
          hin.lag_cnt = icnt;
          boxes = true;
        }
      }
    }
    if (boxes) {
      hin.gs = (int)cfg->k / hin.nb;
      hin.ell = hin.Q > 0 ? hin.r_c / (double)hin.Q : hin.r_c;
      hin.elln = hin.Qn > 0 ? (r_hi - hin.r_c) / (double)hin.Qn : r_hi;
      const int qtot = hin.Q + hin.Qn;
      // interval buffers (grow only; rare: the counts grow with the panel index): equivalent sources of the inner
      // box, node values and Chebyshev coefficients of both boxes
      const size_t wb = ((sizeof(double2) * (size_t)hin.Q * 2 * hin.nb * hin.ngm + 255) / 256) * 256;
      const size_t vb = ((sizeof(double) * (size_t)qtot * 2 * kHkNodes + 255) / 256) * 256;
      if (wb + 2 * vb > cfg->hk_qbytes) {
        HIPCHK(hipStreamSynchronize(st));
        if (cfg->hk_qbuf) HIPCHK(hipFree(cfg->hk_qbuf));
        cfg->hk_qbuf = nullptr;
        cfg->hk_qbytes = 0;
        const size_t want = (wb + 2 * vb) + (wb + 2 * vb) / 2 + (1u << 20);
        HIPCHK(hipMalloc(&cfg->hk_qbuf, want));
        cfg->hk_qbytes = want;
      }
      db.hk_w = (double2*)cfg->hk_qbuf;
      db.hk_v = (double*)((char*)cfg->hk_qbuf + wb);
      db.hk_c2 = (double*)((char*)cfg->hk_qbuf + wb + vb);
    }
  }
  if (boxes && g_debug_hk)
    fprintf(stderr, "[sk] hankel step: panel [%.6g, %.6g] r_hi %.6g  far %d r_c %.4g om0 %.5g nt %d  Qn %d  inner: lags %lld Q %d nb %d ngm %d\n",
            hin.pa, hin.pb, hin.r_hi, hin.has_far, hin.r_c, hin.om0, hin.nt, hin.Qn, hin.lag_cnt, hin.Q, hin.nb, hin.ngm);
  return SK_OK;
}

// The launches of one planned step: spread / FFT / far kernel on `st`, the two boxes beside them on the config's side
// streams (the far kernel waits for the near box only), then the interpolation of the lags up to r_c.  raw != 0: the
// evaluation kernels store the panel sums (db.lw_raw) instead of running the epilogue.
int hk_launch_boxes(sk_config* cfg, DevBuffers& db, const HkInner& hin, int t, long long n, int raw, hipStream_t st) {
  // the boxes (a chain of small launches) run beside the spread and the FFT, which do not depend on them; the
  // per-kernel event timing (sk_profile_enable) keeps everything on one stream
  const bool beside = hin.has_far && !cfg->profile && !g_hk_serial;
  const bool has_near = hin.Qn > 0, has_inner = hin.Q > 0;
  if (beside) {
    // three chains start at once: the spread (issued first) and the FFT on the march's stream, the near box -- all
    // the far kernel waits for, and only a panel that starts below om0 has one -- and the longer chain of the
    // inner box, which only the interpolation of the lags up to r_c waits for
    HIPCHK(hipEventRecord(cfg->fork_event, st));
    if (has_near) HIPCHK(hipStreamWaitEvent(cfg->side_stream2, cfg->fork_event, 0));
    if (has_inner) HIPCHK(hipStreamWaitEvent(cfg->side_stream, cfg->fork_event, 0));
    launch_hk_spread(cfg->dc, cfg->d_ec, db, t & 1, hin.om0, hin.nt, st);
    {
      // the FFT is enqueued before the boxes' seven launches: issued behind them it reached the device 45 us after the
      // spread launch had ended (a launch costs the host 4-6 us; the far kernel waits for the FFT, not for the host).
      // Tried: the boxes' streams at the lowest priority (0.632 against 0.60-0.62 ms), the inner box's chain started
      // behind the spread (equal) or behind the FFT (0.68 ms).
      int rcf = run_fft_hk(cfg, hin.nt, st);
      if (rcf) return rcf;
    }
    if (has_near) {
      launch_hk_boxes(cfg->dc, db, hin, cfg->side_stream2, 1);
      HIPCHK(hipEventRecord(cfg->join_event2, cfg->side_stream2));
    }
    if (has_inner) {
      launch_hk_boxes(cfg->dc, db, hin, cfg->side_stream, 2);
      HIPCHK(hipEventRecord(cfg->join_event, cfg->side_stream));
    }
  } else {
    EvScope e(cfg, st, SK_K_NEAR);
    launch_hk_boxes(cfg->dc, db, hin, st, 3);
  }
  if (hin.has_far) {
    if (!beside) {
      { EvScope e(cfg, st, SK_K_SPREAD); launch_hk_spread(cfg->dc, cfg->d_ec, db, t & 1, hin.om0, hin.nt, st); }
      EvScope e(cfg, st, SK_K_FFT);
      int rcf = run_fft_hk(cfg, hin.nt, st);
      if (rcf) return rcf;
    }
    if (beside && has_near) HIPCHK(hipStreamWaitEvent(st, cfg->join_event2, 0));
    { EvScope e(cfg, st, SK_K_INTERP); launch_hk_far(cfg->dc, cfg->d_ec, db, hin, t & 1, n, raw, st); }
  }
  if (hin.lag_cnt > 0) {
    if (beside && has_inner) HIPCHK(hipStreamWaitEvent(st, cfg->join_event, 0));
    EvScope e(cfg, st, SK_K_NEAR);
    launch_hk_iinterp(cfg->dc, db, hin, t & 1, n, raw, st);
  } else if (beside && has_inner) {
    HIPCHK(hipStreamWaitEvent(st, cfg->join_event, 0));  // (never alone in practice: an inner box has lags)
  }
  return SK_OK;
}

// core march on sorted unique lags already on the device (src/adaptive.jl:110-202)
static int run_core(sk_config* cfg, const double* d_uxs, long long n, double k0, double ks0, double* d_ks, double* d_errs,
                    double* d_scratch6n /* 10 n for logw */, hipStream_t st, bool check_flags,
                    hipEvent_t lags_ready = nullptr) {
  // lags_ready: the bulk of the lags is still being uploaded on another stream (only xs[0] and xs[n-1] are in
  // place): the first step's node generation, spreading and FFT need no more than the largest lag
  // (src/adaptive.jl:152) and run beside the upload; everything that reads all lags waits for the event
  DevBuffers& db = cfg->db;
  db.xs = d_uxs;
  db.ks = d_ks;
  db.errs = d_errs;
  db.accK[0] = d_scratch6n;
  db.accK[1] = db.accK[0] + n;
  db.accE[0] = db.accK[1] + n;
  db.accE[1] = db.accE[0] + n;
  db.P[0] = db.accE[1] + n;
  db.P[1] = db.P[0] + n;
  db.lw_raw = db.P[1] + n;  // 4 n doubles, only allocated (and used) for logw configs
  // batched march: three buffers each for the running sums and the outer-panel partial sums (9 n doubles)
  for (int q = 0; q < 3; ++q) {
    db.bK[q] = d_scratch6n + (size_t)q * n;
    db.bE[q] = d_scratch6n + (size_t)(3 + q) * n;
    db.bP[q] = d_scratch6n + (size_t)(6 + q) * n;
  }
  db.fvals = cfg->d_fvals;
  const bool cb = (cfg->dc.sdf_kind == SK_SDF_CALLBACK);
  const bool logw = cfg->dc.logw != 0;
  // dim >= 2: fast Hankel transform (host-driven per step: the level split depends on r_hi) or direct sums
  const int hk_min = cfg->args.hankel_min_lags == 0 ? kHkDefaultMinLags : cfg->args.hankel_min_lags;
  const bool hankel = cfg->dc.dim > 1 && hk_min > 0 && n >= hk_min;
  // the host needs state_t before each step: callback SDFs (node values), Hankel (level split), logw (the first
  // panel takes two passes and a boundary term, src/quadrature.jl:186-228)
  const bool host_step = cb || hankel || logw;
  if (lags_ready && (host_step || cfg->dc.dim != 1 || g_no_fuse)) {
    HIPCHK(hipStreamWaitEvent(st, lags_ready, 0));
    lags_ready = nullptr;
  }
  db.check_input = (check_flags && !host_step && !lags_ready) ? 1 : 0;
  if (check_flags && host_step) launch_check_sorted(d_uxs, n, db.flags, st);  // k_gen never runs in INIT mode here
  sk_stats& stt = cfg->stats;
  const DevState* hs = &cfg->h_status->state;
  int t = 0;             // step counter: step t evaluates the panel described by state_t
  int mode = SK_MODE_INIT;
  // device-resident dim-1 march on a catalogue SDF: batched steps (sk_batch_kernels.inc)
  const bool batched = !host_step && cfg->dc.dim == 1 && !g_no_fuse;
  if (batched && cfg->k_adaptive) {
    // Automatic batch width.  A config whose last full-width march never met a step at which panels could be batched
    // (every panel a fresh outer panel of a march that is retiring lags: BASELINE config 3) runs the lighter one-panel
    // instantiations; such a march gives the full width's bits (every step holds one panel either way), reports a step
    // at which the full width COULD batch (dc.Kv, DevState::would_batch) and is then repeated at the full width below --
    // the result is a function of the input, not of the call history.
    cfg->dc.K = cfg->narrow ? 1 : cfg->k_full;
    cfg->dc.Kv = cfg->narrow ? cfg->k_full : 0;
  }
  int R = (host_step || batched) ? 1 : 2;
  const int nu_main = cfg->dc.derivative ? cfg->dc.dim / 2 : cfg->dc.dim / 2 - 1;  // src/quadrature.jl:176-180

  // callback SDFs: integrand `g` at the nodes of the current (sub)panel, evaluated on the host
  // (src/quadrature.jl:184-247) and uploaded for k_gen
  auto upload_fvals = [&](int g, std::vector<double>& hv) -> int {
    const std::vector<double>& w = cfg->h_nodes;
    hv.resize(w.size());
    cfg->sdf.batch(w.data(), hv.data(), w.size());
    if (logw) {
      if (g == SK_G_LOGA) {
        cfg->h_dfvals.resize(w.size());
        cfg->dsdf.batch(w.data(), cfg->h_dfvals.data(), w.size());
        for (size_t q = 0; q < w.size(); ++q) hv[q] = hv[q] + w[q] * std::log(w[q]) * cfg->h_dfvals[q];
      } else if (g == SK_G_LOGB) {
        for (size_t q = 0; q < w.size(); ++q) hv[q] = w[q] * std::log(w[q]) * hv[q];
      } else {
        for (size_t q = 0; q < w.size(); ++q) hv[q] *= std::log(w[q]);
      }
    }
    // through a pinned buffer of its own per pass (the A and B passes of a logw panel are enqueued back to back)
    const int slot = (g == SK_G_LOGB) ? 1 : 0;
    if (!cfg->h_fvpin[slot])
      HIPCHK(hipHostMalloc((void**)&cfg->h_fvpin[slot], sizeof(double) * 3 * (size_t)cfg->M, hipHostMallocDefault));
    std::memcpy(cfg->h_fvpin[slot], hv.data(), sizeof(double) * hv.size());
    HIPCHK(hipMemcpyAsync(cfg->d_fvals, cfg->h_fvpin[slot], sizeof(double) * hv.size(), hipMemcpyHostToDevice, st));
    return SK_OK;
  };

  // one evaluation of the current (sub)panel with integrand g and Bessel order nu (dim > 1); raw != 0 stores
  // the panel sums instead of running the epilogue
  unsigned int stream_seq = 0;  // != 0: streaming phase of the batched march (see below)
  bool gen_ahead = false;
  // One-panel launches inside a full-width march (automatic batch width only).  While the march retires lags every step
  // is one fresh outer panel (the "peel": 29 of BASELINE config 5's 33 steps); the K = 1 instantiations evaluate such a
  // step with the same bits as the full width's and ~3 us sooner.  The spread launch of step t says whether step t was
  // of that kind (HostStatus::peel_hint, stored ahead of `progress`); if so the host launches step t + 1 with the
  // one-panel kernels, which evaluate it only if it is of that kind too and otherwise skip it (HostStatus::need_wide;
  // the host launches it again at the full width).  Every decision is a function of device-published state of the
  // step before -- never of timing -- and every evaluated step has the full width's bits: the result is the full-width
  // march's, bit for bit (tests/test_gpu_batched.py).  After two skipped steps a call stays at the full width.
  const bool hybrid_ok = batched && cfg->k_adaptive && cfg->k_full > 1 && cfg->dc.K == cfg->k_full && cfg->plan_b1 && !g_no_hybrid;
  // (tests: every step first with the one-panel kernels, whatever the hint -- they skip what they cannot take)
  const bool hybrid_force = hybrid_ok && getenv("SK_HYBRID_FORCE") != nullptr;
  bool narrow_step = false;
  DevConst dcn = cfg->dc;
  dcn.K = 1;
  dcn.Kv = 0;
  dcn.hybrid = 1;
  auto eval_pass = [&](int g, int nu, int raw, std::vector<double>& hv) -> int {
    if (cb) {
      int rc = upload_fvals(g, hv);
      if (rc) return rc;
    }
    if (batched) {
      const DevConst& dcl = narrow_step ? dcn : cfg->dc;
      { EvScope e(cfg, st, SK_K_SPREAD); launch_spread_b(dcl, cfg->d_ec, db, mode, t, n, k0, ks0, stream_seq, st); }
      {
        EvScope e(cfg, st, SK_K_FFT);
        int rc = run_fft_b(cfg, st, narrow_step);
        if (rc) return rc;
      }
      if (lags_ready) {  // first evaluation: from here on every lag is read
        HIPCHK(hipStreamWaitEvent(st, lags_ready, 0));
        lags_ready = nullptr;
        if (check_flags) launch_check_sorted(d_uxs, n, db.flags, st);
      }
      { EvScope e(cfg, st, SK_K_INTERP); launch_interp_b(dcl, cfg->d_ec, db, t, n, st); }
      return SK_OK;
    }
    if (gen_ahead) gen_ahead = false;  // issued behind k_advance, ahead of the host round trip
    else { EvScope e(cfg, st, SK_K_GEN); launch_gen(cfg->dc, db, mode, t, n, k0, ks0, g, st); }
    if (cfg->dc.dim == 1) {
      { EvScope e(cfg, st, SK_K_SPREAD); launch_spread(cfg->dc, cfg->d_ec, db, t & 1, st); }
      {
        EvScope e(cfg, st, SK_K_FFT);
        int rc = run_fft(cfg, st);
        if (rc) return rc;
      }
      if (lags_ready) {  // first evaluation: from here on every lag is read
        HIPCHK(hipStreamWaitEvent(st, lags_ready, 0));
        lags_ready = nullptr;
        if (check_flags) launch_check_sorted(d_uxs, n, db.flags, st);
      }
      { EvScope e(cfg, st, SK_K_INTERP); launch_interp(cfg->dc, cfg->d_ec, db, t & 1, n, raw, st); }
    } else if (hankel) {
      // ---- the Hankel path (see "nonuniform fast Hankel transform" in sk_kernels.hip)
      HankelPlan hp;
      std::memset(&hp, 0, sizeof(hp));
      hp.nu = nu;
      std::memcpy(hp.a, nu == nu_main ? cfg->hk_a : cfg->hk_a2, sizeof(hp.a));
      const double r_hi = hs->pp.r_hi, pb = hs->pp.pb;
      hp.pa = hs->pp.pa;
      hp.pb = pb;
      int L = 0;
      while (L < kHkMaxLevels - 1 && kHkZ0 * std::pow(kHkBeta, L + 1) / (2.0 * M_PI * r_hi) < pb) ++L;
      // ---- by boxes (sk_hankel_inner.inc): one NUFFT triple for the lags above r_c = r_hi / 8^l, a directly summed
      // near box for the few sources below om0 there, the inner box (all sources, lags up to r_c) by moments.  l from
      // a cost model in Bessel evaluations; the lag counts per power of 8 come from k_advance.  Falls back to the level
      // split when a step does not fit the buffers.
      HkInner hin;
      bool boxes = false;
      {
        int rcp = hk_plan_boxes(cfg, db, hs, nu, hp.a, raw, st, hin, boxes);
        if (rcp) return rcp;
      }
      if (boxes) return hk_launch_boxes(cfg, db, hin, t, n, raw, st);
      // level planning with the lag counts k_advance published: a level (and everything inside it) with at most
      // kHkNodes lags becomes the innermost one and is summed directly at those lags
      for (int l = 0; l <= L; ++l)
        if (cfg->h_status->lvl_cnt[l] <= kHkNodes) {
          L = l;
          hp.direct_inner = 1;
          hp.inner_lo = hs->ix1;
          hp.inner_cnt = cfg->h_status->lvl_cnt[l];
          break;
        }
      hp.L = L;
      hp.near_levels = L + 1;
      for (int l = 0; l <= L; ++l) {
        hp.edge[l] = r_hi / std::pow(kHkBeta, l);
        hp.om[l] = (l < L) ? kHkZ0 * std::pow(kHkBeta, l + 1) / (2.0 * M_PI * r_hi) : INFINITY;
      }
      hp.edge[0] = r_hi;
      hp.edge[L + 1] = 0.0;
      { EvScope e(cfg, st, SK_K_NEAR); launch_hk_near(cfg->dc, db, hp, st); }
      for (int l = 0; l < L; ++l) {
        { EvScope e(cfg, st, SK_K_SPREAD); launch_hk_spread(cfg->dc, cfg->d_ec, db, t & 1, hp.om[l], hk_terms(cfg->dc.w), st); }
        {
          EvScope e(cfg, st, SK_K_FFT);
          FFTCHK(rocfft_execution_info_set_stream(cfg->info_hk, st));
          void* in[1] = {db.hk_fft};
          FFTCHK(rocfft_execute(cfg->plan_hk, in, nullptr, cfg->info_hk));
        }
        { EvScope e(cfg, st, SK_K_INTERP); launch_hk_interp(cfg->dc, cfg->d_ec, db, hp, l, t & 1, n, raw, st); }
      }
      { EvScope e(cfg, st, SK_K_INTERP); launch_hk_interp(cfg->dc, cfg->d_ec, db, hp, L, t & 1, n, raw, st); }
    } else {
      // isotropic d-dimensional kernel, few lags: direct Bessel sums (src/quadrature.jl:145-160)
      if (!host_step) launch_tail(cfg->dc, db, t & 1, st);
      EvScope e(cfg, st, SK_K_INTERP);
      launch_bessel(cfg->dc, db, t & 1, n, nu, raw, st);
    }
    return SK_OK;
  };

  // ---- streaming phase of the batched march.  No round structure: the host keeps ONE step queued behind the running
  // one (the spread launch of step t reports t + 1 once it has derived its state) and stops when a spread launch
  // reports "done" -- at most two steps are launched in vain (typically one), and the device never waits for a host
  // round trip.  Entered after the first step's k_advance (the quick marches end there), or at once when the last
  // march on this config took more than one step (then that k_advance -- 8-16 us plus a host round trip -- buys nothing).
  bool flags_up = false;
  auto stream_rest = [&]() -> int {
    stream_seq = ++cfg->seq;
    volatile unsigned int* prog = &cfg->h_status->progress;
    volatile unsigned int* dseq = &cfg->h_status->done_seq;
    volatile int* hflags = cfg->h_status->flags;
    volatile unsigned int* hint = &cfg->h_status->peel_hint;
    volatile unsigned int* need = &cfg->h_status->need_wide;
    const int t_first = t;
    *hint = 0u;
    *need = 0u;
    *prog = (unsigned int)t;
    const int depth = 1;  // launched-but-not-started steps: one queued behind the running one
    unsigned long long spins = 0;
    int skips = 0;
    bool again_wide = false;
    for (;;) {
      while ((unsigned int)t < *prog + (unsigned int)depth && *dseq != stream_seq) {
        // the kind of step t - 1 (its spread launch stored it ahead of the progress word read above; step t is not
        // launched yet, so nothing newer can be there)
        narrow_step = hybrid_ok && !again_wide && skips < 2 && t > t_first && *hint == 1u;
        if (hybrid_force && !again_wide && t > t_first + 1) narrow_step = true;
        again_wide = false;
        if (narrow_step) cfg->hyb_narrow++;
        // input of unknown order: the first step's launches scanned the lags; the spread launch of the step after it
        // reports what they found (with its progress word) -- stop marching on unsorted / invalid input
        if (check_flags && t_first == 0 && *prog >= 2u && (hflags[0] || hflags[1])) {
          flags_up = true;
          break;
        }
        int rc2 = eval_pass(SK_G_DEFAULT, nu_main, SK_RAW_NONE, cfg->h_fvals);
        if (rc2) return rc2;
        ++t;
        mode = SK_MODE_TRANSITION;
        stt.n_steps_launched++;
      }
      if (flags_up || *dseq == stream_seq) break;
      if (hybrid_ok && t > t_first && *need == (unsigned int)t) {
        // the one-panel launch of step t - 1 met a state at which the full width might batch and evaluated nothing
        // (its FFT and interpolation launches were no-ops): the same step again, at the full width
        std::atomic_thread_fence(std::memory_order_acquire);
        *need = 0u;
        --t;
        ++skips;
        cfg->hyb_skips++;
        again_wide = true;
        continue;
      }
      if ((++spins & 0x3fff) == 0) {
        hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess && *dseq != stream_seq && *prog + (unsigned int)depth <= (unsigned int)t)
          return fail(SK_EHIP, "device finished without reporting progress");
        if (q != hipSuccess && q != hipErrorNotReady) return fail(SK_EHIP, std::string("HIP error: ") + hipGetErrorString(q));
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    narrow_step = false;
    cfg->stats.n_syncs++;
    if (cfg->h_status->flags[1] & 2) return fail(SK_EHIP, "internal error: inconsistent device state detected by a kernel guard");
    if (!flags_up && check_flags && (cfg->h_status->flags[0] || cfg->h_status->flags[1])) flags_up = true;
    if (flags_up) HIPCHK(hipStreamSynchronize(st));  // the queued step must not run into the caller's sort
    return SK_OK;
  };
  if (batched && !g_no_stream && !g_debug_batch && cfg->last_n_steps >= 2) {
    int rcs = stream_rest();
    if (rcs) return rcs;
    if (flags_up) return SK_OK;
  } else
  for (;;) {
    bool lw_first = false;
    if (host_step) {
      // host closure: needs state_t (panel bounds) before the step can be generated
      const unsigned int seq = ++cfg->seq;
      launch_advance(cfg->dc, db, mode, t, n, k0, ks0, seq, st);
      // dim > 1, catalogue SDF, no log factors: the node generation needs the device's state only (not the host's plan
      // of the step): it runs while the status travels; a finished march makes it a no-op
      gen_ahead = false;
      if (!cb && !logw && cfg->dc.dim > 1) {
        EvScope e(cfg, st, SK_K_GEN);
        launch_gen(cfg->dc, db, SK_MODE_ASIS, t, n, k0, ks0, SK_G_DEFAULT, st);
        gen_ahead = true;
      }
      int rc = wait_status(cfg, seq, st);
      if (rc) return rc;
      if (check_flags && (cfg->h_status->flags[0] || cfg->h_status->flags[1])) return SK_OK;
      if (hs->done) break;
      if (cb) {
        host_nodes(cfg, hs->pp, cfg->h_nodes);
        *cfg->h_scal = cfg->sdf(hs->b);  // pinned; the previous iteration's copy completed before its status block
        HIPCHK(hipMemcpyAsync(&db.state[t & 1].fb, cfg->h_scal, sizeof(double), hipMemcpyHostToDevice, st));
      }
      if (cb) launch_tail(cfg->dc, db, t & 1, st);  // catalogue SDFs: k_advance has fitted the tail already
      mode = SK_MODE_ASIS;
      lw_first = logw && hs->pp.jacobi;  // _a == 0 && p != 0 (src/quadrature.jl:185-186)
    }
    for (int r = 0; r < R; ++r) {
      if (lw_first) {
        // I = (I0 - A + 2 pi x B)/(dim - alpha): A with the cos / J_{dim/2-1} kernel, B with sin / J_{dim/2}
        int rc = eval_pass(SK_G_LOGA, cfg->dc.dim / 2 - 1, SK_RAW_A, cfg->h_fvals);
        if (!rc) rc = eval_pass(SK_G_LOGB, cfg->dc.dim / 2, SK_RAW_B, cfg->h_fvals2);
        if (rc) return rc;
        const double pb = hs->pp.pb;
        const double B0 = std::pow(pb, cfg->dc.dim / 2.0 + 1.0 - cfg->dc.alpha) * std::log(pb) * cfg->sdf(pb);  // :189
        EvScope e(cfg, st, SK_K_INTERP);
        launch_lw_finish(cfg->dc, db, t & 1, n, B0, st);
      } else {
        int rc = eval_pass(SK_G_DEFAULT, nu_main, SK_RAW_NONE, cfg->h_fvals);
        if (rc) return rc;
      }
      ++t;
      mode = SK_MODE_TRANSITION;
      stt.n_steps_launched++;
    }
    if (!host_step) {
      const unsigned int seq = ++cfg->seq;
      {
        EvScope e(cfg, st, SK_K_ADVANCE);
        if (batched) launch_advance_b(cfg->dc, db, mode, t, n, k0, ks0, seq, st);
        else launch_advance(cfg->dc, db, mode, t, n, k0, ks0, seq, st);
      }
      int rc = wait_status(cfg, seq, st);
      if (rc) return rc;
      if (cfg->h_status->flags[1] & 2) return fail(SK_EHIP, "internal error: inconsistent device state detected by a kernel guard");
      if (check_flags && (cfg->h_status->flags[0] || cfg->h_status->flags[1])) return SK_OK;
      if (batched && g_debug_batch) {
        const BatchInfo& bi = hs->bt;
        fprintf(stderr, "[sk] t=%d done=%d hi=%lld kb=%d kind=%d cmask=%d spec_ok=%d pending=%d force=%d conf=%d stack=%d "
                        "panel [%.9g, %.9g] slots:", t, hs->done, hs->hi, bi.kb, bi.kind, bi.cmask, hs->spec_ok, bi.tails_pending,
                bi.force_kb, hs->conf, hs->stack_len, hs->pp.pa, hs->pp.pb);
        for (int q = 0; q < bi.kb; ++q) fprintf(stderr, " [%.9g,%.9g t1=%.3e t2=%.3e c=%d]", bi.pa[q], bi.pb[q], bi.tail[q].t1, bi.tail[q].t2num, bi.tail[q].crit);
        fprintf(stderr, " fix[%lld,%lld) n_panels=%lld\n", hs->fix_lo, hs->fix_hi, hs->n_panels);
      }
      if (hs->done) break;
      mode = SK_MODE_ASIS;
      R = g_debug_batch ? 1 : (R < 16 ? 2 * R : 16);
      if (batched && !g_no_stream && !g_debug_batch) {
        int rcs = stream_rest();
        if (rcs) return rcs;
        if (flags_up) return SK_OK;
        break;
      }
    }
  }
  if (batched && hs->done > 0 && hs->fix_hi > hs->fix_lo) {
    // the last retirement happened at slot 0 of a longer batch: outputs from that slot's sums (see k_interp_b)
    launch_fix_b(cfg->dc, db, hs->fix_lo, hs->fix_hi, hs->fix_conf, hs->fix_tail, st);
    HIPCHK(hipStreamSynchronize(st));
  }
  HIPCHK(hipGetLastError());
  stt.n_outer = hs->n_outer;
  stt.n_panels = hs->n_panels;
  stt.n_accepted = hs->n_accepted;
  stt.sum_active = hs->sum_active;
  stt.last_b = hs->b;
  stt.n_steps = batched ? hs->n_steps : hs->n_panels;
  stt.n_rerun = batched ? hs->n_rerun : 0;
  stt.sum_active_steps = batched ? hs->sum_active_steps : hs->sum_active;
  stt.n_fix = batched ? hs->n_fix : 0;
  if (batched) cfg->last_n_steps = hs->n_steps + hs->n_rerun;
  if (batched && cfg->k_adaptive && hs->done > 0) {
    if (cfg->dc.K == cfg->k_full) {
      cfg->narrow = hs->n_panels >= 2 && hs->n_multi == 0;
    } else if (hs->would_batch) {
      // this input batches at the full width: its result is the full-width march's (once; the config stays wide)
      cfg->narrow = false;
      const sk_stats keep = cfg->stats;
      std::memset(&cfg->stats, 0, sizeof(cfg->stats));
      cfg->stats.n_unique = keep.n_unique;
      cfg->ev_recs.clear();
      cfg->ev_used = 0;
      return run_core(cfg, d_uxs, n, k0, ks0, d_ks, d_errs, d_scratch6n, st, false, nullptr);
    }
  }
  cfg->trace_len = (size_t)hs->trace_len;
  cfg->trace_valid = false;
  if (cfg->profile) {
    HIPCHK(hipStreamSynchronize(st));
    std::memset(&cfg->ktimes, 0, sizeof(cfg->ktimes));
    // only the first steps did work; later (speculative) steps exit early
    const long long worked = batched ? hs->n_steps + hs->n_rerun : hs->n_panels;
    int64_t seen[SK_K_COUNT] = {0};
    for (auto& r : cfg->ev_recs) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, r.a, r.b);
      if (r.kid == SK_K_ADVANCE || cfg->dc.dim > 1 || seen[r.kid]++ < worked) {
        cfg->ktimes.total_ms[r.kid] += ms;
        cfg->ktimes.launches[r.kid] += 1;
      }
    }
    cfg->ktimes.active_items[SK_K_GEN] = 3 * cfg->M * hs->n_panels;
    cfg->ktimes.active_items[SK_K_SPREAD] = 2 * cfg->dc.n2 * hs->n_panels;
    cfg->ktimes.active_items[SK_K_FFT] = 2 * cfg->dc.n2 * hs->n_panels;
    cfg->ktimes.active_items[SK_K_INTERP] = hs->sum_active;
    cfg->ktimes.active_items[SK_K_ADVANCE] = hs->n_panels;
    cfg->ktimes.active_items[SK_K_NEAR] = hs->n_panels;
    cfg->ev_recs.clear();
    cfg->ev_used = 0;
  }
#ifdef SK_STAMPS
  if (const char* f = getenv("SK_STAMP_FILE")) {
    std::vector<unsigned long long> h(8 * 8192);
    (void)hipMemcpy(h.data(), db.stamps, h.size() * 8, hipMemcpyDeviceToHost);
    if (FILE* fp = fopen(f, "wb")) { fwrite(h.data(), 8, h.size(), fp); fclose(fp); }
    std::vector<unsigned long long> h2(16 * 4096);
    (void)hipMemcpy(h2.data(), db.stamps2, h2.size() * 8, hipMemcpyDeviceToHost);
    if (FILE* fp = fopen((std::string(f) + ".interp").c_str(), "wb")) { fwrite(h2.data(), 8, h2.size(), fp); fclose(fp); }
  }
#endif
  if (hs->done < 0) return fail(-hs->done, status_message(-hs->done));
  return SK_OK;
}

static int kernel_values_device_impl(sk_config* cfg, const double* d_xs, size_t n, double k0, double ks0, double* d_vals,
                                     double* d_errs, hipStream_t st, int order = SK_ORDER_UNKNOWN,
                                     hipEvent_t lags_ready = nullptr) {
  std::memset(&cfg->stats, 0, sizeof(cfg->stats));
  cfg->hyb_narrow = cfg->hyb_skips = 0;
  cfg->trace_len = 0;
  cfg->trace_valid = false;
  if (n == 0) return SK_OK;
  if (n > (size_t)0x7fffffff) return fail(SK_EINVAL, "at most 2^31-1 lags per call");
  HIPCHK(hipSetDevice(cfg->device));
  const size_t a8 = ((n * 8 + 255) / 256) * 256;
  // 6 n running / partial sums (+ 4 n raw panel sums of the logw passes); 9 n on the batched march
  const size_t nscr = cfg->dc.logw ? 10 : 9;
  int rc = ensure_ws(cfg, nscr * a8);
  if (rc) return rc;
  // Host-array entries have looked at the lags while staging them (sk_hostio.cpp) and say so in `order`.
  // Otherwise the input is checked for "strictly increasing, >= 0" on the device and the march starts
  // speculatively on it as is; the flags come back with the first status block (no extra sync).
  if (order == SK_ORDER_INVALID) return fail(SK_EINVAL, "lags must be non-negative and finite");
  cfg->stats.n_unique = (int64_t)n;
  int unsorted = 1, invalid = 0;
  // lags of unknown order whose predecessor (same count, same config) was unsorted: sort at once -- the check, the
  // speculative first step and the round trip for its flags are 50 us of nothing then; the sort reports what the check
  // would have (invalid lags; an input that was sorted after all, which ends the expectation)
  const bool sort_first = order == SK_ORDER_UNKNOWN && cfg->expect_unsorted && cfg->expect_n == (long long)n && !lags_ready;
  if (order != SK_ORDER_UNSORTED && !sort_first) {
    rc = run_core(cfg, d_xs, (long long)n, k0, ks0, d_vals, d_errs, (double*)cfg->ws, st, order == SK_ORDER_UNKNOWN,
                  lags_ready);
    lags_ready = nullptr;  // waited for in the stream
    if (order == SK_ORDER_SORTED) return rc;
    unsorted = cfg->h_status->flags[0];
    invalid = cfg->h_status->flags[1] & 1;
  }
  if (order == SK_ORDER_UNKNOWN && !sort_first) {
    cfg->expect_unsorted = unsorted && !invalid;
    cfg->expect_n = (long long)n;
  }
  if (order == SK_ORDER_UNKNOWN && !sort_first && (cfg->h_status->flags[1] & 2)) {
    (void)hipStreamSynchronize(st);
    (void)hipMemsetAsync(cfg->db.flags, 0, 2 * sizeof(int), st);
    return fail(SK_EHIP, "internal error: inconsistent device state detected by a kernel guard");
  }
  if (order == SK_ORDER_UNKNOWN && !sort_first && (unsorted || invalid)) {
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemsetAsync(cfg->db.flags, 0, 2 * sizeof(int), st));
    cfg->ev_recs.clear();
    cfg->ev_used = 0;
  }
  if (invalid) return fail(SK_EINVAL, "lags must be non-negative and finite");
  if (!unsorted) return rc;
  // src/adaptive.jl:99-120: unique + sort, then gather back
  if (lags_ready) HIPCHK(hipStreamWaitEvent(st, lags_ready, 0));
  std::memset(&cfg->stats, 0, sizeof(cfg->stats));
  const size_t tb = sort_unique_temp_bytes((long long)n);
  rc = ensure_ws(cfg, (4 + nscr) * a8 + tb + 256);
  if (rc) return rc;
  char* base = (char*)cfg->ws;
  double* uxs = (double*)base; base += a8;
  long long* inv = (long long*)base; base += a8;
  double* ks = (double*)base; base += a8;
  double* errs = (double*)base; base += a8;
  double* scr = (double*)base; base += nscr * a8;
  long long* d_nu = (long long*)base; base += 256;
  long long nu = 0;
  int was_unsorted = 1, bad_lags = 0;
  if (sort_unique_sync(d_xs, (long long)n, uxs, inv, d_nu, cfg->h_nuniq, base, tb, st, &nu, &was_unsorted, &bad_lags, true))
    return fail(SK_EHIP, "sort failed");
  cfg->stats.n_syncs++;
  if (sort_first) {
    if (bad_lags) {
      cfg->expect_unsorted = false;
      return fail(SK_EINVAL, "lags must be non-negative and finite");
    }
    if (!was_unsorted) cfg->expect_unsorted = false;  // sorted after all: ask first next time
  }
  cfg->stats.n_unique = nu;
  rc = run_core(cfg, uxs, nu, k0, ks0, ks, errs, scr, st, false);
  if (rc) return rc;
  launch_gather(ks, errs, inv, (long long)n, d_vals, d_errs, st);
  HIPCHK(hipStreamSynchronize(st));
  return SK_OK;
}

// grow-only device buffers
static int ensure_dev(void** buf, size_t* have, size_t bytes) {
  if (bytes <= *have) return SK_OK;
  if (*buf) HIPCHK(hipFree(*buf));
  *buf = nullptr;
  *have = 0;
  HIPCHK(hipMalloc(buf, bytes + bytes / 4));
  *have = bytes + bytes / 4;
  return SK_OK;
}

// Sort + dedup of host lags on the config's device (the sharded entry's unsorted input): staged upload, rocPRIM
// radix sort + scan as in kernel_values_device_impl, unique set and inverse map back to the host.
int sort_unique_on_device(sk_config* cfg, const double* xs, size_t n, std::vector<double>& uxs, std::vector<int64_t>& inv) {
  HIPCHK(hipSetDevice(cfg->device));
  hipStream_t st = cfg->own_stream;
  const size_t a8 = ((n * 8 + 255) / 256) * 256;
  const size_t tb = sort_unique_temp_bytes((long long)n);
  int rc = ensure_dev(&cfg->d_aux, &cfg->aux_bytes, 3 * a8 + tb + 512);
  if (rc) return rc;
  char* buf = (char*)cfg->d_aux;
  double* d_xs = (double*)buf;
  double* d_ux = (double*)(buf + a8);
  long long* d_inv = (long long*)(buf + 2 * a8);
  long long* d_nu = (long long*)(buf + 3 * a8);
  void* tmp = buf + 3 * a8 + 256;
  rc = h2d_doubles(cfg, d_xs, xs, n, st, nullptr);
  if (rc) return rc;
  long long nu_ll = 0;
  if (sort_unique_sync(d_xs, (long long)n, d_ux, d_inv, d_nu, cfg->h_nuniq, tmp, tb, st, &nu_ll, nullptr, nullptr, true))
    return fail(SK_EHIP, "sort failed");
  const size_t nu = (size_t)nu_ll;
  uxs.resize(nu);
  inv.resize(n);
  rc = d2h_doubles(cfg, uxs.data(), d_ux, nu, st);
  if (!rc) rc = d2h_doubles(cfg, (double*)inv.data(), (const double*)d_inv, n, st);  // 8-byte words
  return rc;
}

}  // namespace sk

// Independent marches at once: one host thread and the config's own stream per config.  A march in dim >= 2 is a chain of
// small launches with a host round trip per step that leaves most of the device idle (step_frac 0.02 on BASELINE config
// 4); the marches of the integrands of a Jacobian (src/derivatives.jl:51-112: one kernel_values call per integrand on the
// same lags) share nothing but their input, so they overlap instead of queueing.  Catalogue SDFs only (a host closure is
// called on the caller's thread, see sk_sdf_batch_fn), distinct configs; SK_MULTI_SERIAL=1 keeps the marches in sequence.
static bool marches_can_overlap(sk_config* const* cfgs, size_t ncfg) {
  if (ncfg < 2 || getenv("SK_MULTI_SERIAL") != nullptr) return false;
  for (size_t c = 0; c < ncfg; ++c) {
    if (cfgs[c]->args.sdf_kind == SK_SDF_CALLBACK) return false;
    for (size_t d = 0; d < c; ++d)
      if (cfgs[d] == cfgs[c]) return false;
  }
  return true;
}
// runs work(c) for every config, on threads of their own when the marches can overlap; first failure wins
template <class Work>
static int for_each_march(sk_config* const* cfgs, size_t ncfg, Work&& work) {
  if (!marches_can_overlap(cfgs, ncfg)) {
    for (size_t c = 0; c < ncfg; ++c) {
      const int rc = work(c);
      if (rc) return rc;
    }
    return SK_OK;
  }
  std::vector<int> rcs(ncfg, SK_OK);
  std::vector<std::string> msgs(ncfg);
  auto run = [&](size_t c) {
    if (hipSetDevice(cfgs[c]->device) != hipSuccess) {
      rcs[c] = SK_EHIP;
      msgs[c] = "hipSetDevice failed";
      return;
    }
    rcs[c] = work(c);
    if (rcs[c]) msgs[c] = sk_last_error();  // the error channel is thread-local
  };
  // one OpenMP thread per march: the runtime keeps its team between calls (fresh std::threads would each build -- and at
  // their exit tear down -- an OpenMP team of their own for the staging copies inside: 9 ms per call measured instead of
  // 2); the staging passes inside run on the march's own thread (nested parallelism is off by default)
#pragma omp parallel for num_threads((int)ncfg) schedule(static, 1)
  for (long long c = 0; c < (long long)ncfg; ++c) run((size_t)c);
  for (size_t c = 0; c < ncfg; ++c)
    if (rcs[c]) return fail(rcs[c], msgs[c]);
  return SK_OK;
}


// ====================================================================== C ABI

extern "C" {

const char* sk_last_error(void) { return g_err.c_str(); }
const char* sk_version(void) { return "spectralkernels_hip 0.1.0 (gfx950)"; }

int sk_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
  return std::strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
}

void sk_config_args_default(sk_config_args* a) {
  std::memset(a, 0, sizeof(*a));
  a->sdf_kind = SK_SDF_MATERN;
  a->sdf_params[0] = 1.0;
  a->sdf_params[1] = 1.0;
  a->sdf_params[2] = 1.5;
  a->sdf_params[3] = 1.0;
  a->dim = 1;
  a->alpha = 0.0;
  a->tol = 1e-8;
  a->criteria = SK_CRIT_BOTH;
  a->m = 4096;
  a->k = 16;
  a->device = -1;
}

int sk_config_create(sk_config** out, const sk_config_args* args) {
  if (!out || !args) return fail(SK_EINVAL, "null argument");
  *out = nullptr;
  int64_t m, k;
  int warned;
  int rc = validate(*args, m, k, warned);
  if (rc) return rc;
  if (args->logw) {
    const double p0 = -args->alpha + (args->dim == 1 ? 0.0 : args->dim / 2.0) + (args->derivative ? 1.0 : 0.0);
    if (p0 != 0.0 && args->dim > 2)  // src/quadrature.jl:221
      return fail(SK_EUNSUPPORTED, "singularity derivative not implemented in d > 2");
    if (args->derivative)  // the integration-by-parts panel (src/quadrature.jl:186-228) ignores `derivative`
      return fail(SK_EUNSUPPORTED, "logw=true together with derivative=true is not supported");
    if (p0 != 0.0 && args->sdf_kind == SK_SDF_CALLBACK && !args->df)
      return fail(SK_EINVAL, "logw=true needs the SDF derivative df (src/quadrature.jl:194)");
  }
  if (args->dim != 1 && (args->dim % 2 != 0))  // Int64(dim/2-1) is inexact for odd dim (src/quadrature.jl:138,179)
    return fail(SK_EINVAL, "dim must be 1 or even");
  if (!sk_device_available())
    return fail(SK_ENODEVICE, "no gfx950 device available: the HIP path has no CPU fallback");
  int dev = args->device;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) return fail(SK_EHIP, "hipGetDevice failed");
  }
  HIPCHK(hipSetDevice(dev));
  {
    // eager module load on an idle device, once per device (see k_module_warmup)
    static std::mutex warm_mu;
    static unsigned long long warmed = 0ull;
    std::lock_guard<std::mutex> lk(warm_mu);
    if (dev < 64 && !(warmed >> dev & 1ull)) {
      (void)hipDeviceSynchronize();
      launch_module_warmup(nullptr);
      (void)hipDeviceSynchronize();
      warmed |= 1ull << dev;
    }
  }
  sk_config* cfg = new sk_config();
  cfg->args = *args;
  cfg->m = m;
  cfg->k = k;
  cfg->M = m * k;
  cfg->device = dev;
  cfg->warned = warned;
  cfg->p = -args->alpha + (args->dim == 1 ? 0.0 : args->dim / 2.0) + (args->derivative ? 1.0 : 0.0);
  cfg->c = (args->dim == 1 ? 2.0 : 2.0 * M_PI) * (args->derivative ? -2.0 * M_PI : 1.0) * (args->logw ? -1.0 : 1.0);
  cfg->sdf.kind = args->sdf_kind;
  std::memcpy(cfg->sdf.prm, args->sdf_params, sizeof(cfg->sdf.prm));
  cfg->sdf.prm[7] = (double)args->sdf_dparam;
  cfg->sdf.f = args->f;
  cfg->sdf.ctx = args->f_ctx;
  cfg->dsdf = cfg->sdf;  // only ever called for callback SDFs (the catalogue's derivative lives on the device)
  cfg->dsdf.f = args->df;
  cfg->dsdf.ctx = args->df_ctx;
  std::memset(&cfg->stats, 0, sizeof(cfg->stats));
  // rules (src/adaptive.jl:47-49)
  gauss_rule(m, 0.0, cfg->leg[0], cfg->leg[1]);
  gauss_rule(2 * m, 0.0, cfg->leg[2], cfg->leg[3]);
  if (cfg->p != 0.0) {
    gauss_rule(m, cfg->p, cfg->jac[0], cfg->jac[1]);
    gauss_rule(2 * m, cfg->p, cfg->jac[2], cfg->jac[3]);
    if (cfg->jac[0].empty() || cfg->jac[2].empty()) {
      delete cfg;
      return fail(SK_EINVAL, "Gauss-Jacobi rule generation failed");
    }
  } else {
    for (int i = 0; i < 4; ++i) cfg->jac[i] = cfg->leg[i];
  }
  // NUFFT geometry
  DevConst& dc = cfg->dc;
  std::memset(&dc, 0, sizeof(dc));
  dc.m = m; dc.k = k; dc.M = cfg->M;
  dc.dim = args->dim; dc.derivative = args->derivative; dc.crit0 = args->criteria; dc.has_tail = args->has_tail; dc.logw = args->logw;
  dc.alpha = args->alpha; dc.p = cfg->p; dc.c = cfg->c; dc.tol = args->tol; dc.tail = args->tail;
  dc.sdf_kind = args->sdf_kind;
  std::memcpy(dc.sdf_prm, args->sdf_params, sizeof(dc.sdf_prm));
  dc.sdf_prm[7] = (double)args->sdf_dparam;
  if (dc.sdf_kind == SK_SDF_MATERN) {  // device fast path for half-integer exponents (sdf_eval_dev)
    const double h = 2.0 * (dc.sdf_prm[2] + 0.5 * dc.sdf_prm[3]);
    dc.sdf_prm[4] = (h == std::floor(h) && h >= 1.0 && h <= 64.0) ? h : 0.0;
  }
  dc.w = choose_width(*args);
  {
    int64_t need = (int64_t)std::ceil(0.99 * (double)cfg->M) + dc.w;
    int64_t n1 = cfg->M >= need && (cfg->M % 2 == 0) ? cfg->M : need;
    if (n1 < 4 * dc.w) n1 = 4 * dc.w;
    dc.n1 = next_smooth_even(n1);
    dc.n2 = 2 * dc.n1;
  }
  // Panels per batched step: the transforms of a step (4 K of length n1) stay at what one default-rule panel pair
  // costs today (K n1 <= 2^17): two panels per step for the default rule, four for the small rules
  dc.K = args->batch_panels ? args->batch_panels : (dc.n1 > 40000 ? 2 : 4);
  if (const char* g = getenv("SK_BATCH_GUESS")) dc.guess_mode = !strcmp(g, "always") ? 1 : (!strcmp(g, "never") ? 2 : 0);
  dc.delta = 2.0 / (double)(dc.n1 - dc.w);
  dc.tau_fac = (double)dc.n2 / (double)(dc.n1 - dc.w);
  dc.log2m = -1;
  if ((m & (m - 1)) == 0 && 2 * cfg->M * 2 < (int64_t)1 << 31) {
    dc.log2m = 0;
    while (((int64_t)1 << dc.log2m) < m) ++dc.log2m;
  }
  dc.inv_k = ((k & (k - 1)) == 0) ? 1.0 / (double)k : 0.0;
  dc.theta_fac = 2.0 * M_PI / (double)(dc.n1 - dc.w);
  dc.theta_max = 1.001 * (M_PI / 2.0) * (double)cfg->M / (double)(dc.n1 - dc.w);
  dc.inv_theta_max = 1.0 / dc.theta_max;
  build_es_tables(dc.w, dc.n1, dc.n2, dc.theta_max, cfg->tab);
  dc.nco = cfg->tab.nco;
  dc.ncheb = cfg->tab.ncheb;
#define UP(vec, field)                                     \
  do {                                                     \
    rc = upload(cfg, vec, &field);                         \
    if (rc) { sk_config_destroy(cfg); return rc; }         \
  } while (0)
  UP(cfg->leg[0], dc.leg_no1); UP(cfg->leg[1], dc.leg_wt1); UP(cfg->leg[2], dc.leg_no2); UP(cfg->leg[3], dc.leg_wt2);
  UP(cfg->jac[0], dc.jac_no1); UP(cfg->jac[1], dc.jac_wt1); UP(cfg->jac[2], dc.jac_no2); UP(cfg->jac[3], dc.jac_wt2);
  UP(cfg->tab.deconv, dc.deconv);
  {
    std::vector<double2> otw((size_t)(dc.n1 / 2 + 1));
    for (int64_t l = 0; l <= dc.n1 / 2; ++l) {
      const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)l / (long double)dc.n2;
      otw[(size_t)l] = make_double2((double)cosl(a), (double)sinl(a));
    }
    UP(otw, dc.otw);
  }
  std::memset(&cfg->ec, 0, sizeof(cfg->ec));
  std::memcpy(cfg->ec.poly, cfg->tab.poly.data(), cfg->tab.poly.size() * sizeof(double));
  std::memcpy(cfg->ec.cheb, cfg->tab.cheb.data(), cfg->tab.cheb.size() * sizeof(double));
  {
    std::vector<EsConst> v(1, cfg->ec);
    const EsConst* dptr = nullptr;
    UP(v, dptr);
    cfg->d_ec = const_cast<EsConst*>(dptr);
  }
  int cus = 256;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && v > 0) cus = v;
  }
  for (int variant = 0; variant < 2; ++variant) {
    if (variant == 1 && cfg->p == 0.0) {
      for (int rule = 0; rule < 2; ++rule) {
        dc.pos[1][rule] = dc.pos[0][rule];
        dc.cellstart[1][rule] = dc.cellstart[0][rule];
      }
      dc.tiles[1] = dc.tiles[0];
      dc.ntiles[1] = dc.ntiles[0];
      continue;
    }
    std::vector<int> cs2[2];
    for (int rule = 0; rule < 2; ++rule) {
      std::vector<double> pos;
      build_positions(cfg, variant, rule, pos, cs2[rule]);
      UP(pos, dc.pos[variant][rule]);
      UP(cs2[rule], dc.cellstart[variant][rule]);
    }
    std::vector<int4> tiles;
    build_tiles(cfg, cs2, 4 * cus, tiles);
    UP(tiles, dc.tiles[variant]);
    dc.ntiles[variant] = (int)tiles.size();
  }
#undef UP
  DevBuffers& db = cfg->db;
  std::memset(&db, 0, sizeof(db));
#define DA(count, field)                                   \
  do {                                                     \
    rc = dalloc(cfg, count, &field);                       \
    if (rc) { sk_config_destroy(cfg); return rc; }         \
  } while (0)
  DA(2, db.state); DA(kStackCap, db.stack); DA(kTraceCap, db.trace);
  DA((size_t)3 * cfg->M, db.src); DA((size_t)2 * dc.n2, db.fft); DA((size_t)2 * dc.n2, db.fft_out);
  DA((size_t)3 * cfg->M, cfg->d_fvals);
  DA(2, db.flags); DA(2, db.red); DA(2, db.redb);
  DA((size_t)4 * dc.K * dc.n1, db.bfft); DA((size_t)4 * dc.K * dc.n1, db.bfft_out);
  (void)hipMemset(db.flags, 0, 2 * sizeof(int));
#ifdef SK_STAMPS
  DA(8 * 8192, db.stamps);
  DA(16 * 4096, db.stamps2);
  (void)hipMemset(db.stamps2, 0, 16 * 4096 * 8);
#endif
#undef DA
  if (hipHostMalloc((void**)&cfg->h_status, sizeof(HostStatus), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&db.host_status, cfg->h_status, 0) != hipSuccess ||
      hipHostMalloc((void**)&cfg->h_nuniq, 4 * sizeof(long long)) != hipSuccess ||
      hipHostMalloc((void**)&cfg->h_scal, 32 * sizeof(double)) != hipSuccess) {
    sk_config_destroy(cfg);
    return fail(SK_EHIP, "hipHostMalloc failed");
  }
  std::memset(cfg->h_status, 0, sizeof(HostStatus));
  if (hipStreamCreate(&cfg->own_stream) != hipSuccess || hipStreamCreate(&cfg->copy_stream) != hipSuccess ||
      hipEventCreateWithFlags(&cfg->up_event, hipEventDisableTiming) != hipSuccess) {
    sk_config_destroy(cfg);
    return fail(SK_EHIP, "hipStreamCreate failed");
  }
  // rocFFT plans, e^{+i...}
  std::call_once(g_fft_once, [] { rocfft_setup(); });
  // four length-n1 transforms (even / odd output samples of both rules, see k_spread) instead of two of length n2
  size_t len[1] = {(size_t)dc.n1};
  rocfft_status fs = rocfft_plan_create(&cfg->plan, rocfft_placement_notinplace, rocfft_transform_type_complex_inverse,
                                        rocfft_precision_double, 1, len, 4, nullptr);
  if (fs != rocfft_status_success) {
    sk_config_destroy(cfg);
    return fail(SK_EHIP, "rocfft_plan_create failed");
  }
  {
    // batched march: even / odd output samples of both rules of every slot
    if (rocfft_plan_create(&cfg->plan_b, rocfft_placement_notinplace, rocfft_transform_type_complex_inverse,
                           rocfft_precision_double, 1, len, (size_t)4 * dc.K, nullptr) != rocfft_status_success) {
      sk_config_destroy(cfg);
      return fail(SK_EHIP, "rocfft_plan_create (batched march) failed");
    }
    size_t wbb = 0;
    rocfft_plan_get_work_buffer_size(cfg->plan_b, &wbb);
    rocfft_execution_info_create(&cfg->info_b);
    if (wbb) {
      if (hipMalloc(&cfg->fft_work_b, wbb) != hipSuccess) {
        sk_config_destroy(cfg);
        return fail(SK_EHIP, "hipMalloc(fft work) failed");
      }
      rocfft_execution_info_set_work_buffer(cfg->info_b, cfg->fft_work_b, wbb);
    }
  }
  cfg->k_full = dc.K;
  if (args->batch_panels == 0 && dc.K > 1 && !std::getenv("SK_K_FIXED")) {
    if (rocfft_plan_create(&cfg->plan_b1, rocfft_placement_notinplace, rocfft_transform_type_complex_inverse,
                           rocfft_precision_double, 1, len, (size_t)4, nullptr) != rocfft_status_success) {
      sk_config_destroy(cfg);
      return fail(SK_EHIP, "rocfft_plan_create (one-panel march) failed");
    }
    size_t wb1 = 0;
    rocfft_plan_get_work_buffer_size(cfg->plan_b1, &wb1);
    rocfft_execution_info_create(&cfg->info_b1);
    if (wb1) {
      if (hipMalloc(&cfg->fft_work_b1, wb1) != hipSuccess) {
        sk_config_destroy(cfg);
        return fail(SK_EHIP, "hipMalloc(fft work) failed");
      }
      rocfft_execution_info_set_work_buffer(cfg->info_b1, cfg->fft_work_b1, wb1);
    }
    cfg->k_adaptive = true;
  }
  {
    // the finufft1d3-level seam (sk_panel_sums) keeps one plain length-n2 transform, in place
    size_t lens[1] = {(size_t)dc.n2};
    if (rocfft_plan_create(&cfg->plan_seam, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                           rocfft_precision_double, 1, lens, 1, nullptr) != rocfft_status_success) {
      sk_config_destroy(cfg);
      return fail(SK_EHIP, "rocfft_plan_create (seam) failed");
    }
    size_t wbs = 0;
    rocfft_plan_get_work_buffer_size(cfg->plan_seam, &wbs);
    rocfft_execution_info_create(&cfg->info_seam);
    if (wbs) {
      if (hipMalloc(&cfg->fft_work_seam, wbs) != hipSuccess) {
        sk_config_destroy(cfg);
        return fail(SK_EHIP, "hipMalloc(fft work) failed");
      }
      rocfft_execution_info_set_work_buffer(cfg->info_seam, cfg->fft_work_seam, wbs);
    }
  }
  size_t wb = 0;
  rocfft_plan_get_work_buffer_size(cfg->plan, &wb);
  rocfft_execution_info_create(&cfg->info);
  if (wb) {
    if (hipMalloc(&cfg->fft_work, wb) != hipSuccess) {
      sk_config_destroy(cfg);
      return fail(SK_EHIP, "hipMalloc(fft work) failed");
    }
    rocfft_execution_info_set_work_buffer(cfg->info, cfg->fft_work, wb);
  }
  if (args->dim > 1) {
    // Hankel path: batched plan for the K asymptotic terms x 2 rules, buffers, a_k(nu)
    size_t len2[1] = {(size_t)dc.n2};
    if (rocfft_plan_create(&cfg->plan_hk, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                           rocfft_precision_double, 1, len2, 2 * hk_terms(dc.w), nullptr) != rocfft_status_success) {
      sk_config_destroy(cfg);
      return fail(SK_EHIP, "rocfft_plan_create (Hankel batch) failed");
    }
    size_t wb2 = 0;
    rocfft_plan_get_work_buffer_size(cfg->plan_hk, &wb2);
    rocfft_execution_info_create(&cfg->info_hk);
    if (wb2) {
      if (hipMalloc(&cfg->fft_work_hk, wb2) != hipSuccess) {
        sk_config_destroy(cfg);
        return fail(SK_EHIP, "hipMalloc(fft work) failed");
      }
      rocfft_execution_info_set_work_buffer(cfg->info_hk, cfg->fft_work_hk, wb2);
    }
    void* p2 = nullptr; void* p3 = nullptr; void* p4 = nullptr;
    if (hipMalloc(&p2, sizeof(double2) * (size_t)kHkTerms * 2 * dc.n2) != hipSuccess ||
        hipMalloc(&p3, sizeof(double) * (size_t)(kHkMaxLevels + 1) * 2 * kHkNodes * kHkSplit) != hipSuccess ||
        hipMalloc(&p4, sizeof(double) * (size_t)(kHkMaxLevels + 1) * 2 * kHkNodes) != hipSuccess) {
      sk_config_destroy(cfg);
      return fail(SK_EHIP, "hipMalloc(Hankel buffers) failed");
    }
    cfg->owned.push_back(p2); cfg->owned.push_back(p3); cfg->owned.push_back(p4);
    db.hk_fft = (double2*)p2; db.hk_nodes = (double*)p3; db.hk_cheb = (double*)p4;
    {
      // exact-cosine DCT table of the kHkNodes-point Chebyshev transforms: cos(pi j / (2 N)) with j = m (2 i + 1) mod 4 N
      // in integer arithmetic, then the node abscissae cos(pi (2 i + 1) / (2 N))
      std::vector<double> tab((size_t)kHkNodes * kHkNodes + kHkNodes);
      auto ctab = [](int j) {
        j %= 4 * kHkNodes;
        if (j > 2 * kHkNodes) j = 4 * kHkNodes - j;
        return j == kHkNodes ? 0.0 : std::cos(M_PI * (double)j / (2.0 * kHkNodes));
      };
      for (int i = 0; i < kHkNodes; ++i)
        for (int mm = 0; mm < kHkNodes; ++mm) tab[(size_t)i * kHkNodes + mm] = ctab(mm * (2 * i + 1));
      for (int i = 0; i < kHkNodes; ++i) tab[(size_t)kHkNodes * kHkNodes + i] = ctab(2 * i + 1);
      void* p5 = nullptr;
      if (hipMalloc(&p5, tab.size() * sizeof(double)) != hipSuccess ||
          hipMemcpy(p5, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        sk_config_destroy(cfg);
        return fail(SK_EHIP, "hipMalloc(Hankel tables) failed");
      }
      cfg->owned.push_back(p5);
      db.hk_dct = (double*)p5;
    }
    if (hipStreamCreateWithFlags(&cfg->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&cfg->side_stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&cfg->fork_event, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&cfg->join_event, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&cfg->join_event2, hipEventDisableTiming) != hipSuccess) {
      sk_config_destroy(cfg);
      return fail(SK_EHIP, "hipStreamCreate (Hankel side stream) failed");
    }
    if (!std::getenv("SK_HK_LEVELS")) {
      // inner box (sk_hankel_inner.inc): partial moments of every source wave; the bands are chosen per step
      cfg->hk_nb = 1;
      cfg->hk_ngmax = kHkNgCap;
      void* p6 = nullptr;
      const size_t npart = (size_t)(3 * cfg->M / 256 + 2 * cfg->k + 2), nmom = (size_t)2 * cfg->k;
      if (hipMalloc(&p6, sizeof(double) * (npart + nmom) * cfg->hk_ngmax) != hipSuccess) {
        sk_config_destroy(cfg);
        return fail(SK_EHIP, "hipMalloc(Hankel moments) failed");
      }
      cfg->owned.push_back(p6);
      db.hk_mpart = (double*)p6;
      db.hk_mom = (double*)p6 + npart * cfg->hk_ngmax;
    }
    const int nu = args->derivative ? args->dim / 2 : args->dim / 2 - 1;
    cfg->hk_a[0] = 1.0;
    for (int kq = 1; kq < kHkTerms; ++kq)
      cfg->hk_a[kq] = cfg->hk_a[kq - 1] * (4.0 * nu * nu - (2.0 * kq - 1.0) * (2.0 * kq - 1.0)) / (8.0 * kq);
    const double nu2 = dc.dim / 2;
    cfg->hk_a2[0] = 1.0;
    for (int kq = 1; kq < kHkTerms; ++kq)
      cfg->hk_a2[kq] = cfg->hk_a2[kq - 1] * (4.0 * nu2 * nu2 - (2.0 * kq - 1.0) * (2.0 * kq - 1.0)) / (8.0 * kq);
  }
  *out = cfg;
  if (warned) {
    set_error(
        "Tolerances < 1e-12 are not recommended. Switching to a smaller quadrature rule for higher accuracy (but "
        "slower) computations.");
    return SK_WARN_QUADSPEC;
  }
  return SK_OK;
}

void sk_config_destroy(sk_config* cfg) {
  if (!cfg) return;
  hipSetDevice(cfg->device);
  if (cfg->plan) rocfft_plan_destroy(cfg->plan);
  if (cfg->plan_b1) rocfft_plan_destroy(cfg->plan_b1);
  if (cfg->info_b1) rocfft_execution_info_destroy(cfg->info_b1);
  if (cfg->fft_work_b1) hipFree(cfg->fft_work_b1);
  if (cfg->plan_b) rocfft_plan_destroy(cfg->plan_b);
  if (cfg->info_b) rocfft_execution_info_destroy(cfg->info_b);
  if (cfg->fft_work_b) hipFree(cfg->fft_work_b);
  if (cfg->plan_seam) rocfft_plan_destroy(cfg->plan_seam);
  if (cfg->info_seam) rocfft_execution_info_destroy(cfg->info_seam);
  if (cfg->fft_work_seam) hipFree(cfg->fft_work_seam);
  for (int q = 0; q <= kHkTerms; ++q) {
    if (cfg->plan_hk_nt[q]) rocfft_plan_destroy(cfg->plan_hk_nt[q]);
    if (cfg->info_hk_nt[q]) rocfft_execution_info_destroy(cfg->info_hk_nt[q]);
    if (cfg->fft_work_hk_nt[q]) hipFree(cfg->fft_work_hk_nt[q]);
  }
  if (cfg->plan_hk) rocfft_plan_destroy(cfg->plan_hk);
  if (cfg->info_hk) rocfft_execution_info_destroy(cfg->info_hk);
  if (cfg->fft_work_hk) hipFree(cfg->fft_work_hk);
  if (cfg->info) rocfft_execution_info_destroy(cfg->info);
  if (cfg->fft_work) hipFree(cfg->fft_work);
  for (void* p : cfg->owned) hipFree(p);
  if (cfg->ws) hipFree(cfg->ws);
  if (cfg->d_io) hipFree(cfg->d_io);
  if (cfg->h_in) hipHostFree(cfg->h_in);
  if (cfg->h_out) hipHostFree(cfg->h_out);
  if (cfg->d_aux) hipFree(cfg->d_aux);
  if (cfg->d_joint) hipFree(cfg->d_joint);
  if (cfg->hk_qbuf) hipFree(cfg->hk_qbuf);
  if (cfg->d_jfv) hipFree(cfg->d_jfv);
  if (cfg->h_jfv) hipHostFree(cfg->h_jfv);
  for (int q = 0; q <= kJointTr; ++q) {
    if (cfg->plan_joint[q]) rocfft_plan_destroy(cfg->plan_joint[q]);
    if (cfg->info_joint[q]) rocfft_execution_info_destroy(cfg->info_joint[q]);
    if (cfg->fft_work_joint[q]) hipFree(cfg->fft_work_joint[q]);
  }
  for (hipEvent_t e : cfg->cp_events) hipEventDestroy(e);
  if (cfg->d_lags) hipFree(cfg->d_lags);
  if (cfg->d_plags) hipFree(cfg->d_plags);
  if (cfg->h_status) hipHostFree(cfg->h_status);
  if (cfg->h_nuniq) hipHostFree(cfg->h_nuniq);
  if (cfg->h_scal) hipHostFree(cfg->h_scal);
  for (double* q : cfg->h_fvpin)
    if (q) hipHostFree(q);
  if (cfg->own_stream) hipStreamDestroy(cfg->own_stream);
  if (cfg->copy_stream) hipStreamDestroy(cfg->copy_stream);
  if (cfg->side_stream) hipStreamDestroy(cfg->side_stream);
  if (cfg->side_stream2) hipStreamDestroy(cfg->side_stream2);
  if (cfg->join_event2) hipEventDestroy(cfg->join_event2);
  if (cfg->fork_event) hipEventDestroy(cfg->fork_event);
  if (cfg->join_event) hipEventDestroy(cfg->join_event);
  if (cfg->up_event) hipEventDestroy(cfg->up_event);
  for (hipEvent_t e : cfg->ev_pool) hipEventDestroy(e);
  delete cfg;
}

int sk_config_info(const sk_config* cfg, double* p, double* c, int64_t* m, int64_t* k, int* nufft_width) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  if (p) *p = cfg->p;
  if (c) *c = cfg->c;
  if (m) *m = cfg->m;
  if (k) *k = cfg->k;
  if (nufft_width) *nufft_width = cfg->dc.w;
  return SK_OK;
}

int sk_config_rules(const sk_config* cfg, int which, double* no1, double* wt1, double* no2, double* wt2) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  const std::vector<double>* r = which ? cfg->jac : cfg->leg;
  if (no1) std::memcpy(no1, r[0].data(), r[0].size() * 8);
  if (wt1) std::memcpy(wt1, r[1].data(), r[1].size() * 8);
  if (no2) std::memcpy(no2, r[2].data(), r[2].size() * 8);
  if (wt2) std::memcpy(wt2, r[3].data(), r[3].size() * 8);
  return SK_OK;
}

int sk_compute_k0(sk_config* cfg, double* k0) {
  if (!cfg || !k0) return fail(SK_EINVAL, "null argument");
  *k0 = compute_k0_host(cfg->sdf, cfg->args.dim, cfg->p, cfg->c, cfg->args.derivative, cfg->args.logw, cfg->args.tol);
  return SK_OK;
}

static int kernel_values_device_order(sk_config* cfg, const double* d_xs, size_t n, double k0, int param_derivative,
                                      double* d_vals, double* d_errs, hipStream_t st, int order,
                                      hipEvent_t lags_ready = nullptr) {
  double ks0 = k0;  // src/adaptive.jl:134-145
  if (cfg->args.derivative) ks0 = 0.0;
  else if (param_derivative) {
    int rc = sk_compute_k0(cfg, &ks0);
    if (rc) return rc;
  }
  return kernel_values_device_impl(cfg, d_xs, n, k0, ks0, d_vals, d_errs, st, order, lags_ready);
}

int sk_kernel_values_device(sk_config* cfg, const double* d_xs, size_t n, double k0, int param_derivative, double* d_vals,
                            double* d_errs, void* stream) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  hipStream_t st = stream ? (hipStream_t)stream : cfg->own_stream;
  return kernel_values_device_order(cfg, d_xs, n, k0, param_derivative, d_vals, d_errs, st, SK_ORDER_UNKNOWN);
}

static double ms_since(const std::chrono::steady_clock::time_point& t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int sk_kernel_values(sk_config* cfg, const double* xs, size_t n, double k0, int has_k0, int param_derivative,
                     double* vals_out, double* errs_out) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  if (!has_k0) {
    int rc = sk_compute_k0(cfg, &k0);
    if (rc) return rc;
  }
  if (n == 0) return SK_OK;
  RoctxRange range("kernel_values");
  HIPCHK(hipSetDevice(cfg->device));
  const auto t0 = std::chrono::steady_clock::now();
  // grow-only device staging buffer of the host-array entry (lags, values, errors)
  int rc = ensure_dev((void**)&cfg->d_io, &cfg->io_bytes, 3 * n * sizeof(double));
  if (rc) return rc;
  double* d = cfg->d_io;
  hipStream_t st = cfg->own_stream;
  // The bulk upload runs on the copy stream beside the first step's spreading and FFT, which only need the
  // largest lag (and whether the smallest is zero): those two go ahead on the compute stream.
  int order = SK_ORDER_UNKNOWN;
  hipEvent_t ready = nullptr;
  if (n >= 65536) {
    cfg->h_scal[1] = xs[0];
    cfg->h_scal[2] = xs[n - 1];
    HIPCHK(hipMemcpyAsync(d, &cfg->h_scal[1], sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d + n - 1, &cfg->h_scal[2], sizeof(double), hipMemcpyHostToDevice, st));
    rc = h2d_doubles(cfg, d, xs, n, cfg->copy_stream, &order);
    if (!rc && hipEventRecord(cfg->up_event, cfg->copy_stream) != hipSuccess) rc = fail(SK_EHIP, "hipEventRecord failed");
    ready = cfg->up_event;
  } else {
    rc = h2d_doubles(cfg, d, xs, n, st, &order);
  }
  if (rc) {
    (void)hipStreamSynchronize(cfg->copy_stream);
    return rc;
  }
  const double t_up = ms_since(t0);
  rc = kernel_values_device_order(cfg, d, n, k0, param_derivative, d + n, d + 2 * n, st, order, ready);
  if (rc && ready) (void)hipStreamSynchronize(cfg->copy_stream);
  const double t_march = ms_since(t0);
  if (!rc) rc = d2h_doubles2(cfg, vals_out, errs_out, d + n, n, st);
  const double t_all = ms_since(t0);
  cfg->host_timing[0] = t_up;
  cfg->host_timing[1] = t_march - t_up;
  cfg->host_timing[2] = t_all - t_march;
  cfg->host_timing[3] = t_all;
  return rc;
}

int sk_host_threads(int cap) { return host_threads(cap); }

int sk_last_step_kinds(const sk_config* cfg, long long* out2) {
  if (!cfg || !out2) return fail(SK_EINVAL, "null argument");
  out2[0] = cfg->hyb_narrow;
  out2[1] = cfg->hyb_skips;
  return SK_OK;
}

int sk_last_host_timing(const sk_config* cfg, double* ms4) {
  if (!cfg || !ms4) return fail(SK_EINVAL, "null argument");
  for (int i = 0; i < 4; ++i) ms4[i] = cfg->host_timing[i];
  return SK_OK;
}

int sk_kernel_values_multi(sk_config* const* cfgs, size_t ncfg, const double* xs, size_t n, const double* k0s,
                           const int* param_derivative, double* vals_out, double* errs_out) {
  if (!cfgs || ncfg == 0 || !cfgs[0]) return fail(SK_EINVAL, "null config");
  for (size_t c = 0; c < ncfg; ++c)
    if (!cfgs[c] || cfgs[c]->device != cfgs[0]->device) return fail(SK_EINVAL, "configs must live on one device");
  if (n == 0) return SK_OK;
  if (n > (size_t)0x7fffffff) return fail(SK_EINVAL, "at most 2^31-1 lags per call");
  sk_config* c0 = cfgs[0];
  HIPCHK(hipSetDevice(c0->device));
  hipStream_t st = c0->own_stream;
  // device buffers of this call (grow-only, owned by the first config): lags, sorted unique lags, inverse map, sort
  // scratch; per config its unique results and its gathered results (values and errors adjacent)
  const size_t a8 = ((n * 8 + 255) / 256) * 256;
  const size_t tb = sort_unique_temp_bytes((long long)n);
  int rc = ensure_dev(&c0->d_aux, &c0->aux_bytes, (3 + 4 * ncfg) * a8 + tb + 512);
  if (rc) return rc;
  char* buf = (char*)c0->d_aux;
  double* d_xs = (double*)buf;
  double* uxs = (double*)(buf + a8);
  long long* inv = (long long*)(buf + 2 * a8);
  char* per = buf + 3 * a8;  // config c: ks, errs, then gathered values and (n doubles later) gathered errors
  long long* d_nu = (long long*)(buf + (3 + 4 * ncfg) * a8);
  void* tmp = (char*)d_nu + 256;
  int order = SK_ORDER_UNKNOWN;
  rc = h2d_doubles(c0, d_xs, xs, n, st, &order);
  if (rc) return rc;
  if (order == SK_ORDER_INVALID) return fail(SK_EINVAL, "lags must be non-negative and finite");
  // src/adaptive.jl:99-120 once for all configs: unique + sort (skipped for strictly increasing input; pinned
  // input of unknown order is sorted, which is harmless)
  long long nu = (long long)n;
  const bool sorted = (order == SK_ORDER_SORTED);
  if (!sorted) {
    if (sort_unique_sync(d_xs, (long long)n, uxs, inv, d_nu, c0->h_nuniq, tmp, tb, st, &nu)) return fail(SK_EHIP, "sort failed");
  } else {
    HIPCHK(hipStreamSynchronize(st));  // the marches below run on the configs' own streams
  }
  return for_each_march(cfgs, ncfg, [&](size_t c) -> int {
    sk_config* cf = cfgs[c];
    hipStream_t sc = cf->own_stream;
    double* ks = (double*)(per + (4 * c) * a8);
    double* errs = (double*)(per + (4 * c + 1) * a8);
    double* gk = (double*)(per + (4 * c + 2) * a8);
    double* ge = gk + n;
    const int pd = param_derivative ? param_derivative[c] : 0;
    int r;
    if (sorted) {
      r = kernel_values_device_order(cf, d_xs, n, k0s[c], pd, gk, ge, sc, SK_ORDER_SORTED);
    } else {
      // (sorted and unique by construction; the device check of the first march still catches invalid lags)
      r = kernel_values_device_order(cf, uxs, (size_t)nu, k0s[c], pd, ks, errs, sc,
                                     order == SK_ORDER_UNSORTED ? SK_ORDER_SORTED : SK_ORDER_UNKNOWN);
      if (!r) launch_gather(ks, errs, inv, (long long)n, gk, ge, sc);
    }
    if (!r) r = d2h_doubles2(cf, vals_out + c * n, errs_out + c * n, gk, n, sc);
    return r;
  });
}

// grow-only buffer of 3 n doubles: lags, values, errors of the pair entries
static int ensure_lags(sk_config* cfg, size_t n) {
  const size_t need = 3 * n * sizeof(double);
  if (need <= cfg->lags_bytes) return SK_OK;
  if (cfg->d_lags) HIPCHK(hipFree(cfg->d_lags));
  cfg->d_lags = nullptr;
  cfg->lags_bytes = 0;
  HIPCHK(hipMalloc((void**)&cfg->d_lags, need + need / 4));
  cfg->lags_bytes = need + need / 4;
  return SK_OK;
}
static size_t pair_count(size_t npts, const void* pairs, size_t npairs) {
  return pairs ? npairs : 1 + npts * (npts - 1) / 2;
}

int sk_kernel_values_pairs_device(sk_config* cfg, const double* d_pts, size_t npts, int pdim, const int64_t* d_pairs,
                                  size_t npairs, double k0, int param_derivative, double* d_vals, double* d_errs,
                                  void* stream) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  if (pdim < 1 || npts < 1) return fail(SK_EINVAL, "need at least one point and pdim >= 1");
  const size_t n = pair_count(npts, d_pairs, npairs);
  if (n == 0) return SK_OK;
  HIPCHK(hipSetDevice(cfg->device));
  // lags in a buffer of their own: the march's workspace may be reallocated by the call below
  size_t need = n * sizeof(double);
  if (need > cfg->pl_bytes) {
    if (cfg->d_plags) HIPCHK(hipFree(cfg->d_plags));
    cfg->d_plags = nullptr;
    cfg->pl_bytes = 0;
    HIPCHK(hipMalloc((void**)&cfg->d_plags, need + need / 4));
    cfg->pl_bytes = need + need / 4;
  }
  hipStream_t st = stream ? (hipStream_t)stream : cfg->own_stream;
  launch_pair_lags(d_pts, (long long)npts, pdim, (const long long*)d_pairs, (long long)n, cfg->d_plags, st);
  return sk_kernel_values_device(cfg, cfg->d_plags, n, k0, param_derivative, d_vals, d_errs, st);
}

int sk_kernel_values_pairs(sk_config* cfg, const double* pts, size_t npts, int pdim, const int64_t* pairs, size_t npairs,
                           double k0, int has_k0, int param_derivative, double* vals_out, double* errs_out) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  if (pdim < 1 || npts < 1) return fail(SK_EINVAL, "need at least one point and pdim >= 1");
  if (!has_k0) {
    int rc = sk_compute_k0(cfg, &k0);
    if (rc) return rc;
  }
  const size_t n = pair_count(npts, pairs, npairs);
  if (n == 0) return SK_OK;
  HIPCHK(hipSetDevice(cfg->device));
  int rc = ensure_lags(cfg, n);
  if (rc) return rc;
  hipStream_t st = cfg->own_stream;
  // points and index pairs in the config's grow-only scratch (staged uploads: no pageable asynchronous copies)
  const size_t pts_d = npts * (size_t)pdim, pts_b = ((pts_d * 8 + 255) / 256) * 256;
  rc = ensure_dev(&cfg->d_aux, &cfg->aux_bytes, pts_b + (pairs ? 2 * npairs * sizeof(int64_t) : 0));
  if (rc) return rc;
  double* d_pts = (double*)cfg->d_aux;
  int64_t* d_pairs = pairs ? (int64_t*)((char*)cfg->d_aux + pts_b) : nullptr;
  rc = h2d_doubles(cfg, d_pts, pts, pts_d, st, nullptr);
  if (!rc && pairs) {
    // the staging buffer is reused by the second upload: the first one must have left it
    if (hipStreamSynchronize(st) != hipSuccess) rc = fail(SK_EHIP, "stream synchronisation failed");
    if (!rc) rc = h2d_doubles(cfg, (double*)d_pairs, (const double*)pairs, 2 * npairs, st, nullptr);  // 8-byte words
  }
  double* d = cfg->d_lags;
  if (!rc) rc = sk_kernel_values_pairs_device(cfg, d_pts, npts, pdim, d_pairs, npairs, k0, param_derivative, d + n, d + 2 * n, st);
  if (!rc) rc = d2h_doubles2(cfg, vals_out, errs_out, d + n, n, st);
  return rc;
}

}  // extern "C"

namespace sk {
// Entries [e0, e0 + count) of a pair list (pairs != NULL) or of the dense list "0 followed by all i < j" (pairs == NULL):
// a shard's block of sk_kernel_values_pairs_sharded.  Points and the block's pairs go up, the lags are formed, sorted
// and deduplicated on this config's device, the results of the block come down in pair order.
int kernel_values_pairs_range(sk_config* cfg, const double* pts, size_t npts, int pdim, const int64_t* pairs, size_t e0,
                              size_t count, double k0, int param_derivative, double* vals_out, double* errs_out) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  if (count == 0) return SK_OK;
  HIPCHK(hipSetDevice(cfg->device));
  int rc = ensure_lags(cfg, count);
  if (rc) return rc;
  hipStream_t st = cfg->own_stream;
  const size_t pts_d = npts * (size_t)pdim, pts_b = ((pts_d * 8 + 255) / 256) * 256;
  rc = ensure_dev(&cfg->d_aux, &cfg->aux_bytes, pts_b + (pairs ? 2 * count * sizeof(int64_t) : 0));
  if (rc) return rc;
  double* d_pts = (double*)cfg->d_aux;
  int64_t* d_pairs = pairs ? (int64_t*)((char*)cfg->d_aux + pts_b) : nullptr;
  rc = h2d_doubles(cfg, d_pts, pts, pts_d, st, nullptr);
  if (!rc && pairs) {
    if (hipStreamSynchronize(st) != hipSuccess) rc = fail(SK_EHIP, "stream synchronisation failed");
    if (!rc) rc = h2d_doubles(cfg, (double*)d_pairs, (const double*)(pairs + 2 * e0), 2 * count, st, nullptr);
  }
  if (rc) return rc;
  size_t need = count * sizeof(double);
  if (need > cfg->pl_bytes) {
    if (cfg->d_plags) HIPCHK(hipFree(cfg->d_plags));
    cfg->d_plags = nullptr;
    cfg->pl_bytes = 0;
    HIPCHK(hipMalloc((void**)&cfg->d_plags, need + need / 4));
    cfg->pl_bytes = need + need / 4;
  }
  launch_pair_lags(d_pts, (long long)npts, pdim, (const long long*)d_pairs, (long long)count, cfg->d_plags, st, (long long)e0);
  double* d = cfg->d_lags;
  rc = sk_kernel_values_device(cfg, cfg->d_plags, count, k0, param_derivative, d + count, d + 2 * count, st);
  if (!rc) rc = d2h_doubles2(cfg, vals_out, errs_out, d + count, count, st);
  return rc;
}
// One march per config on lags that are already on the device (any order, e.g. pairwise lags): what sk_kernel_values_multi
// does for a host array.  *d_vals / *d_errs: ncfg x n device arrays in input order, valid until the next multi / pairs call
// on cfgs[0].  (The dim >= 2 Jacobian entry: one fast-Hankel march per integrand beats the joint march's direct Bessel
// sums at every size.)
int multi_device_core(sk_config* const* cfgs, size_t ncfg, const double* d_lags, size_t n, double k0,
                          const int* param_derivative, double** d_vals, double** d_errs) {
  sk_config* c0 = cfgs[0];
  HIPCHK(hipSetDevice(c0->device));
  hipStream_t st = c0->own_stream;
  const size_t a8 = ((n * 8 + 255) / 256) * 256;
  const size_t tb = sort_unique_temp_bytes((long long)n);
  int rc = ensure_dev(&c0->d_aux, &c0->aux_bytes, (2 + 4 * ncfg) * a8 + tb + 512);
  if (rc) return rc;
  char* buf = (char*)c0->d_aux;
  double* uxs = (double*)buf;
  long long* inv = (long long*)(buf + a8);
  double* outv = (double*)(buf + 2 * a8);
  double* oute = outv + ncfg * n;
  char* per = buf + (2 + 2 * ncfg) * a8;  // config c: its unique values and error estimates
  long long* d_nu = (long long*)(buf + (2 + 4 * ncfg) * a8);
  void* tmp = (char*)d_nu + 256;
  long long nu = 0;
  int was_unsorted = 1, bad_lags = 0;
  if (sort_unique_sync(d_lags, (long long)n, uxs, inv, d_nu, c0->h_nuniq, tmp, tb, st, &nu, &was_unsorted, &bad_lags))
    return fail(SK_EHIP, "sort failed");
  if (bad_lags) return fail(SK_EINVAL, "lags must be non-negative and finite");
  // one march per config, on the configs' own streams and side by side where they can (for_each_march)
  rc = for_each_march(cfgs, ncfg, [&](size_t c) -> int {
    hipStream_t sc = cfgs[c]->own_stream;
    double* ks = (double*)(per + (2 * c) * a8);
    double* errs = (double*)(per + (2 * c + 1) * a8);
    int r = kernel_values_device_order(cfgs[c], uxs, (size_t)nu, k0, param_derivative ? param_derivative[c] : 0, ks, errs, sc,
                                       SK_ORDER_SORTED);
    if (r) return r;
    launch_gather(ks, errs, inv, (long long)n, outv + c * n, oute + c * n, sc);
    if (hipStreamSynchronize(sc) != hipSuccess) return fail(SK_EHIP, "stream synchronisation failed");
    return SK_OK;
  });
  if (rc) return rc;
  *d_vals = outv;
  *d_errs = oute;
  return SK_OK;
}

}  // namespace sk

extern "C" {

int sk_dense_cov_matrix_device(sk_config* cfg, const double* d_pts, size_t npts, int pdim, double* d_out, void* stream) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  if (pdim < 1 || npts < 1) return fail(SK_EINVAL, "need at least one point and pdim >= 1");
  double k0;
  int rc = sk_compute_k0(cfg, &k0);  // src/utils.jl:51: kernel_values without k0
  if (rc) return rc;
  HIPCHK(hipSetDevice(cfg->device));
  const size_t n = 1 + npts * (npts - 1) / 2;
  rc = ensure_lags(cfg, n);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : cfg->own_stream;
  double* d = cfg->d_lags;
  rc = sk_kernel_values_pairs_device(cfg, d_pts, npts, pdim, nullptr, 0, k0, 0, d + n, d + 2 * n, st);
  if (rc) return rc;
  launch_fill_cov(d + n, (long long)npts, d_out, st);
  HIPCHK(hipStreamSynchronize(st));
  return SK_OK;
}

int sk_dense_cov_matrix(sk_config* cfg, const double* pts, size_t npts, int pdim, double* out) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  if (pdim < 1 || npts < 1) return fail(SK_EINVAL, "need at least one point and pdim >= 1");
  HIPCHK(hipSetDevice(cfg->device));
  hipStream_t st = cfg->own_stream;
  const size_t pts_d = npts * (size_t)pdim, pts_b = ((pts_d * 8 + 255) / 256) * 256;
  int rc = ensure_dev(&cfg->d_aux, &cfg->aux_bytes, pts_b + npts * npts * sizeof(double));
  if (rc) return rc;
  double* d_pts = (double*)cfg->d_aux;
  double* d_out = (double*)((char*)cfg->d_aux + pts_b);
  rc = h2d_doubles(cfg, d_pts, pts, pts_d, st, nullptr);
  if (!rc) rc = sk_dense_cov_matrix_device(cfg, d_pts, npts, pdim, d_out, st);
  if (!rc) rc = d2h_doubles(cfg, out, d_out, npts * npts, st);
  return rc;
}

int sk_panel_sums(sk_config* cfg, const double* nodes, const double* coef, size_t M, double lo, double hi,
                  const double* xs, size_t n, int method, double* out_re, double* out_im) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  if (n == 0) return SK_OK;
  HIPCHK(hipSetDevice(cfg->device));
  hipStream_t st = cfg->own_stream;
  double *d_nodes = nullptr, *d_coef = nullptr, *d_xs = nullptr, *d_re = nullptr, *d_im = nullptr;
  HIPCHK(hipMalloc((void**)&d_nodes, (2 * M + 3 * n + 8) * sizeof(double)));
  d_coef = d_nodes + M;
  d_xs = d_coef + M;
  d_re = d_xs + n;
  d_im = d_re + n;
  int rc = SK_OK;
  auto done = [&](int code) {
    hipFree(d_nodes);
    return code;
  };
  // test seam: plain synchronous copies (own_stream is a blocking stream, ordered behind them)
  if (hipMemcpy(d_nodes, nodes, M * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_coef, coef, M * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_xs, xs, n * 8, hipMemcpyHostToDevice) != hipSuccess)
    return done(fail(SK_EHIP, "H2D copy failed"));
  if (method == 1) {
    launch_direct(d_nodes, d_coef, (int64_t)M, d_xs, (int64_t)n, d_re, d_im, st);
  } else {
    double r_hi = 0.0;
    for (size_t i = 0; i < n; ++i) {
      if (!(xs[i] >= 0.0)) return done(fail(SK_EINVAL, "lags must be non-negative"));
      r_hi = std::fmax(r_hi, xs[i]);
    }
    if (!(hi > lo)) return done(fail(SK_EINVAL, "need hi > lo"));
    for (size_t j = 0; j < M; ++j)
      if (!(nodes[j] >= lo && nodes[j] <= hi)) return done(fail(SK_EINVAL, "nodes must lie in [lo, hi]"));
    if (r_hi == 0.0) r_hi = 1.0;
    const DevConst& dc = cfg->dc;
    const double Cx = 0.5 * (lo + hi), X = 0.5 * (hi - lo);
    const double xr = X * r_hi;
    if (xr > 0.25 * (double)cfg->M * (1.0 + 1e-9))
      return done(fail(SK_EINVAL, "space-bandwidth product (hi-lo)/2*max(xs) exceeds m*k/4 of this config"));
    const double kappa = 0.5 * xr, tau = xr * (double)dc.n2 / (double)(dc.n1 - dc.w);
    const double theta_scale = 2.0 * M_PI * xr / (double)(dc.n1 - dc.w);
    if (hipMemsetAsync(cfg->db.fft, 0, sizeof(double2) * 2 * dc.n2, st) != hipSuccess)
      return done(fail(SK_EHIP, "memset failed"));
    launch_seam_spread(dc, d_nodes, d_coef, (int64_t)M, Cx, X, kappa, cfg->db.fft, st);
    launch_seam_deconv(dc, cfg->db.fft, st);
    {
      if (rocfft_execution_info_set_stream(cfg->info_seam, st) != rocfft_status_success) return done(fail(SK_EHIP, "rocFFT stream"));
      void* in[1] = {cfg->db.fft};
      if (rocfft_execute(cfg->plan_seam, in, nullptr, cfg->info_seam) != rocfft_status_success)
        return done(fail(SK_EHIP, "rocfft_execute (seam) failed"));
    }
    launch_seam_interp(dc, cfg->d_ec, cfg->db.fft, d_xs, (int64_t)n, Cx, r_hi, tau, theta_scale, d_re, d_im, st);
  }
  if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(out_re, d_re, n * 8, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(out_im, d_im, n * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return done(fail(SK_EHIP, "D2H copy failed"));
  if (hipGetLastError() != hipSuccess) return done(fail(SK_EHIP, "kernel launch failed"));
  return done(SK_OK);
}

// What a launch costs before it does any work (bench.py "launch_floor_us"), with the same event brackets as the
// per-kernel times: [0] an empty kernel on the fused spread launch's grid, [1] that launch reduced to its
// step-state prologue (the state says "done": every workgroup derives the state and exits), [2] k_interp reduced
// to its state loads, [3] the empty kernel launched back to back (one bracket around `reps` launches, per launch).
int sk_launch_floor(sk_config* cfg, int reps, double* us4) {
  if (!cfg || !us4 || reps < 1) return fail(SK_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(cfg->device));
  hipStream_t st = cfg->own_stream;
  HIPCHK(hipStreamSynchronize(st));
  DevState fin;
  std::memset(&fin, 0, sizeof(fin));
  fin.done = 1;
  fin.rhi_for = -1;
  HIPCHK(hipMemcpy(&cfg->db.state[0], &fin, sizeof(fin), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(&cfg->db.state[1], &fin, sizeof(fin), hipMemcpyHostToDevice));
  // lags of its own: the interpolation kernels read xs[min(i, n-1)] ahead of the state (see launch_interp_b), and the
  // last call's array may be gone
  const long long nfl = 1 << 20;
  struct Scratch {  // released on every path out of this function
    double* d = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    ~Scratch() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
      if (d) (void)hipFree(d);
    }
  } sc;
  HIPCHK(hipMalloc((void**)&sc.d, nfl * sizeof(double)));
  HIPCHK(hipMemset(sc.d, 0, nfl * sizeof(double)));
  DevBuffers dbf = cfg->db;
  dbf.xs = sc.d;
  HIPCHK(hipEventCreate(&sc.a));
  HIPCHK(hipEventCreate(&sc.b));
  hipEvent_t a = sc.a, b = sc.b;
  const unsigned grid = launch_spread_b_grid(cfg->dc);
  auto bracket = [&](int which, int count) -> double {
    (void)hipEventRecord(a, st);
    for (int q = 0; q < count; ++q) {
      if (which == 0) launch_empty(grid, st);
      else if (which == 1) launch_spread_b(cfg->dc, cfg->d_ec, dbf, SK_MODE_ASIS, 0, nfl, 0.0, 0.0, 0u, st);
      else launch_interp_b(cfg->dc, cfg->d_ec, dbf, 0, nfl, st);  // (a finished state: returns at once)
    }
    (void)hipEventRecord(b, st);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return 1e3 * ms;
  };
  for (int which = 0; which < 3; ++which) {
    bracket(which, 2);  // warm
    double sum = 0.0;
    for (int r = 0; r < reps; ++r) sum += bracket(which, 1);
    us4[which] = sum / reps;
  }
  us4[3] = bracket(0, reps) / reps;
  HIPCHK(hipGetLastError());
  return SK_OK;
}

int sk_profile_enable(sk_config* cfg, int on) {
  if (!cfg) return fail(SK_EINVAL, "null config");
  cfg->profile = on ? 1 : 0;
  return SK_OK;
}

int sk_profile_read(const sk_config* cfg, sk_kernel_times* out) {
  if (!cfg || !out) return fail(SK_EINVAL, "null argument");
  *out = cfg->ktimes;
  return SK_OK;
}

int sk_last_stats(const sk_config* cfg, sk_stats* out) {
  if (!cfg || !out) return fail(SK_EINVAL, "null argument");
  *out = cfg->stats;
  return SK_OK;
}

int sk_last_trace(const sk_config* cfg_c, double* rec, size_t cap, size_t* len) {
  if (!cfg_c || !len) return fail(SK_EINVAL, "null argument");
  sk_config* cfg = const_cast<sk_config*>(cfg_c);
  if (!cfg->trace_valid) {  // fetched lazily: the records stay in device memory until the next call
    cfg->trace.resize(cfg->trace_len);
    if (cfg->trace_len) {
      HIPCHK(hipSetDevice(cfg->device));
      HIPCHK(hipMemcpy(cfg->trace.data(), cfg->db.trace, sizeof(TraceRec) * cfg->trace_len, hipMemcpyDeviceToHost));
    }
    cfg->trace_valid = true;
  }
  size_t nrec = cfg->trace.size() < cap ? cfg->trace.size() : cap;
  for (size_t i = 0; i < nrec; ++i) {
    const TraceRec& r = cfg->trace[i];
    rec[5 * i + 0] = r.a;
    rec[5 * i + 1] = r.b;
    rec[5 * i + 2] = r.max_err;
    rec[5 * i + 3] = r.n_active;
    rec[5 * i + 4] = r.accepted;
  }
  *len = nrec;
  return SK_OK;
}

int sk_abi_layout(int which, size_t* out, size_t cap, size_t* len) {
  if (!out || !len) return fail(SK_EINVAL, "null argument");
  std::vector<size_t> v;
#define F(T, f) v.push_back(offsetof(T, f))
  if (which == 0) {
    v.push_back(sizeof(sk_config_args));
    F(sk_config_args, sdf_kind); F(sk_config_args, sdf_params); F(sk_config_args, f); F(sk_config_args, f_ctx);
    F(sk_config_args, df); F(sk_config_args, df_ctx); F(sk_config_args, dim); F(sk_config_args, alpha);
    F(sk_config_args, tol); F(sk_config_args, derivative); F(sk_config_args, logw); F(sk_config_args, criteria);
    F(sk_config_args, has_tail); F(sk_config_args, tail); F(sk_config_args, m); F(sk_config_args, k);
    F(sk_config_args, device); F(sk_config_args, nufft_width); F(sk_config_args, hankel_min_lags);
    F(sk_config_args, sdf_dparam); F(sk_config_args, batch_panels);
  } else if (which == 1) {
    v.push_back(sizeof(sk_stats));
    F(sk_stats, n_unique); F(sk_stats, n_outer); F(sk_stats, n_panels); F(sk_stats, n_accepted); F(sk_stats, sum_active);
    F(sk_stats, n_steps_launched); F(sk_stats, n_syncs); F(sk_stats, last_b); F(sk_stats, n_steps); F(sk_stats, n_rerun);
    F(sk_stats, sum_active_steps); F(sk_stats, n_fix);
  } else if (which == 2) {
    v.push_back(sizeof(sk_kernel_times));
    F(sk_kernel_times, total_ms); F(sk_kernel_times, launches); F(sk_kernel_times, active_items);
  } else {
    return fail(SK_EINVAL, "sk_abi_layout: which must be 0, 1 or 2");
  }
#undef F
  *len = v.size();
  for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
  return SK_OK;
}

// host-only hooks (no device needed): used by the CPU test tier
int sk_host_gauss_rule(int64_t n, double p, double* nodes, double* weights) {
  std::vector<double> x, w;
  gauss_rule(n, p, x, w);
  if ((int64_t)x.size() != n) return fail(SK_EINVAL, "rule generation failed");
  std::memcpy(nodes, x.data(), n * 8);
  std::memcpy(weights, w.data(), n * 8);
  return SK_OK;
}

int sk_host_compute_k0(const sk_config_args* a, double* k0) {
  if (!a || !k0) return fail(SK_EINVAL, "null argument");
  int64_t m, k;
  int warned;
  int rc = validate(*a, m, k, warned);
  if (rc) return rc;
  SdfHost f;
  f.kind = a->sdf_kind;
  std::memcpy(f.prm, a->sdf_params, sizeof(f.prm));
  f.prm[7] = (double)a->sdf_dparam;
  f.f = a->f;
  f.ctx = a->f_ctx;
  double p = -a->alpha + (a->dim == 1 ? 0.0 : a->dim / 2.0) + (a->derivative ? 1.0 : 0.0);
  double c = (a->dim == 1 ? 2.0 : 2.0 * M_PI) * (a->derivative ? -2.0 * M_PI : 1.0) * (a->logw ? -1.0 : 1.0);
  *k0 = compute_k0_host(f, a->dim, p, c, a->derivative, a->logw, a->tol);
  return SK_OK;
}

int sk_host_compute_k0_part(const sk_config_args* a, double t_lo, double t_hi, double* part) {
  if (!a || !part) return fail(SK_EINVAL, "null argument");
  int64_t m, k;
  int warned;
  int rc = validate(*a, m, k, warned);
  if (rc) return rc;
  if (!(t_lo >= 0.0 && t_hi <= 1.0 && t_lo < t_hi)) return fail(SK_EINVAL, "need 0 <= t_lo < t_hi <= 1");
  SdfHost f;
  f.kind = a->sdf_kind;
  std::memcpy(f.prm, a->sdf_params, sizeof(f.prm));
  f.prm[7] = (double)a->sdf_dparam;
  f.f = a->f;
  f.ctx = a->f_ctx;
  double p = -a->alpha + (a->dim == 1 ? 0.0 : a->dim / 2.0) + (a->derivative ? 1.0 : 0.0);
  double c = (a->dim == 1 ? 2.0 : 2.0 * M_PI) * (a->derivative ? -2.0 * M_PI : 1.0) * (a->logw ? -1.0 : 1.0);
  *part = compute_k0_host(f, a->dim, p, c, a->derivative, a->logw, a->tol, t_lo, t_hi);
  return SK_OK;
}

int sk_host_validate(const sk_config_args* a, int64_t* m, int64_t* k) {
  if (!a) return fail(SK_EINVAL, "null argument");
  int64_t mm, kk;
  int warned;
  int rc = validate(*a, mm, kk, warned);
  if (rc) return rc;
  if (m) *m = mm;
  if (k) *k = kk;
  return warned ? SK_WARN_QUADSPEC : SK_OK;
}

// ES-kernel tables for host tests: phi(z), phi_hat(xi) and the piecewise polynomial value.
int sk_host_es_eval(int w, double z, double* phi, double* phi_poly) {
  static thread_local EsTables t;
  static thread_local int tw = 0;
  if (tw != w) {
    build_es_tables(w, 64, 128, 1.6, t);
    tw = w;
  }
  if (phi) *phi = es_phi(z, w, t.beta);
  if (phi_poly) {
    double zq = z + 0.5 * w;
    int q = (int)std::floor(zq);
    q = q < 0 ? 0 : (q > w - 1 ? w - 1 : q);
    double y = zq - q - 0.5;
    int qq = q < w / 2 ? q : w - 1 - q;
    const double* pe = &t.poly[(size_t)qq * 2 * t.nco];
    const double* po = pe + t.nco;
    double E = pe[0], O = po[0];
    for (int i = 1; i < t.nco; ++i) {
      E = E * y * y + pe[i];
      O = O * y * y + po[i];
    }
    *phi_poly = (q < w / 2) ? E + y * O : E - y * O;
  }
  return SK_OK;
}

}  // extern "C"
