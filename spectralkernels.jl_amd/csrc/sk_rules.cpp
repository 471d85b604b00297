// sk_rules.cpp -- Gauss-Legendre / Gauss-Jacobi(0,p) rules on the host.
//
// Counterpart of the FastGaussQuadrature.jl calls at src/quadrature.jl:37-42 (third-party,
// not in the reference tree).  Own algorithm: Golub-Welsch eigenvalues of the Jacobi matrix
// as starting values (Legendre: the classical cosine guess), then Newton on the three-term
// recurrence in long double, weights from the derivative formula
//   w_j = 2^(p+1) / ((1 - x_j^2) P_n'(x_j)^2)            (alpha = 0, beta = p).
#include <sched.h>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <map>
#include <mutex>
#include <thread>
#include "sk_internal.h"

namespace sk {

int host_threads(int cap) {
  static const int avail = [] {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
    // cgroup v2: "<quota> <period>" or "max <period>"; v1: two files, quota -1 = unlimited
    long long quota = -1, period = 0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = std::atoll(q);
      std::fclose(f);
    } else {
      if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
        std::fclose(g);
      }
      if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(g, "%lld", &period) != 1) period = 0;
        std::fclose(g);
      }
    }
    if (quota > 0 && period > 0) {
      const int q = (int)((quota + period - 1) / period);
      if (q < n) n = q;
    }
    return n < 1 ? 1 : n;
  }();
  return cap > 0 && cap < avail ? cap : avail;
}

namespace {

typedef long double ld;

// P_n^{(0,b)}(x) and its derivative by the three-term recurrence.
inline void jacobi_eval(int64_t n, ld b, ld x, ld& pn, ld& dpn) {
  ld p0 = 1.0L, p1 = ((b + 2.0L) * x - b) / 2.0L;  // P_1^{(0,b)} = ((b+2)x - b)/2
  if (n == 0) { pn = 1; dpn = 0; return; }
  for (int64_t k = 2; k <= n; ++k) {
    ld kk = (ld)k, s = 2 * kk + b;  // 2k + a + b, a = 0
    ld c1 = 2 * kk * (kk + b) * (s - 2);
    ld c2 = (s - 1) * (s * (s - 2) * x - b * b);
    ld c3 = 2 * (kk - 1) * (kk + b - 1) * s;
    ld p2 = (c2 * p1 - c3 * p0) / c1;
    p0 = p1; p1 = p2;
  }
  ld nn = (ld)n, s = 2 * nn + b;
  // (2n+b)(1-x^2) P_n' = n[-b - (2n+b)x] P_n + 2 n (n+b) P_{n-1}
  dpn = (nn * (-b - s * x) * p1 + 2 * nn * (nn + b) * p0) / (s * (1 - x * x));
  pn = p1;
}

inline void legendre_eval(int64_t n, ld x, ld& pn, ld& dpn) {
  ld p0 = 1.0L, p1 = x;
  for (int64_t k = 2; k <= n; ++k) {
    ld p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    p0 = p1; p1 = p2;
  }
  pn = p1;
  dpn = n * (x * p1 - p0) / (x * x - 1);
}

// Eigenvalues of a symmetric tridiagonal matrix (implicit QL with Wilkinson shifts).
// d: diagonal (n), e: sub-diagonal in e[0..n-2].  Overwrites d with eigenvalues.
bool tridiag_eigenvalues(std::vector<double>& d, std::vector<double>& e) {
  const int64_t n = (int64_t)d.size();
  e.push_back(0.0);
  for (int64_t l = 0; l < n; ++l) {
    int iter = 0;
    int64_t mm;
    do {
      for (mm = l; mm < n - 1; ++mm) {
        double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
        if (std::fabs(e[mm]) <= 2.3e-16 * dd) break;
      }
      if (mm != l) {
        if (++iter > 200) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[mm] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int64_t i;
        for (i = mm - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) { d[i + 1] -= p; e[mm] = 0.0; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[mm] = 0.0;
      }
    } while (mm != l);
  }
  return true;
}

std::mutex g_cache_mu;
std::map<std::pair<int64_t, double>, std::pair<std::vector<double>, std::vector<double>>> g_cache;

}  // namespace

void gauss_rule(int64_t n, double p, std::vector<double>& nodes, std::vector<double>& weights) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_cache.find({n, p});
    if (it != g_cache.end()) { nodes = it->second.first; weights = it->second.second; return; }
  }
  nodes.assign(n, 0.0);
  weights.assign(n, 0.0);
  if (p == 0.0) {
    const int64_t half = (n + 1) / 2;
#pragma omp parallel for schedule(dynamic, 64) num_threads(host_threads(8))
    for (int64_t j = 0; j < half; ++j) {
      // j-th largest root
      ld x = cosl(3.14159265358979323846264338327950288L * (j + 0.75L) / (n + 0.5L));
      ld pn, dpn;
      for (int it = 0; it < 100; ++it) {
        legendre_eval(n, x, pn, dpn);
        ld dx = pn / dpn;
        x -= dx;
        if (fabsl(dx) <= 2e-19L * fmaxl(1.0L, fabsl(x))) break;
      }
      legendre_eval(n, x, pn, dpn);
      ld wgt = 2.0L / ((1 - x * x) * dpn * dpn);
      nodes[n - 1 - j] = (double)x;
      weights[n - 1 - j] = (double)wgt;
      nodes[j] = (double)(-x);
      weights[j] = (double)wgt;
    }
    if (n % 2) nodes[n / 2] = 0.0;
  } else {
    // Golub-Welsch starting values for weight (1+x)^p
    std::vector<double> d(n), e(n > 1 ? n - 1 : 0);
    const double b = p;
    for (int64_t i = 0; i < n; ++i) {
      double s = 2.0 * i + b;
      d[i] = (i == 0) ? b / (b + 2.0) : (b * b) / (s * (s + 2.0));
    }
    for (int64_t i = 1; i < n; ++i) {
      double s = 2.0 * i + b;
      double num = 4.0 * i * i * (i + b) * (i + b);
      double den = s * s * (s + 1.0) * (s - 1.0);
      e[i - 1] = std::sqrt(num / den);
    }
    if (!tridiag_eigenvalues(d, e)) { nodes.clear(); weights.clear(); return; }
    std::sort(d.begin(), d.end());
    const ld bl = (ld)p;
    const ld c0 = powl(2.0L, bl + 1.0L);
#pragma omp parallel for schedule(dynamic, 64) num_threads(host_threads(8))
    for (int64_t j = 0; j < n; ++j) {
      ld x = d[j], pn, dpn;
      for (int it = 0; it < 50; ++it) {
        jacobi_eval(n, bl, x, pn, dpn);
        ld dx = pn / dpn;
        x -= dx;
        if (fabsl(dx) <= 2e-19L * fmaxl(1e-3L, 1.0L - fabsl(x))) break;
      }
      jacobi_eval(n, bl, x, pn, dpn);
      nodes[j] = (double)x;
      weights[j] = (double)(c0 / ((1 - x * x) * dpn * dpn));
    }
  }
  std::lock_guard<std::mutex> lk(g_cache_mu);
  g_cache[{n, p}] = {nodes, weights};
}

}  // namespace sk
