// sk_k0.cpp -- K(0) by adaptive Gauss-Kronrod quadrature on [0, inf)  (src/adaptive.jl:74-91).
//
// The reference calls QuadGK.quadgk(integrand, 0, Inf; atol=0, rtol=min(1e-8, 1e-2 tol))
// (third-party, unpinned).  Restated from QuadGK's published algorithm: the half-line is
// mapped by w = t/(1-t), t in [0,1); a 15-point Kronrod rule with embedded 7-point Gauss rule
// on each segment; the segment with the largest error estimate is bisected until the summed
// estimate is below rtol*|I| (or the evaluation budget is spent).
#include <cmath>
#include <queue>
#include <vector>
#include "sk_internal.h"

namespace sk {

double sdf_eval_host(int kind, const double* p, double w) {
  const int dparam = (int)p[7];  // 0: the SDF; j >= 1: d/d(parameter j - 1), as sdf_eval_dev
  switch (kind) {
    case SK_SDF_MATERN: {
      const double x = p[1] * p[1] + w * w, e = -p[2] - p[3] / 2.0, S = p[0] * std::pow(x, e);
      if (dparam == 0) return S;
      if (dparam == 1) return S / p[0];
      if (dparam == 2) return S * e * 2.0 * p[1] / x;
      if (dparam == 3) return -S * std::log(x);
      return NAN;
    }
    case SK_SDF_EXPONENTIAL: {
      const double S = std::exp(-p[0] * std::fabs(w));
      return dparam == 0 ? S : (dparam == 1 ? -std::fabs(w) * S : NAN);
    }
    case SK_SDF_GAUSSIAN: {
      const double S = std::exp(-(w * w) / (p[0] * p[0]));
      return dparam == 0 ? S : (dparam == 1 ? S * 2.0 * w * w / (p[0] * p[0] * p[0]) : NAN);
    }
    default: return NAN;
  }
}

double SdfHost::operator()(double w) const {
  if (kind == SK_SDF_CALLBACK) {
    double out;
    f(&w, &out, 1, ctx);
    return out;
  }
  return sdf_eval_host(kind, prm, w);
}

void SdfHost::batch(const double* w, double* out, size_t n) const {
  if (kind == SK_SDF_CALLBACK) {
    f(w, out, n, ctx);
    return;
  }
  for (size_t i = 0; i < n; ++i) out[i] = sdf_eval_host(kind, prm, w[i]);
}

namespace {
// 15-point Kronrod abscissae (positive half, descending) and weights; 7-point Gauss weights.
const double XGK[8] = {0.991455371120812639206854697526329, 0.949107912342758524526189684047851,
                       0.864864423359769072789712788640926, 0.741531185599394439863864773280788,
                       0.586087235467691130294144838258730, 0.405845151377397166906606412076961,
                       0.207784955007898467600689403773245, 0.000000000000000000000000000000000};
const double WGK[8] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                       0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                       0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                       0.204432940075298892414161999234649, 0.209482141084727828012999174891714};
const double WG[4] = {0.129484966168869693270611432679082, 0.279705391489276667901467771423780,
                      0.381830050505118944950369775488975, 0.417959183673469387755102040816327};

struct Seg {
  double a, b, I, E;
  bool operator<(const Seg& o) const { return E < o.E; }
};

template <class F>
Seg eval_seg(const F& g, double a, double b, long& nev) {
  double c = 0.5 * (a + b), h = 0.5 * (b - a);
  double fc = g(c);
  double rk = WGK[7] * fc, rg = WG[3] * fc;
  for (int j = 0; j < 7; ++j) {
    double dx = h * XGK[j];
    double f1 = g(c - dx), f2 = g(c + dx);
    rk += WGK[j] * (f1 + f2);
    if (j % 2 == 1) rg += WG[j / 2] * (f1 + f2);
  }
  nev += 15;
  Seg s{a, b, rk * h, std::fabs((rk - rg) * h)};
  return s;
}
}  // namespace

double compute_k0_host(const SdfHost& f, int dim, double p, double c, int derivative, int logw, double tol, double t_lo,
                       double t_hi) {
  // src/adaptive.jl:77-80
  double L = 1.0;
  const double f0 = std::fabs(f(0.0));
  for (int it = 0; it < 2000 && std::pow(L, p) * std::fabs(f(L)) > f0 / 2; ++it) L *= 2;
  const double nu = dim / 2.0 - 1 + (derivative ? 1 : 0);
  const double gnu = std::tgamma(nu + 1);
  auto integrand = [&](double w) -> double {
    double wl = w * L;
    double v = std::pow(wl, p) * (logw ? std::log(wl) : 1.0) * f(wl) * L;
    if (dim != 1) v *= std::pow(M_PI * w, nu) / gnu;  // src/adaptive.jl:84-86
    return v;
  };
  // t in [0,1) -> w = t/(1-t)
  auto g = [&](double t) -> double {
    double om = 1.0 - t;
    if (om <= 0) return 0.0;
    double v = integrand(t / om) / (om * om);
    return std::isfinite(v) ? v : 0.0;
  };
  const double rtol = std::fmin(1e-8, 1e-2 * tol);
  long nev = 0;
  std::priority_queue<Seg> heap;
  Seg s0 = eval_seg(g, t_lo, t_hi, nev);
  double I = s0.I, E = s0.E;
  heap.push(s0);
  const long maxev = 10000000;  // QuadGK default maxevals = 10^7
  while (E > rtol * std::fabs(I) && nev < maxev) {
    Seg s = heap.top();
    heap.pop();
    double mid = 0.5 * (s.a + s.b);
    if (!(mid > s.a && mid < s.b)) { heap.push(Seg{s.a, s.b, s.I, 0.0}); E -= s.E; continue; }
    Seg l = eval_seg(g, s.a, mid, nev), r = eval_seg(g, mid, s.b, nev);
    I += (l.I + r.I) - s.I;
    E += (l.E + r.E) - s.E;
    heap.push(l);
    heap.push(r);
    if ((heap.size() & 1023) == 0) {  // re-sum against drift
      std::priority_queue<Seg> tmp = heap;
      I = 0; E = 0;
      while (!tmp.empty()) { I += tmp.top().I; E += tmp.top().E; tmp.pop(); }
    }
  }
  return c * I;
}

}  // namespace sk
