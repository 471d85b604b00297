// sk_tables.cpp -- "exponential of semicircle" spreading-kernel tables for the type-3 NUFFT.
//
// The reference reaches FINUFFT (third-party, unpinned) through finufft1d3 (src/utils.jl:10)
// with eps = 1e-15.  This is an independent construction following the published ES-kernel
// algorithm (Barnett, Magland, af Klinteberg 2019): phi(z) = exp(beta (sqrt(1-(2z/w)^2) - 1)),
// beta = 2.30 w for upsampling 2; piecewise-polynomial evaluation; Fourier transform by
// Gauss-Legendre quadrature.
#include <cmath>
#include <vector>
#include "sk_internal.h"
#include "sk_device.h"

namespace sk {
typedef long double ld;
static const ld PI_L = 3.14159265358979323846264338327950288L;

static ld es_phi_l(ld z, int w, ld beta) {
  ld t = 1.0L - (2.0L * z / w) * (2.0L * z / w);
  if (t <= 0) return (t == 0) ? expl(-beta) : 0.0L;
  return expl(beta * (sqrtl(t) - 1.0L));
}
double es_phi(double z, int w, double beta) { return (double)es_phi_l(z, w, beta); }

namespace {
struct HatQuad {
  std::vector<ld> z, f;  // nodes on [0, w/2] and phi(z) * weight
};
HatQuad make_hat_quad(int w, ld beta) {
  int nq = 6 * w + 32;
  std::vector<double> x, wt;
  gauss_rule(nq, 0.0, x, wt);
  HatQuad h;
  h.z.resize(nq);
  h.f.resize(nq);
  for (int i = 0; i < nq; ++i) {
    ld z = ((ld)x[i] + 1.0L) * w / 4.0L;
    h.z[i] = z;
    h.f[i] = es_phi_l(z, w, beta) * (ld)wt[i] * w / 4.0L;
  }
  return h;
}
ld hat_eval(const HatQuad& h, ld xi) {
  ld s = 0;
  for (size_t i = 0; i < h.z.size(); ++i) s += h.f[i] * cosl(xi * h.z[i]);
  return 2.0L * s;
}
}  // namespace

double es_phi_hat(double xi, int w, double beta) {
  HatQuad h = make_hat_quad(w, beta);
  return (double)hat_eval(h, xi);
}

void build_es_tables(int w, int64_t n1, int64_t n2, double theta_max, EsTables& t) {
  t.w = w;
  t.beta = 2.30 * w;
  if (w == 2) t.beta = 2.20 * w;
  else if (w == 3) t.beta = 2.26 * w;
  else if (w == 4) t.beta = 2.38 * w;
  const ld beta = t.beta;
  // ---- piecewise polynomials, degree 2*nco-1 (es_nco, sk_device.h: at the kernel's own cutoff accuracy exp(-beta))
  const int nco = es_nco(w);
  const int deg = 2 * nco - 1;
  t.nco = nco;
  t.poly.assign((size_t)(w / 2) * 2 * nco, 0.0);
  const int np = deg + 1;
  for (int q = 0; q < w / 2; ++q) {
    // Chebyshev interpolation in s = 2y on [-1,1]
    std::vector<ld> fv(np), cc(np);
    for (int i = 0; i < np; ++i) {
      ld s = cosl(PI_L * (i + 0.5L) / np);
      fv[i] = es_phi_l((ld)q - (w - 1) / 2.0L + s / 2.0L, w, beta);
    }
    for (int k = 0; k < np; ++k) {
      ld acc = 0;
      for (int i = 0; i < np; ++i) acc += fv[i] * cosl(PI_L * k * (i + 0.5L) / np);
      cc[k] = acc * 2.0L / np;
    }
    cc[0] /= 2;
    // Chebyshev series -> monomials in s (T_{k+1} = 2 s T_k - T_{k-1})
    std::vector<ld> mono(np, 0.0L), Tkm1(np, 0.0L), Tk(np, 0.0L), Tn(np);
    Tkm1[0] = 1;  // T0
    for (int i = 0; i < np; ++i) mono[i] += cc[0] * Tkm1[i];
    if (np > 1) {
      Tk[1] = 1;  // T1
      for (int i = 0; i < np; ++i) mono[i] += cc[1] * Tk[i];
    }
    for (int k = 2; k < np; ++k) {
      for (int i = 0; i < np; ++i) Tn[i] = (i > 0 ? 2 * Tk[i - 1] : 0.0L) - Tkm1[i];
      for (int i = 0; i < np; ++i) mono[i] += cc[k] * Tn[i];
      Tkm1 = Tk;
      Tk = Tn;
    }
    // s = 2y
    ld sc = 1;
    for (int i = 0; i < np; ++i) { mono[i] *= sc; sc *= 2; }
    double* E = &t.poly[(size_t)q * 2 * nco];
    double* O = E + nco;
    // Horner order: highest power first.  E(u)=sum e_i u^i with u=y^2 (even powers), O likewise.
    for (int i = 0; i < nco; ++i) {
      E[nco - 1 - i] = (double)mono[2 * i];
      O[nco - 1 - i] = (double)mono[2 * i + 1];
    }
  }
  // ---- Fourier transform tables
  HatQuad h = make_hat_quad(w, beta);
  t.deconv.resize((size_t)(n1 / 2 + 1));
#pragma omp parallel for schedule(static) num_threads(host_threads(8))
  for (int64_t l = 0; l <= n1 / 2; ++l)
    t.deconv[l] = (double)(1.0L / hat_eval(h, 2.0L * PI_L * (ld)l / (ld)n2));
  // 1/phi_hat(theta) ~ sum_k cheb[k] T_k(2 (theta/theta_max)^2 - 1)
  t.theta_max = theta_max;
  const int nc = 16;
  t.ncheb = nc;
  t.cheb.assign(nc, 0.0);
  std::vector<ld> fv(nc);
  for (int i = 0; i < nc; ++i) {
    ld s = cosl(PI_L * (i + 0.5L) / nc);
    ld th = sqrtl((s + 1) / 2) * theta_max;
    fv[i] = 1.0L / hat_eval(h, th);
  }
  for (int k = 0; k < nc; ++k) {
    ld acc = 0;
    for (int i = 0; i < nc; ++i) acc += fv[i] * cosl(PI_L * k * (i + 0.5L) / nc);
    t.cheb[k] = (double)(acc * 2.0L / nc);
  }
  t.cheb[0] /= 2;
}

}  // namespace sk
