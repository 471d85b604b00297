// sk_kernels.hip -- gfx950 kernels of the kernel_values hot path.
//
// One (sub)panel evaluation ("step") is
//   k_gen     nodes, SDF, Gauss weights, pre-phase      (src/quadrature.jl:49-95 updatequadbufs!)
//   k_spread  output-driven ES spreading + deconvolution (type-3 NUFFT, replaces finufft1d3,
//   rocFFT    length n2, batch 2 (m- and 2m-rule)         src/utils.jl:10 / src/quadrature.jl:109-110)
//   k_interp  interpolation at the active lags, post-phase, scaling, |I2-I1|, max reduction
//                                                        (src/quadrature.jl:130-136, :250-258)
//   k_accum   accept/reject (src/quadrature.jl:260-263), outer accumulation (src/adaptive.jl:163-164),
//             convergence predicate (src/adaptive.jl:183-198, :222-233)
//   k_advance scalar transition: bisection stack (src/quadrature.jl:268-270), next outer panel
//             (src/adaptive.jl:149-152), tail fit (src/adaptive.jl:204-220)
// All state lives in HBM (DevState ping-pong), so steps run back to back with no host hop.
#include <cstring>
#include <cmath>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "sk_device.h"
#include "../../include/spectralkernels_hip.h"

namespace sk {

#define SK_PI 3.14159265358979323846

// exp(2 pi i a b) with the product reduced modulo 1 exactly (FMA residual kept).
__device__ __forceinline__ double2 cis_turns_prod(double a, double b) {
  double p = a * b;
  double e = fma(a, b, -p);
  double fr = p - rint(p);
  double s, c;
  sincospi(2.0 * (fr + e), &s, &c);
  return make_double2(c, s);
}

__device__ __forceinline__ double sdf_eval_dev(int kind, const double* prm, double w) {
  switch (kind) {
    case SK_SDF_MATERN: return prm[0] * pow(prm[1] * prm[1] + w * w, -prm[2] - prm[3] * 0.5);
    case SK_SDF_EXPONENTIAL: return exp(-prm[0] * fabs(w));
    case SK_SDF_GAUSSIAN: return exp(-(w * w) / (prm[0] * prm[0]));
    default: return __builtin_nan("");
  }
}

// ------------------------------------------------------------------ scalar state logic

__device__ void set_panel(DevState& s, const DevConst& dc, const double* xs, double pa, double pb) {
  PanelParams& pp = s.pp;
  pp.pa = pa;
  pp.pb = pb;
  if (!(fabs(pb - pa) > 1e-16)) {  // src/utils.jl:28-36
    s.done = -SK_ESUBDIVIDE;
    return;
  }
  pp.Cx = 0.5 * (pa + pb);
  pp.X = 0.5 * (pb - pa);
  double r_hi = xs[s.hi - 1];
  pp.r_hi = r_hi;
  double xr = pp.X * r_hi;
  pp.kappa = 0.5 * xr;
  pp.tau = xr * (double)dc.n2 / (double)(dc.n1 - dc.w);
  pp.theta_scale = 2.0 * SK_PI * xr / (double)(dc.n1 - dc.w);
  pp.jacobi = (dc.p != 0.0 && pa == 0.0) ? 1 : 0;
}

// src/adaptive.jl:149-152: next outer panel or finish.
__device__ void setup_outer(DevState& s, const DevConst& dc, const double* xs) {
  if (s.hi > s.ix1) {
    double a = s.b;
    double b = s.b + (double)dc.M / (2.0 * xs[s.hi - 1]);
    s.a = a;
    s.b = b;
    s.n_outer += 1;
    if (dc.sdf_kind != SK_SDF_CALLBACK) s.fb = sdf_eval_dev(dc.sdf_kind, dc.sdf_prm, b);
    set_panel(s, dc, xs, a, b);
  } else {
    s.done = 1;
  }
}

// src/adaptive.jl:204-220 with all 1000 sample points equal to b: minimum-norm least squares.
__device__ __forceinline__ void tail_fit(const DevConst& dc, double b, double fb_in, double& tc, double& td) {
  double fb = fabs(fb_in);
  double d;
  if (!dc.has_tail) {
    double L = log(b), y = log(fb);
    d = isfinite(y) ? L * y / (1.0 + L * L) : __builtin_nan("");
  } else {
    d = dc.tail;
  }
  d -= dc.alpha;
  tc = (pow(b, d) * fb) / pow(b, 2.0 * d);
  td = d;
}

// src/adaptive.jl:222-229
__device__ __forceinline__ double trunc_est(double b, double x, double c, double d, int dim) {
  double t1 = -c / (d + dim) * pow(b, d + dim);
  double t2 = c * pow(b, d + (dim - 1) * 0.5) / (2.0 * SK_PI * pow(x, (dim + 1) * 0.5));
  return fmin(t1, t2);
}

__device__ __forceinline__ bool step_accept(const DevState& s, double tol) {
  double maxe = __longlong_as_double((long long)s.max_err_bits);
  return !s.nan1 && (maxe < tol * s.k0abs);  // src/quadrature.jl:260
}

__global__ void k_init(DevConst dc, DevBuffers db, long long n, double k0, double ks0) {
  DevState s;
  memset(&s, 0, sizeof(s));
  s.n = n;
  s.k0abs = fabs(k0);
  s.ks0 = ks0;
  s.hi = n;
  s.ix1 = (n > 0 && db.xs[0] == 0.0) ? 1 : 0;  // src/adaptive.jl:132-133
  s.crit = dc.crit0;
  s.a = 0.0;
  s.b = 0.0;
  s.hi_new = (unsigned long long)s.ix1;
  setup_outer(s, dc, db.xs);
  db.state[0] = s;
  db.state[1] = s;
}

__global__ void k_advance(DevConst dc, DevBuffers db, int slot) {
  DevState s = db.state[slot];
  if (!s.done) {
    const bool accept = step_accept(s, dc.tol);
    const long long nact = s.hi - s.ix1;
    if (s.trace_len < kTraceCap) {
      TraceRec r;
      r.a = s.pp.pa;
      r.b = s.pp.pb;
      r.max_err = __longlong_as_double((long long)s.max_err_bits);
      r.n_active = (double)nact;
      r.accepted = accept ? 1.0 : 0.0;
      db.trace[s.trace_len++] = r;
    }
    s.n_panels += 1;
    s.sum_active += nact;
    s.retire_pending = 0;
    if (!s.nan1 && s.nan2) {  // src/quadrature.jl:165 (throws only when int1 is NaN-free)
      s.done = -SK_ENAN;
    } else if (accept) {
      s.n_accepted += 1;
      if (s.stack_len > 0) {
        double2 ab = db.stack[--s.stack_len];
        set_panel(s, dc, db.xs, ab.x, ab.y);
      } else {
        // outer panel complete: src/adaptive.jl:168-199
        if (s.crit != SK_CRIT_PANEL) {
          double tc, td;
          tail_fit(dc, s.b, s.fb, tc, td);
          if (isnan(tc) || isnan(td)) s.crit = SK_CRIT_PANEL;
        }
        s.hi_prev = s.hi;
        s.hi = (long long)s.hi_new;
        s.retire_pending = 1;
        setup_outer(s, dc, db.xs);
      }
    } else {
      // src/quadrature.jl:268-270: push left, then right; right is popped next.
      if (s.stack_len + 2 > kStackCap) {
        s.done = -SK_ESTACK;
      } else {
        double mid = 0.5 * (s.pp.pa + s.pp.pb);
        db.stack[s.stack_len++] = make_double2(s.pp.pa, mid);
        set_panel(s, dc, db.xs, mid, s.pp.pb);
      }
    }
    s.max_err_bits = 0ull;
    s.hi_new = (unsigned long long)s.ix1;
    s.nan1 = 0;
    s.nan2 = 0;
  } else {
    s.retire_pending = 0;
  }
  db.state[slot ^ 1] = s;
}

// ------------------------------------------------------------------ k_gen

__global__ void __launch_bounds__(256) k_gen(DevConst dc, DevBuffers db, int slot) {
  const DevState& s = db.state[slot];
  if (s.done) return;
  const PanelParams pp = s.pp;
  const int64_t M = dc.M;
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 3 * M) return;
  const int rule = j < M ? 0 : 1;
  const int64_t jj = rule ? j - M : j;
  const int64_t mr = rule ? 2 * dc.m : dc.m;
  const int64_t sp = jj / mr, q = jj - sp * mr;
  const double pa = pp.pa, pb = pp.pb;
  double sa = (sp == 0) ? pa : pa + (pb - pa) * ((double)sp / (double)dc.k);
  double sb = (sp + 1 == dc.k) ? pb : pa + (pb - pa) * ((double)(sp + 1) / (double)dc.k);
  const double bmad2 = 0.5 * (sb - sa), bpad2 = 0.5 * (sb + sa);
  double w, wq;
  if (pp.jacobi && sp == 0) {  // src/quadrature.jl:61-78
    const double* no = rule ? dc.jac_no2 : dc.jac_no1;
    const double* wt = rule ? dc.jac_wt2 : dc.jac_wt1;
    w = bmad2 * no[q] + bpad2;
    wq = wt[q] * pow(bmad2, dc.p + 1.0);
  } else {  // src/quadrature.jl:82-92
    const double* no = rule ? dc.leg_no2 : dc.leg_no1;
    const double* wt = rule ? dc.leg_wt2 : dc.leg_wt1;
    w = bmad2 * no[q] + bpad2;
    wq = wt[q] * bmad2 * (dc.p == 0.0 ? 1.0 : pow(w, dc.p));
  }
  const double f = (dc.sdf_kind == SK_SDF_CALLBACK) ? db.fvals[j] : sdf_eval_dev(dc.sdf_kind, dc.sdf_prm, w);
  const double cj = wq * f;
  const double P = dc.pos[pp.jacobi][rule][jj];
  const double u = (P - 0.5 * (double)dc.n1) * dc.delta;
  const double2 ph = cis_turns_prod(pp.kappa, u);
  double2* dst = rule ? db.src2 : db.src1;
  dst[jj] = make_double2(cj * ph.x, cj * ph.y);
}

// ------------------------------------------------------------------ ES kernel evaluation

// Even/odd Horner evaluation of pair q at offset y: E(y^2), O(y^2).
template <int W>
__device__ __forceinline__ void es_pair(const double* __restrict__ poly, int q, double y2, double& E, double& O) {
  constexpr int NCO = W / 2 + 2;
  const double* pe = poly + q * 2 * NCO;
  const double* po = pe + NCO;
  E = pe[0];
  O = po[0];
#pragma unroll
  for (int i = 1; i < NCO; ++i) {
    E = fma(E, y2, pe[i]);
    O = fma(O, y2, po[i]);
  }
}

// ------------------------------------------------------------------ k_spread

template <int W>
__global__ void __launch_bounds__(256) k_spread(DevConst dc, DevBuffers db, int slot) {
  constexpr int NCO = W / 2 + 2;
  __shared__ double spoly[(W / 2) * 2 * NCO];
  const DevState& s = db.state[slot];
  if (s.done) return;
  for (int i = threadIdx.x; i < (W / 2) * 2 * NCO; i += blockDim.x) spoly[i] = dc.poly[i];
  __syncthreads();
  const int variant = s.pp.jacobi;
  const int64_t n1 = dc.n1, n2 = dc.n2;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * n2) return;
  const int rule = idx >= n2 ? 1 : 0;
  const int64_t g = rule ? idx - n2 : idx;
  const int64_t l = (g < n2 / 2) ? g : g - n2;
  double2 out = make_double2(0.0, 0.0);
  if (l >= -(n1 / 2) && l < n1 - n1 / 2) {
    const int64_t L = l + n1 / 2;
    const int* cs = dc.cellstart[variant][rule];
    const double* pos = dc.pos[variant][rule];
    const double2* src = rule ? db.src2 : db.src1;
    const int64_t c0 = L - W / 2 < 0 ? 0 : L - W / 2;
    const int64_t c1 = L + W / 2 > n1 ? n1 : L + W / 2;
    const int jlo = cs[c0], jhi = cs[c1];
    double ar = 0.0, ai = 0.0;
    for (int j = jlo; j < jhi; ++j) {
      const double zq = (double)L - pos[j] + 0.5 * W;  // in [0, W]
      int q = (int)floor(zq);
      q = q < 0 ? 0 : (q > W - 1 ? W - 1 : q);
      const double y = zq - (double)q - 0.5;
      const int qq = q < W / 2 ? q : W - 1 - q;
      double E, O;
      es_pair<W>(spoly, qq, y * y, E, O);
      const double phi = (q < W / 2) ? fma(y, O, E) : fma(-y, O, E);
      const double2 cj = src[j];
      ar = fma(cj.x, phi, ar);
      ai = fma(cj.y, phi, ai);
    }
    const double dcv = dc.deconv[l < 0 ? -l : l];
    out = make_double2(ar * dcv, ai * dcv);
  }
  db.fft[idx] = out;
}

// ------------------------------------------------------------------ interpolation core

// Returns F = exp(2 pi i Cx r) * G(v) for both rules: the panel sum sum_j c_j exp(2 pi i w_j r).
template <int W>
__device__ __forceinline__ void interp_core(const DevConst& dc, const double* __restrict__ spoly,
                                            const double* __restrict__ scheb, const double2* __restrict__ fft, int nrule,
                                            double r, double Cx, double r_hi, double tau, double theta_scale,
                                            double2& F1, double2& F2) {
  const int64_t n2 = dc.n2;
  const double Rc = 0.5 * r_hi;
  const double v = (r - Rc) / Rc;
  const double t = tau * v;
  const double tcn = t - 0.5 * W;
  const double k0d = ceil(tcn);
  const double y = (k0d - tcn) - 0.5;
  const double y2 = y * y;
  long long k0 = (long long)k0d;
  double wt[W];
#pragma unroll
  for (int q = 0; q < W / 2; ++q) {
    double E, O;
    es_pair<W>(spoly, q, y2, E, O);
    wt[q] = fma(y, O, E);
    wt[W - 1 - q] = fma(-y, O, E);
  }
  double h1r = 0, h1i = 0, h2r = 0, h2i = 0;
  long long kk = k0 < 0 ? k0 + n2 : k0;
#pragma unroll
  for (int q = 0; q < W; ++q) {
    const double2 u1 = fft[kk];
    h1r = fma(u1.x, wt[q], h1r);
    h1i = fma(u1.y, wt[q], h1i);
    if (nrule > 1) {
      const double2 u2 = fft[n2 + kk];
      h2r = fma(u2.x, wt[q], h2r);
      h2i = fma(u2.y, wt[q], h2i);
    }
    kk = (kk + 1 == n2) ? 0 : kk + 1;
  }
  // 1/phi_hat(theta) by Clenshaw in s = 2 (theta/theta_max)^2 - 1
  const double th = theta_scale * v / dc.theta_max;
  double sarg = 2.0 * th * th - 1.0;
  sarg = sarg > 1.0 ? 1.0 : sarg;
  double b1 = 0.0, b2 = 0.0;
  for (int kq = dc.ncheb - 1; kq >= 1; --kq) {
    const double b0 = fma(2.0 * sarg, b1, scheb[kq] - b2);
    b2 = b1;
    b1 = b0;
  }
  const double invhat = fma(sarg, b1, scheb[0] - b2);
  const double2 ph = cis_turns_prod(Cx, r);
  h1r *= invhat; h1i *= invhat; h2r *= invhat; h2i *= invhat;
  F1 = make_double2(h1r * ph.x - h1i * ph.y, h1r * ph.y + h1i * ph.x);
  F2 = make_double2(h2r * ph.x - h2i * ph.y, h2r * ph.y + h2i * ph.x);
}

template <int W>
__global__ void __launch_bounds__(256) k_interp(DevConst dc, DevBuffers db, int slot, long long n) {
  constexpr int NCO = W / 2 + 2;
  __shared__ double spoly[(W / 2) * 2 * NCO];
  __shared__ double scheb[kMaxCheb];
  __shared__ unsigned long long smax[4];
  __shared__ int sflag[2];
  DevState& s = db.state[slot];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // deferred 2*trunc_err for the lags retired by the previous outer panel (src/adaptive.jl:194)
  if (s.retire_pending && i >= s.hi && i < s.hi_prev) db.errs[i] += db.tmpE[i];
  if (s.done) return;
  const long long ix1 = s.ix1, hi = s.hi;
  if ((long long)blockIdx.x * blockDim.x >= hi) return;
  for (int t = threadIdx.x; t < (W / 2) * 2 * NCO; t += blockDim.x) spoly[t] = dc.poly[t];
  for (int t = threadIdx.x; t < dc.ncheb; t += blockDim.x) scheb[t] = dc.cheb[t];
  if (threadIdx.x < 2) sflag[threadIdx.x] = 0;
  __syncthreads();
  double e = 0.0;
  int n1f = 0, n2f = 0;
  if (i >= ix1 && i < hi) {
    const double r = db.xs[i];
    double2 F1, F2;
    interp_core<W>(dc, spoly, scheb, db.fft, 2, r, s.pp.Cx, s.pp.r_hi, s.pp.tau, s.pp.theta_scale, F1, F2);
    double scale = dc.c;  // src/quadrature.jl:250-255
    double I1 = (dc.derivative ? F1.y : F1.x) * scale;
    double I2 = (dc.derivative ? F2.y : F2.x) * scale;
    e = fabs(I2 - I1);
    n1f = isnan(I1);
    n2f = isnan(I2);
    db.tmpI[i] = I2;
    db.tmpE[i] = e;
  }
  // block max over the bit patterns (e >= 0 or NaN; NaN patterns compare above +inf)
  unsigned long long bits = (unsigned long long)__double_as_longlong(e);
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(bits, off, 64);
    bits = o > bits ? o : bits;
  }
  if (__any(n1f) && (threadIdx.x & 63) == 0) atomicOr(&sflag[0], 1);
  if (__any(n2f) && (threadIdx.x & 63) == 0) atomicOr(&sflag[1], 1);
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long b = smax[0];
    for (int w2 = 1; w2 < (int)(blockDim.x >> 6); ++w2) b = smax[w2] > b ? smax[w2] : b;
    if (b) atomicMax(&s.max_err_bits, b);
    if (sflag[0]) atomicOr(&s.nan1, 1);
    if (sflag[1]) atomicOr(&s.nan2, 1);
  }
}

// ------------------------------------------------------------------ k_accum

__global__ void __launch_bounds__(256) k_accum(DevConst dc, DevBuffers db, int slot, long long n) {
  DevState& s = db.state[slot];
  if (s.done) return;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long ix1 = s.ix1, hi = s.hi;
  if ((long long)blockIdx.x * blockDim.x >= hi) return;
  const bool accept = step_accept(s, dc.tol);
  if (!accept) return;
  const bool outer_done = (s.stack_len == 0);
  bool unconv = false;
  if (i >= ix1 && i < hi) {
    double pk = db.P[i] + db.tmpI[i];
    double pe = db.E[i] + db.tmpE[i];
    if (!outer_done) {
      db.P[i] = pk;
      db.E[i] = pe;
    } else {
      db.P[i] = 0.0;
      db.E[i] = 0.0;
      db.ks[i] += pk;    // src/adaptive.jl:163-164
      db.errs[i] += pe;
      int crit = s.crit;
      double tc = 0.0, td = 0.0;
      if (crit != SK_CRIT_PANEL) {
        tail_fit(dc, s.b, s.fb, tc, td);
        if (isnan(tc) || isnan(td)) crit = SK_CRIT_PANEL;  // src/adaptive.jl:170-175
      }
      const double T2 = dc.tol * s.k0abs * 0.5;
      const double te = (crit == SK_CRIT_PANEL) ? 0.0 : trunc_est(s.b, db.xs[i], tc, td, dc.dim);
      const bool conv = (crit == SK_CRIT_PANEL || te < T2) && (crit == SK_CRIT_TAILS || fabs(pk) < T2);
      db.tmpE[i] = 2.0 * te;
      unconv = !conv;
    }
  }
  if (outer_done) {
    // highest unconverged index + 1, wave-reduced then one atomic per wave
    unsigned long long v = unconv ? (unsigned long long)(i + 1) : 0ull;
    for (int off = 32; off > 0; off >>= 1) {
      unsigned long long o = __shfl_xor(v, off, 64);
      v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v) atomicMax(&s.hi_new, v);
  }
}

__global__ void __launch_bounds__(256) k_final(DevConst dc, DevBuffers db, int slot, long long n) {
  const DevState& s = db.state[slot];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (s.done && s.retire_pending && i >= s.hi && i < s.hi_prev) db.errs[i] += db.tmpE[i];
  if (s.done == 1 && i == 0 && s.ix1 == 1) {  // src/adaptive.jl:132-146
    db.ks[0] = s.ks0;
    db.errs[0] = __builtin_nan("");
  }
}

// ------------------------------------------------------------------ launchers

static inline unsigned nblk(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_init(const DevConst& dc, const DevBuffers& db, long long n, double k0, double ks0, hipStream_t st) {
  hipLaunchKernelGGL(k_init, dim3(1), dim3(1), 0, st, dc, db, n, k0, ks0);
}
void launch_gen(const DevConst& dc, const DevBuffers& db, int slot, hipStream_t st) {
  hipLaunchKernelGGL(k_gen, dim3(nblk(3 * dc.M, 256)), dim3(256), 0, st, dc, db, slot);
}
void launch_advance(const DevConst& dc, const DevBuffers& db, int slot, hipStream_t st) {
  hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, st, dc, db, slot);
}
void launch_accum(const DevConst& dc, const DevBuffers& db, int slot, long long n, hipStream_t st) {
  hipLaunchKernelGGL(k_accum, dim3(nblk(n, 256)), dim3(256), 0, st, dc, db, slot, n);
}
void launch_final(const DevConst& dc, const DevBuffers& db, int slot, long long n, hipStream_t st) {
  hipLaunchKernelGGL(k_final, dim3(nblk(n, 256)), dim3(256), 0, st, dc, db, slot, n);
}

#define SK_DISPATCH_W(w, CALL) \
  switch (w) {                 \
    case 4: { constexpr int W = 4; CALL; } break;   \
    case 6: { constexpr int W = 6; CALL; } break;   \
    case 8: { constexpr int W = 8; CALL; } break;   \
    case 10: { constexpr int W = 10; CALL; } break; \
    case 12: { constexpr int W = 12; CALL; } break; \
    case 14: { constexpr int W = 14; CALL; } break; \
    default: { constexpr int W = 16; CALL; } break; \
  }

void launch_spread(const DevConst& dc, const DevBuffers& db, int slot, hipStream_t st) {
  SK_DISPATCH_W(dc.w, hipLaunchKernelGGL(k_spread<W>, dim3(nblk(2 * dc.n2, 256)), dim3(256), 0, st, dc, db, slot));
}
void launch_interp(const DevConst& dc, const DevBuffers& db, int slot, long long n, hipStream_t st) {
  SK_DISPATCH_W(dc.w, hipLaunchKernelGGL(k_interp<W>, dim3(nblk(n, 256)), dim3(256), 0, st, dc, db, slot, n));
}

// ------------------------------------------------------------------ seam kernels (sk_panel_sums)

__device__ __forceinline__ double es_phi_dev(double z, int w, double beta) {
  double t = 1.0 - (2.0 * z / w) * (2.0 * z / w);
  return t > 0.0 ? exp(beta * (sqrt(t) - 1.0)) : 0.0;
}

// Input-driven spread for arbitrary nodes: one thread per node, f64 atomics into the spread grid.
// The ES kernel is evaluated in closed form here (independent of the polynomial tables).
__global__ void k_seam_spread(DevConst dc, const double* nodes, const double* coef, int64_t M, double Cx, double X,
                              double kappa, double beta, double2* fft) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const double u = (nodes[j] - Cx) / X;
  const double P = u / dc.delta + 0.5 * (double)dc.n1;
  const double2 ph = cis_turns_prod(kappa, u);
  const double cr = coef[j] * ph.x, ci = coef[j] * ph.y;
  const int w = dc.w;
  long long L0 = (long long)ceil(P - 0.5 * w);
  for (int q = 0; q < w; ++q) {
    long long L = L0 + q;
    if (L < 0 || L >= dc.n1) continue;
    const double phi = es_phi_dev((double)L - P, w, beta);
    const long long l = L - dc.n1 / 2;
    const long long g = l < 0 ? l + dc.n2 : l;
    atomicAdd(&fft[g].x, cr * phi);
    atomicAdd(&fft[g].y, ci * phi);
  }
}

__global__ void k_seam_deconv(DevConst dc, double2* fft) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= dc.n2) return;
  const int64_t l = (g < dc.n2 / 2) ? g : g - dc.n2;
  if (l >= -(dc.n1 / 2) && l < dc.n1 - dc.n1 / 2) {
    const double d = dc.deconv[l < 0 ? -l : l];
    fft[g].x *= d;
    fft[g].y *= d;
  }
}

template <int W>
__global__ void __launch_bounds__(256) k_seam_interp(DevConst dc, const double2* fft, const double* xs, int64_t n,
                                                      double Cx, double r_hi, double tau, double theta_scale,
                                                      double* out_re, double* out_im) {
  constexpr int NCO = W / 2 + 2;
  __shared__ double spoly[(W / 2) * 2 * NCO];
  __shared__ double scheb[kMaxCheb];
  for (int t = threadIdx.x; t < (W / 2) * 2 * NCO; t += blockDim.x) spoly[t] = dc.poly[t];
  for (int t = threadIdx.x; t < dc.ncheb; t += blockDim.x) scheb[t] = dc.cheb[t];
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double2 F1, F2;
  interp_core<W>(dc, spoly, scheb, fft, 1, xs[i], Cx, r_hi, tau, theta_scale, F1, F2);
  out_re[i] = F1.x;
  out_im[i] = F1.y;
}

// Direct summation (src/quadrature.jl:113-127): one block per lag.
__global__ void __launch_bounds__(256) k_direct(const double* nodes, const double* coef, int64_t M, const double* xs,
                                                 int64_t n, double* out_re, double* out_im) {
  __shared__ double sr[4], si[4];
  const int64_t i = blockIdx.x;
  if (i >= n) return;
  const double x = xs[i];
  double ar = 0.0, ai = 0.0;
  for (int64_t j = threadIdx.x; j < M; j += blockDim.x) {
    const double2 ph = cis_turns_prod(nodes[j], x);
    const double cj = coef[j];
    ar = fma(cj, ph.x, ar);
    ai = fma(cj, ph.y, ai);
  }
  for (int off = 32; off > 0; off >>= 1) {
    ar += __shfl_xor(ar, off, 64);
    ai += __shfl_xor(ai, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { sr[threadIdx.x >> 6] = ar; si[threadIdx.x >> 6] = ai; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out_re[i] = sr[0] + sr[1] + sr[2] + sr[3];
    out_im[i] = si[0] + si[1] + si[2] + si[3];
  }
}

void launch_seam_spread(const DevConst& dc, const double* nodes, const double* coef, int64_t M, double Cx, double X,
                        double kappa, double2* fft, hipStream_t st) {
  double beta = 2.30 * dc.w;
  if (dc.w == 4) beta = 2.38 * dc.w;
  hipLaunchKernelGGL(k_seam_spread, dim3(nblk(M, 256)), dim3(256), 0, st, dc, nodes, coef, M, Cx, X, kappa, beta, fft);
}
void launch_seam_deconv(const DevConst& dc, double2* fft, hipStream_t st) {
  hipLaunchKernelGGL(k_seam_deconv, dim3(nblk(dc.n2, 256)), dim3(256), 0, st, dc, fft);
}
void launch_seam_interp(const DevConst& dc, const double2* fft, const double* xs, int64_t n, double Cx, double r_hi,
                        double tau, double theta_scale, double* out_re, double* out_im, hipStream_t st) {
  SK_DISPATCH_W(dc.w, hipLaunchKernelGGL(k_seam_interp<W>, dim3(nblk(n, 256)), dim3(256), 0, st, dc, fft, xs, n, Cx,
                                         r_hi, tau, theta_scale, out_re, out_im));
}
void launch_direct(const double* nodes, const double* coef, int64_t M, const double* xs, int64_t n, double* out_re,
                   double* out_im, hipStream_t st) {
  hipLaunchKernelGGL(k_direct, dim3((unsigned)n), dim3(256), 0, st, nodes, coef, M, xs, n, out_re, out_im);
}

// ------------------------------------------------------------------ sort / dedup (src/adaptive.jl:99-120)

__global__ void k_check_sorted(const double* xs, long long n, int* flags) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = xs[i];
  bool bad = !(x >= 0.0);  // negative or NaN
  bool unsorted = (i + 1 < n) && !(x < xs[i + 1]);
  if (__any(unsorted) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1);
}
void launch_check_sorted(const double* xs, long long n, int* flags, hipStream_t st) {
  hipLaunchKernelGGL(k_check_sorted, dim3(nblk(n, 256)), dim3(256), 0, st, xs, n, flags);
}

__global__ void k_make_keys(const double* xs, long long n, unsigned long long* keys, long long* idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = xs[i];
  keys[i] = (x == 0.0) ? 0ull : (unsigned long long)__double_as_longlong(x);  // -0.0 == 0.0
  idx[i] = i;
}
__global__ void k_heads(const unsigned long long* keys, long long n, long long* head) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
__global__ void k_scatter_unique(const unsigned long long* keys, const long long* perm, const long long* uid1,
                                 long long n, double* uxs, long long* inv, long long* n_unique) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long u = uid1[i] - 1;
  if (i == 0 || uid1[i] != uid1[i - 1]) uxs[u] = __longlong_as_double((long long)keys[i]);
  inv[perm[i]] = u;
  if (i == n - 1) *n_unique = uid1[i];
}
__global__ void k_gather(const double* ks, const double* errs, const long long* inv, long long n, double* vals,
                         double* eo) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long u = inv[i];
  vals[i] = ks[u];
  eo[i] = errs[u];
}

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_unique_temp_bytes(long long n) {
  size_t sort_b = 0, scan_b = 0;
  unsigned long long* k = nullptr;
  long long* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, sort_b, k, k, v, v, (size_t)n, 0, 64, 0);
  (void)rocprim::inclusive_scan(nullptr, scan_b, v, v, (size_t)n, rocprim::plus<long long>(), 0);
  size_t lib = sort_b > scan_b ? sort_b : scan_b;
  return align_up(lib) + 6 * align_up((size_t)n * 8) + 256;
}

void sort_unique(const double* xs, long long n, double* uxs, long long* inv, long long* d_n_unique, void* temp,
                 size_t temp_bytes, hipStream_t st) {
  char* base = (char*)temp;
  const size_t arr = align_up((size_t)n * 8);
  unsigned long long* keys_in = (unsigned long long*)base; base += arr;
  unsigned long long* keys_out = (unsigned long long*)base; base += arr;
  long long* idx_in = (long long*)base; base += arr;
  long long* idx_out = (long long*)base; base += arr;
  long long* head = (long long*)base; base += arr;
  long long* uid1 = (long long*)base; base += arr;
  void* lib = base;
  size_t lib_bytes = temp_bytes - 6 * arr;
  hipLaunchKernelGGL(k_make_keys, dim3(nblk(n, 256)), dim3(256), 0, st, xs, n, keys_in, idx_in);
  size_t sb = lib_bytes;
  (void)rocprim::radix_sort_pairs(lib, sb, keys_in, keys_out, idx_in, idx_out, (size_t)n, 0, 64, st);
  hipLaunchKernelGGL(k_heads, dim3(nblk(n, 256)), dim3(256), 0, st, keys_out, n, head);
  sb = lib_bytes;
  (void)rocprim::inclusive_scan(lib, sb, head, uid1, (size_t)n, rocprim::plus<long long>(), st);
  hipLaunchKernelGGL(k_scatter_unique, dim3(nblk(n, 256)), dim3(256), 0, st, keys_out, idx_out, uid1, n, uxs, inv,
                     d_n_unique);
}

void launch_gather(const double* ks, const double* errs, const long long* inv, long long n, double* vals, double* eo,
                   hipStream_t st) {
  hipLaunchKernelGGL(k_gather, dim3(nblk(n, 256)), dim3(256), 0, st, ks, errs, inv, n, vals, eo);
}

}  // namespace sk
