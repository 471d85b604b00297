// sk_internal.h -- shared host/device declarations of the kernel_values hot path.
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include "../../include/spectralkernels_hip.h"

namespace sk {

// ---------------------------------------------------------------- host numerics
// Gauss-Jacobi rule for weight (1-x)^0 (1+x)^p on [-1,1], ascending nodes; p == 0 gives
// Gauss-Legendre.  Counterpart of FastGaussQuadrature's gausslegendre / gaussjacobi(n,0,p)
// used at src/quadrature.jl:37-42.
void gauss_rule(int64_t n, double p, std::vector<double>& nodes, std::vector<double>& weights);

// SDF catalogue on the host (same formulas as the device catalogue in sk_kernels.hip).
double sdf_eval_host(int kind, const double* prm, double w);

struct SdfHost {
  int kind;
  double prm[8];
  sk_sdf_batch_fn f;
  void* ctx;
  double operator()(double w) const;
  void batch(const double* w, double* out, size_t n) const;
};

// compute_k0 (src/adaptive.jl:74-91): adaptive Gauss-Kronrod on [0, inf).
double compute_k0_host(const SdfHost& f, int dim, double p, double c, int derivative, int logw, double tol,
                       double t_lo = 0.0, double t_hi = 1.0);

// ES ("exponential of semicircle") spreading kernel tables.
struct EsTables {
  int w;          // kernel width in grid points (even)
  double beta;    // shape parameter
  int nco;        // coefficients per half polynomial: E and O have nco each (degree 2*nco-1)
  // pair q < w/2: g_q(y) = phi(q-(w-1)/2+y) = E_q(y^2) + y O_q(y^2), y in [-1/2,1/2];
  // tap q uses E+yO, tap w-1-q uses E-yO.  Layout: [q][0..nco) = E, [q][nco..2nco) = O,
  // highest power first (Horner order).
  std::vector<double> poly;
  int ncheb;      // Chebyshev terms of 1/phi_hat(theta) as a function of s = 2(theta/theta_max)^2-1
  double theta_max;
  std::vector<double> cheb;
  // 1/phi_hat(2 pi l / n2), l = 0..n1/2
  std::vector<double> deconv;
};
double es_phi(double z, int w, double beta);
double es_phi_hat(double xi, int w, double beta);
void build_es_tables(int w, int64_t n1, int64_t n2, double theta_max, EsTables& t);

// ---------------------------------------------------------------- host threads
// CPUs this process may really use: the smaller of its affinity mask and its cgroup CPU quota (cpu.max / cfs_quota_us),
// at most `cap`.  A container that shows 256 logical CPUs under a quota of 16 throttles a 256-thread OpenMP team --
// and, while its workers spin out their block time, everything else in the process -- for the rest of each 100 ms
// scheduler period (measured: config creation 100-300 ms instead of 15, one call in eight stalled for 90 ms).
int host_threads(int cap);

// ---------------------------------------------------------------- error channel
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

}  // namespace sk
