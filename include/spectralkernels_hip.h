/*
 * spectralkernels_hip.h -- C ABI of the MI355X-native kernel_values hot path.
 *
 * Drop-in boundary for SpectralKernels.jl's exported API (src/SpectralKernels.jl:12):
 *   AdaptiveKernelConfig(f; kwargs...)           src/adaptive.jl:24-59
 *   compute_k0(config)                           src/adaptive.jl:74-91
 *   kernel_values(config, xs; k0, param_derivative, verbose) -> (vals, errs)
 *                                                src/adaptive.jl:95-108
 * and, for tests, the narrowest internal seam finufft1d3(w, s, x) (src/utils.jl:10, called at
 * src/quadrature.jl:109-110).  The reference has no FFI of its own for this path (its only
 * foreign hop is FINUFFT.jl -> libfinufft); a Julia maintainer binds these symbols with
 * `ccall` -- see INTEGRATION.md and spectralkernels.jl_amd/julia/SpectralKernelsHIP.jl.
 *
 * Conventions: plain pointers and sizes; all floating point is IEEE double; every function
 * returns an sk_status (0 = OK); sk_last_error() gives the message the reference would have
 * thrown.  A config handle owns its device buffers and is NOT re-entrant (the reference's
 * config shares mutable scratch too, src/adaptive.jl:17-21).  The library owns all device
 * memory except in the *_device entry, which takes HBM-resident arrays.
 */
#ifndef SPECTRALKERNELS_HIP_H
#define SPECTRALKERNELS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SK_OK = 0,
  SK_EINVAL = 1,       /* ctor checks: criteria, alpha >= dim, p <= -1 (src/adaptive.jl:29-35, src/quadrature.jl:40) */
  SK_ESUBDIVIDE = 2,   /* "(sub)interval split too many times (b - a < 1e-16)" (src/utils.jl:28-36) */
  SK_ENAN = 3,         /* "NaN detected in panel integral" (src/quadrature.jl:165) */
  SK_ESTACK = 4,       /* splitting stack capacity 5000 exceeded (src/quadrature.jl:7-15) */
  SK_EHIP = 5,         /* HIP / rocFFT runtime failure */
  SK_ENODEVICE = 6,    /* no gfx950 device visible: the product path has no CPU fallback */
  SK_EUNSUPPORTED = 7, /* option outside the implemented scope (logw with dim > 2, as in the reference; logw with derivative) */
  SK_WARN_QUADSPEC = 100 /* tol < 1e-12 shrank quadspec to (2^12, 1) (src/adaptive.jl:37-40); config is valid */
} sk_status;

/* SDF catalogue evaluated on the device.  SK_SDF_CALLBACK evaluates a host closure in
 * batches on the caller's thread (one batch of 3*m*k nodes per (sub)panel). */
typedef enum {
  SK_SDF_CALLBACK = 0,
  SK_SDF_MATERN = 1,      /* params {phi, rho, nu, d}: phi*(rho^2+w^2)^(-nu-d/2)   scripts/matern_pair.jl:17 */
  SK_SDF_EXPONENTIAL = 2, /* params {a}: exp(-a*|w|)                            test/exponential_sdf_1d.jl:3 */
  SK_SDF_GAUSSIAN = 3     /* params {a}: exp(-(w/a)^2)  (super-algebraic tail -> :panel fallback) */
} sk_sdf_kind;

typedef enum { SK_CRIT_BOTH = 0, SK_CRIT_PANEL = 1, SK_CRIT_TAILS = 2 } sk_criteria;

/* Batched SDF callback: out[i] = S(w[i]), i < n.  Called only on the thread that called
 * sk_kernel_values / sk_compute_k0. */
typedef void (*sk_sdf_batch_fn)(const double* w, double* out, size_t n, void* ctx);

/* Keyword arguments of AdaptiveKernelConfig (src/adaptive.jl:24-27), same meaning. */
typedef struct {
  int sdf_kind;            /* sk_sdf_kind */
  double sdf_params[8];
  sk_sdf_batch_fn f;       /* used iff sdf_kind == SK_SDF_CALLBACK */
  void* f_ctx;
  sk_sdf_batch_fn df;      /* derivative of the SDF: callback SDFs with logw=1 (src/quadrature.jl:194) */
  void* df_ctx;
  int dim;                 /* 1 (cos/sin kernel); even dims: Bessel kernels (fast Hankel transform or direct sums) */
  double alpha;            /* |w|^-alpha singularity at the origin */
  double tol;              /* relative to K(0) */
  int derivative;          /* K'(r) */
  int logw;
  int criteria;            /* sk_criteria */
  int has_tail;
  double tail;             /* power-law exponent override */
  int64_t m, k;            /* quadspec: k sub-panels of m (and 2m) Gauss nodes per NUFFT */
  int device;              /* HIP device ordinal; -1 = current device */
  int nufft_width;         /* 0 = choose from tol; else spreading width w in [4,16] */
  int hankel_min_lags;     /* dim >= 2: use the fast Hankel transform for >= this many unique lags, the direct
                              Bessel-sum kernel below (0 = default 8; -1 = always direct) */
  int sdf_dparam;          /* catalogue SDFs: 0 = the SDF itself; j >= 1 = its derivative with respect to
                              sdf_params[j-1] -- the integrands of kernel_sdf_derivatives (src/derivatives.jl:61-72),
                              evaluated on the device.  Matern: phi, rho, nu; exponential, gaussian: a. */
  int batch_panels;        /* device-resident dim-1 march: most panels evaluated per launch triple (consecutive panels
                              of equal width share the NUFFT geometry); 0 = choose from the rule size, else 1, 2, 4, 8 */
} sk_config_args;

typedef struct sk_config sk_config; /* opaque */

/* Fills the defaults of src/adaptive.jl:24-27: dim=1, alpha=0, tol=1e-8, :both, quadspec=(4096,16). */
void sk_config_args_default(sk_config_args* args);

/* AdaptiveKernelConfig(...) -- src/adaptive.jl:24-59.  Returns SK_WARN_QUADSPEC with a valid
 * handle when the tol<1e-12 rule applies. */
int sk_config_create(sk_config** out, const sk_config_args* args);
void sk_config_destroy(sk_config* cfg);

/* Read back derived constants (src/adaptive.jl:42-45): p, c, effective (m,k). */
int sk_config_info(const sk_config* cfg, double* p, double* c, int64_t* m, int64_t* k, int* nufft_width);

/* Copies the config's Gauss rules (QuadRule, src/quadrature.jl:27-47) to the caller:
 * which = 0 Legendre, 1 Jacobi(0,p); no1/wt1 have m entries, no2/wt2 2m.  Host-only. */
int sk_config_rules(const sk_config* cfg, int which, double* no1, double* wt1, double* no2, double* wt2);

/* compute_k0(config) -- src/adaptive.jl:74-91.  Host quadrature (adaptive Gauss-Kronrod). */
int sk_compute_k0(sk_config* cfg, double* k0);

/* Statistics of the last sk_kernel_values* call on this config (panel schedule). */
typedef struct {
  int64_t n_unique;      /* targets after dedup */
  int64_t n_outer;       /* outer panels marched (src/adaptive.jl:149 loop iterations) */
  int64_t n_panels;      /* (sub)panel evaluations = NUFFT pairs Q */
  int64_t n_accepted;
  int64_t sum_active;    /* sum over panel evaluations of active targets (a-bar * N) */
  int64_t n_steps_launched; /* device steps enqueued incl. speculative ones */
  int64_t n_syncs;       /* host<->device synchronisations */
  double last_b;         /* upper end of the last panel */
  int64_t n_steps;       /* device steps whose panels were applied (a step evaluates 1..batch_panels panels) */
  int64_t n_rerun;       /* steps discarded and re-run with fewer panels (a later panel of the batch was rejected) */
  int64_t sum_active_steps; /* sum over applied steps of active targets: each step reads and writes a lag's sums once */
  int64_t n_fix;         /* steps whose guessed run was cut short by a retirement at its first panel (outputs of the retired
                            lags taken from that panel's sums by the next launch) */
} sk_stats;
int sk_last_stats(const sk_config* cfg, sk_stats* out);

/* kernel_values(config, xs; k0, param_derivative) -- src/adaptive.jl:95-108.
 * xs: host array, any order, duplicates allowed, entries >= 0.  vals/errs: caller-allocated
 * host arrays of length n, input order; errs is NaN at xs == 0 (src/adaptive.jl:137-144).
 * has_k0 == 0 computes k0 with sk_compute_k0 first. */
int sk_kernel_values(sk_config* cfg, const double* xs, size_t n, double k0, int has_k0, int param_derivative,
                     double* vals_out, double* errs_out);
/* Wall-clock split of the last sk_kernel_values call on this config, in ms: [0] staged upload of the lags
 * (enqueue side), [1] march, [2] download into the caller's arrays, [3] whole call. */
int sk_last_host_timing(const sk_config* cfg, double* ms4);
/* Automatic batch width, last march on this config: [0] steps launched with the one-panel kernels inside a full-width march
 * (while the march retires lags every step is one fresh outer panel, src/adaptive.jl:149-199: the lighter instantiations
 * evaluate it with the same bits), [1] steps those launches skipped and the host launched again at the full width. */
int sk_last_step_kinds(const sk_config* cfg, long long* out2);
/* Host threads the library's teams are sized by: the smaller of this process's CPU affinity mask and its cgroup CPU
 * quota (cpu.max / cfs_quota_us), at most `cap` (cap <= 0: no cap).  The staging copies, rule builders and shard scans
 * use min(8, that); an OpenMP team sized by the CPUs a container merely shows is throttled under a quota. */
int sk_host_threads(int cap);

/* Memory owned by the library's HIP runtime, so that a caller (the Julia shim, tests, bench.py) needs no HIP
 * binding of its own.  sk_host_alloc returns PINNED host memory: arrays that live in it are copied by DMA
 * directly, without the staging pass that pageable arrays need (sk_kernel_values detects this per pointer).
 * sk_memcpy_* are synchronous; device < 0 = current device. */
int sk_device_count(int* n);
int sk_device_malloc(void** d_ptr, size_t bytes, int device);
int sk_device_free(void* d_ptr);
int sk_memcpy_h2d(void* d_dst, const void* src, size_t bytes);
int sk_memcpy_d2h(void* dst, const void* d_src, size_t bytes);
int sk_device_synchronize(int device);
int sk_device_mem_info(size_t* free_bytes, size_t* total_bytes); /* current device */
int sk_host_alloc(void** ptr, size_t bytes);
int sk_host_free(void* ptr);

/* Same computation on HBM-resident arrays: d_xs, d_vals, d_errs are device pointers on the
 * config's device; stream is a hipStream_t (NULL = default stream).  Any order / duplicates
 * allowed (device sort + dedup); already strictly increasing input skips that stage. The call
 * returns after the results are complete on `stream` (it synchronises the stream). */
int sk_kernel_values_device(sk_config* cfg, const double* d_xs, size_t n, double k0, int param_derivative,
                            double* d_vals, double* d_errs, void* stream);

/* Several configs on ONE lag set -- the Jacobian pattern of src/derivatives.jl:61-112 (one kernel_values call per
 * dS/dtheta_j integrand, one with derivative=true for the warping gradients, one with logw=true for d/dalpha, all
 * at the same warp_lags).  The lags are uploaded, sorted and deduplicated once on the device; every config then
 * marches on the sorted unique set with its own integrand and the results are scattered back per config.  The marches
 * run side by side (one host thread and the config's own stream each) when all SDFs are catalogue entries and the configs
 * are distinct; SK_MULTI_SERIAL=1 runs them one after the other -- same bits.
 * cfgs: ncfg configs on the same device; k0s / param_derivative: one entry per config (as in sk_kernel_values
 * with has_k0 = 1); vals_out / errs_out: ncfg x n, row c = config c in input order. */
int sk_kernel_values_multi(sk_config* const* cfgs, size_t ncfg, const double* xs, size_t n, const double* k0s,
                           const int* param_derivative, double* vals_out, double* errs_out);

/* The same integrands in ONE joint panel march: shared nodes, spreading weights, tap positions,
 * interpolation weights and post-phases; per integrand only the source strengths (one batched type-3 NUFFT with
 * one transform per integrand, two for logw) and the running sums.  A (sub)panel is accepted when every integrand
 * passes max|I2-I1| < tol |k0| (src/quadrature.jl:260), a lag retires when every integrand's convergence predicate
 * holds (src/adaptive.jl:183-198): results within tol of the separate marches of sk_kernel_values_multi, whose
 * schedules can only be shorter.  cfgs[0] must be the plain kernel (derivative = logw = 0); all configs: one dim,
 * equal quadspec / tol / alpha / criteria / tail, at most one logw integrand; one k0 for all, as the reference passes
 * it (src/derivatives.jl:57,71,81).  dim 1: type-3 NUFFT with one transform per integrand; even dim >= 2: below
 * hankel_min_lags (8 by default) lags the direct Bessel sums of src/quadrature.jl:145-160 with J_nu(2 pi w r) evaluated
 * once per (lag, node) for all integrands; from hankel_min_lags on this entry and sk_kernel_jacobian_pairs are NOT a joint
 * march: they run one fast-Hankel march per integrand (separate schedules, results bit-identical to separate
 * sk_kernel_values calls), side by side on the configs' own streams.  (A joint march through the Hankel kernels exists
 * for dim 2 -- SK_JOINT_HK=1 -- and is slower: its schedule is the union of the integrands' bisection trees)
 * SDFs: all catalogue entries
 * (incl. sdf_dparam), evaluated on the device, or all host callbacks, each called once per panel on all its nodes
 * (no logw integrand then).  vals_out / errs_out: ncfg x n. */
int sk_kernel_values_joint(sk_config* const* cfgs, size_t ncfg, const double* xs, size_t n, double k0,
                           const int* param_derivative, double* vals_out, double* errs_out);

/* gen_kernel + gen_kernel_jacobian for point pairs on the device (src/model.jl:53-77, src/derivatives.jl:86-112):
 * the lags norm(pts[j] - pts[k]) of the pairs are formed on the device (pts: the WARPED points, npts x pdim row-major;
 * pairs: npairs x 2 zero-based), ONE joint march evaluates every integrand, and the Jacobian is assembled on the
 * device.  role[t] of cfgs[t]: 0 = the kernel itself (cfgs[0]; vals_out, npairs values in pair order -- the values of
 * the reference's lookup table Dict(raw_pairs => values)); 1 = Jacobian column col[t] takes this integrand
 * (SDF-parameter derivatives, the logw integrand; evaluated with param_derivative = true as at :68,80); 2 = K'
 * (derivative = true): column warp_cols[q] takes K'(lag_e) * dlag[e * nwarp + q], dlag = d lag_e / d warp-parameter q
 * from the caller's differentiation of the warp (:46-54).  jac_out: npairs x ncols row-major; columns nobody claims
 * are zero.  Config requirements as for sk_kernel_values_joint. */
int sk_kernel_jacobian_pairs(sk_config* const* cfgs, size_t ncfg, const int* role, const int* col, const double* pts,
                             size_t npts, int pdim, const int64_t* pairs, size_t npairs, double k0, const double* dlag,
                             int nwarp, const int* warp_cols, int ncols, double* vals_out, double* jac_out);

/* One process, several GPUs (the reference's kernel_values is one call on one host Vector,
 * src/adaptive.jl:95-98): the sorted unique lag set is cut into contiguous blocks, one per device; every device
 * holds a replica of the config and marches its own block (its panel widths follow its own largest lag,
 * src/adaptive.jl:152) with no exchange after K(0), which is computed once and shared.  Each shard reads its
 * block of `xs` and writes its block of vals/errs in place.  devices == NULL: devices 0..ndev-1.
 * Same contract as sk_kernel_values (any order, duplicates allowed; results within tol of the single-device
 * call -- the panel schedules of the shards differ). */
typedef struct sk_sharded sk_sharded; /* opaque */
int sk_sharded_create(sk_sharded** out, const sk_config_args* args, const int* devices, int ndev);
void sk_sharded_destroy(sk_sharded* sh);
int sk_sharded_count(const sk_sharded* sh);
int sk_kernel_values_sharded(sk_sharded* sh, const double* xs, size_t n, double k0, int has_k0, int param_derivative,
                             double* vals_out, double* errs_out);
/* Split of the sorted unique lag set over the shards: 0 = contiguous blocks (default), 1 = strided (shard s takes lags
 * s, s + ns, ...: every shard sees the whole lag range, hence the same panel schedule and equal work -- the blocks'
 * schedules differ, the block with the smallest lags marches longest). */
int sk_sharded_set_split(sk_sharded* sh, int strided);
/* kernel_values on the lags of point pairs (sk_kernel_values_pairs: src/model.jl:64-65, src/utils.jl:41-64), sharded:
 * the points are replicated, the pair list (or the dense list "0 followed by all i < j" for pairs == NULL) is cut into
 * contiguous blocks, one per device; every device forms the lags of its block, sorts and deduplicates them and marches
 * on them -- no lag array and no sort on the host.  vals/errs in pair order.  BASELINE config 4: 10^6 pairwise lags
 * across the GPUs of a node in one call. */
int sk_kernel_values_pairs_sharded(sk_sharded* sh, const double* pts, size_t npts, int pdim, const int64_t* pairs,
                                   size_t npairs, double k0, int has_k0, int param_derivative, double* vals_out,
                                   double* errs_out);
/* Last call, per shard: panel statistics, wall time (ms) and block [lo, hi) of the sorted unique set (strided: first
 * index and the set's size; pairs: block of the pair list). */
int sk_sharded_last(const sk_sharded* sh, int shard, sk_stats* st, double* ms, int64_t* lo, int64_t* hi);
/* Test hook (CPU tier): shards whose device call is replaced by `fn` (same meaning as sk_kernel_values with
 * has_k0 = 1 on the block) -- exercises the split / dedup / merge logic without a device. */
typedef int (*sk_shard_fn)(int shard, const double* xs, size_t n, double k0, int param_derivative, double* vals,
                           double* errs, void* ctx);
int sk_sharded_create_injected(sk_sharded** out, int nshard, sk_shard_fn fn, void* ctx);

/* Callers' side of the path: lags from point pairs, generated on the device so that an n-point set never
 * materialises its n(n-1)/2 lags on the host.
 *   src/model.jl:64-65   warp_lags = [norm(warp_pts[j] - warp_pts[k]) for (j,k) in kernel_index_pairs]
 *   src/utils.jl:43-46   xs = [0.0; |pts[i]-pts[j]| for i < j]             (pairs == NULL)
 * pts: npts x pdim row-major coordinates; pairs: npairs x 2 zero-based indices, or NULL for the dense list
 * "0 followed by all i < j" (then npairs is ignored and the output length is 1 + npts(npts-1)/2).
 * vals/errs in pair order, as kernel_values on those lags would return them (src/model.jl:76). */
int sk_kernel_values_pairs(sk_config* cfg, const double* pts, size_t npts, int pdim, const int64_t* pairs, size_t npairs,
                           double k0, int has_k0, int param_derivative, double* vals_out, double* errs_out);
/* the same on HBM-resident arrays (d_pts, d_pairs, d_vals, d_errs device pointers; d_pairs may be NULL) */
int sk_kernel_values_pairs_device(sk_config* cfg, const double* d_pts, size_t npts, int pdim, const int64_t* d_pairs,
                                  size_t npairs, double k0, int param_derivative, double* d_vals, double* d_errs,
                                  void* stream);
/* build_dense_cov_matrix(cfg, pts) -- src/utils.jl:41-64: the symmetric npt x npt matrix K(|p_i - p_j|), K(0) =
 * compute_k0 on the diagonal; out: npt*npt doubles (host).  sk_dense_cov_matrix_device: d_pts, d_out in HBM. */
int sk_dense_cov_matrix(sk_config* cfg, const double* pts, size_t npts, int pdim, double* out);
int sk_dense_cov_matrix_device(sk_config* cfg, const double* d_pts, size_t npts, int pdim, double* d_out, void* stream);

/* finufft1d3-level seam (src/utils.jl:10; src/quadrature.jl:109-110): for real coefficients
 * out_re/out_im[i] = sum_j coef[j] * exp(2*pi*i * nodes[j] * xs[i]).  Host pointers.
 * method: 0 = type-3 NUFFT kernels (nodes must lie in [lo, hi], xs >= 0), 1 = direct-sum kernel. */
int sk_panel_sums(sk_config* cfg, const double* nodes, const double* coef, size_t M, double lo, double hi,
                  const double* xs, size_t n, int method, double* out_re, double* out_im);


/* Per-kernel device timing of the hot path, measured with HIP events on the stream the kernels
 * are launched on.  Off by default (events add host overhead); when enabled every launch of the
 * next sk_kernel_values* calls is bracketed by two events.  Kernel ids (SK_K_GEN has no launches on the
 * device-resident dim-1 path, where node generation is fused into the spread launch; SK_K_ADVANCE is the
 * once-per-round status kernel; a config is not thread-safe: one call at a time per sk_config): */
enum { SK_K_GEN = 0, SK_K_SPREAD = 1, SK_K_FFT = 2, SK_K_INTERP = 3, SK_K_ADVANCE = 4,
       SK_K_NEAR = 5, /* dim >= 2: near-field Bessel sums at the Chebyshev nodes (k_hk_near + k_hk_cheb) */
       SK_K_COUNT = 6 };
typedef struct {
  double total_ms[SK_K_COUNT]; /* summed over the launches of the last call */
  int64_t launches[SK_K_COUNT];
  int64_t active_items[SK_K_COUNT]; /* summed work items: nodes (gen), grid points (spread, fft), active lags (interp), panels (advance) */
} sk_kernel_times;
/* What a launch costs before it does any work, in us, with the same event brackets: [0] an empty kernel on the
 * fused spread launch's grid, [1] that launch reduced to its step-state prologue, [2] k_interp reduced to its state
 * loads, [3] the empty kernel launched back to back (one bracket around `reps` launches, per launch).  Not to be
 * called while another call on this config is in flight. */
int sk_launch_floor(sk_config* cfg, int reps, double* us4);
int sk_profile_enable(sk_config* cfg, int on);
int sk_profile_read(const sk_config* cfg, sk_kernel_times* out);

/* Up to `cap` (a, b, max_err, n_active, accepted) records of the last call, in evaluation order.
 * rec is cap x 5 doubles, row-major.  Returns the number of records written in *len. */
int sk_last_trace(const sk_config* cfg, double* rec, size_t cap, size_t* len);

/* Host-only hooks (no device needed) used by the CPU test tier: the config's Gauss rule
 * generator (counterpart of FastGaussQuadrature at src/quadrature.jl:37-42), compute_k0 on the
 * host (src/adaptive.jl:74-91), constructor validation (src/adaptive.jl:29-40) and the ES
 * spreading kernel: closed form phi(z) and the piecewise-polynomial value used on the device. */
int sk_host_gauss_rule(int64_t n, double p, double* nodes, double* weights);
int sk_host_compute_k0(const sk_config_args* args, double* k0);
/* Partial K(0): the same integral restricted to the mapped variable t in [t_lo, t_hi) of [0,1),
 * w = t/(1-t); the parts of a partition of [0,1) sum to sk_host_compute_k0 (multi-GPU set-up). */
int sk_host_compute_k0_part(const sk_config_args* args, double t_lo, double t_hi, double* part);
int sk_host_validate(const sk_config_args* args, int64_t* m, int64_t* k);
int sk_host_es_eval(int w, double z, double* phi, double* phi_poly);
/* The staging pass of the host-array upload (chunks staged by `threads` host threads, order of the lags checked in
 * the same pass -- src/adaptive.jl:99,113) on plain memory: stage receives the copy; *order = 1 strictly increasing,
 * 2 not, 3 a negative / NaN lag.  threads <= 0: the library's default (SK_COPY_THREADS). */
int sk_host_stage_scan(const double* src, size_t cnt, double* stage, int threads, int* order);

/* Layout of the structs that cross this boundary, for bindings that mirror them by hand (the Julia shim, ctypes):
 * which = 0 sk_config_args, 1 sk_stats, 2 sk_kernel_times.  out receives sizeof(struct) followed by the byte offset of
 * every field in declaration order (at most cap entries); *len = number of entries of the full description. */
int sk_abi_layout(int which, size_t* out, size_t cap, size_t* len);

const char* sk_last_error(void);
const char* sk_version(void);
/* 1 iff a gfx950 device is usable by this process. */
int sk_device_available(void);

#ifdef __cplusplus
}
#endif
#endif
