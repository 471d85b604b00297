// sk_joint.cpp -- sk_kernel_values_joint: several integrands on one lag set in ONE panel march.
// Host driver of the kernels in sk_joint_kernels.inc; the schedule rule and what is shared are described there.
// Counterpart of the call pattern of src/derivatives.jl:51-112 (gen_kernel_jacobian): kernel_sdf_derivatives (one
// kernel_values per dS/dtheta_j), kernel_warping_gradients (derivative=true), kernel_singularity_derivative
// (logw=true), all on the same warp_lags with the same k0.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "sk_config.h"
#include <string>
#include <vector>
#include <cstdio>

namespace sk {
int joint_wait_status(sk_config* cfg, unsigned int seq, hipStream_t st);  // sk_api.cpp (wait_status)
void joint_host_nodes(const sk_config* cfg, const PanelParams& pp, std::vector<double>& w);  // sk_api.cpp (host_nodes)
}

namespace {

int ensure_dev_buf(void** buf, size_t* have, size_t bytes) {
  if (bytes <= *have) return SK_OK;
  if (*buf) HIPCHK(hipFree(*buf));
  *buf = nullptr;
  *have = 0;
  HIPCHK(hipMalloc(buf, bytes + bytes / 8));
  *have = bytes + bytes / 8;
  return SK_OK;
}

size_t a256(size_t b) { return (b + 255) & ~(size_t)255; }

// joint_core: this lag set would take the fast Hankel transform but a step does not fit its buffers (or the route is
// switched off): the caller runs one march per integrand instead (sk_kernel_values_multi / multi_device_core)
constexpr int kJointHkFallback = 1000;
// dim 2, catalogue SDFs, lag sets of hankel_min_lags and more: the joint march through the Hankel kernels (round 4).
// Opt-in (SK_JOINT_HK=1): the joint schedule is the union of the integrands' bisection trees -- a (sub)panel is accepted
// when EVERY integrand passes -- and in dim 2 a panel evaluation is expensive whatever the number of lags (spread + FFT of
// the default rule).  On the nll_2d integrands at 10^5 lags the joint march takes 5 panels x 4 transforms where the four
// separate marches take 2 + 2 + 2 + 5 panel evaluations (K' alone needs the five): 3.2-7.3 ms against 2.1 ms for the
// separate marches side by side (tests/test_gpu_joint.py).  The default therefore stays one fast-Hankel march per
// integrand, overlapped; integrand sets with equal schedules (K and parameter derivatives only) are where this route pays.
bool joint_hankel_wanted(const sk_config* c0, size_t n) {
  const char* e = getenv("SK_JOINT_HK");
  if (!e || e[0] != '1') return false;
  const int hk_min = c0->args.hankel_min_lags == 0 ? kHkDefaultMinLags : c0->args.hankel_min_lags;
  return c0->args.dim == 2 && c0->args.sdf_kind != SK_SDF_CALLBACK && hk_min > 0 && n >= (size_t)hk_min && c0->hk_nb > 0;
}

}  // namespace

namespace {

// What a joint call accepts, whichever way it is then evaluated (joint march, or one fast-Hankel march per integrand for
// the larger lag sets in dim >= 2): the same inputs are rejected with the same codes at every size.
int joint_check_args(sk_config* const* cfgs, size_t ncfg) {
  if (!cfgs || ncfg == 0 || !cfgs[0]) return fail(SK_EINVAL, "null config");
  if (ncfg > (size_t)kJointMax) return fail(SK_EINVAL, "at most 8 integrands per joint march");
  const sk_config* c0 = cfgs[0];
  if (c0->args.derivative || c0->args.logw)
    return fail(SK_EINVAL, "joint march: the first config must be the plain kernel (it provides the shared nodes)");
  int n_logw = 0, ntr = 0;
  for (size_t t = 0; t < ncfg; ++t) {
    const sk_config* c = cfgs[t];
    if (!c) return fail(SK_EINVAL, "null config");
    const sk_config_args& a = c->args;
    if (c->device != c0->device) return fail(SK_EINVAL, "joint march: configs must live on one device");
    if (a.dim != c0->args.dim) return fail(SK_EINVAL, "joint march: the configs must have one dim");
    // catalogue SDFs are evaluated on the device; host callbacks once per panel for all integrands -- not mixed
    if ((a.sdf_kind == SK_SDF_CALLBACK) != (c0->args.sdf_kind == SK_SDF_CALLBACK))
      return fail(SK_EUNSUPPORTED, "joint march: all integrands catalogue SDFs or all host callbacks");
    if (a.sdf_kind == SK_SDF_CALLBACK && a.logw)
      return fail(SK_EUNSUPPORTED, "joint march: logw with a callback SDF (use sk_kernel_values_multi)");
    if (c->m != c0->m || c->k != c0->k || a.tol != c0->args.tol || a.alpha != c0->args.alpha || a.criteria != c0->args.criteria ||
        a.has_tail != c0->args.has_tail || (a.has_tail && a.tail != c0->args.tail) || c->dc.w != c0->dc.w)
      return fail(SK_EINVAL, "joint march: quadspec, tol, alpha, convergence_criteria, tail and NUFFT width must agree");
    if (a.logw && a.sdf_dparam) return fail(SK_EUNSUPPORTED, "joint march: logw with a parameter-derivative integrand");
    // one boundary factor jB0 per march (DevState): a second logw integrand would be given the first one's
    if (a.logw && n_logw++) return fail(SK_EUNSUPPORTED, "joint march: at most one logw integrand");
    ntr += a.logw ? 2 : 1;
    if (ntr > kJointTr) return fail(SK_EINVAL, "joint march: too many raw transforms");
  }
  return SK_OK;
}

// The joint march behind both entries.  Lags: a host array `xs` (staged upload) or, if xs == nullptr, `d_lags_in`
// already on the device (any order).  On return *d_vals / *d_errs point at nt x n device arrays in INPUT order
// (row t = integrand t), valid until the next joint call on cfgs[0].
int joint_core(sk_config* const* cfgs, size_t ncfg, const double* xs, const double* d_lags_in, size_t n, double k0,
               const int* param_derivative, double** d_vals, double** d_errs, size_t extra_bytes, void** extra) {
  if (!cfgs || ncfg == 0 || !cfgs[0]) return fail(SK_EINVAL, "null config");
  if (ncfg > (size_t)kJointMax) return fail(SK_EINVAL, "at most 8 integrands per joint march");
  sk_config* c0 = cfgs[0];
  if (c0->args.derivative || c0->args.logw)
    return fail(SK_EINVAL, "joint march: the first config must be the plain kernel (it provides the shared nodes)");
  // ---- the integrands must share everything but the integrand itself
  JointSpec js;
  std::memset(&js, 0, sizeof(js));
  js.nt = (int)ncfg;
  int n_logw = 0;
  for (size_t t = 0; t < ncfg; ++t) {
    const sk_config* c = cfgs[t];
    if (!c) return fail(SK_EINVAL, "null config");
    const sk_config_args& a = c->args;
    if (c->device != c0->device) return fail(SK_EINVAL, "joint march: configs must live on one device");
    if (a.dim != c0->args.dim) return fail(SK_EINVAL, "joint march: the configs must have one dim");
    // catalogue SDFs are evaluated on the device; host callbacks once per panel for all integrands -- not mixed
    if ((a.sdf_kind == SK_SDF_CALLBACK) != (c0->args.sdf_kind == SK_SDF_CALLBACK))
      return fail(SK_EUNSUPPORTED, "joint march: all integrands catalogue SDFs or all host callbacks");
    if (a.sdf_kind == SK_SDF_CALLBACK && a.logw)
      return fail(SK_EUNSUPPORTED, "joint march: logw with a callback SDF (use sk_kernel_values_multi)");
    if (c->m != c0->m || c->k != c0->k || a.tol != c0->args.tol || a.alpha != c0->args.alpha || a.criteria != c0->args.criteria ||
        a.has_tail != c0->args.has_tail || (a.has_tail && a.tail != c0->args.tail) || c->dc.w != c0->dc.w)
      return fail(SK_EINVAL, "joint march: quadspec, tol, alpha, convergence_criteria, tail and NUFFT width must agree");
    if (a.logw && a.sdf_dparam) return fail(SK_EUNSUPPORTED, "joint march: logw with a parameter-derivative integrand");
    // one boundary factor jB0 per march (DevState): a second logw integrand would be given the first one's
    if (a.logw && n_logw++) return fail(SK_EUNSUPPORTED, "joint march: at most one logw integrand");
    const int ntr_needed = a.logw ? 2 : 1;
    if (js.ntr + ntr_needed > kJointTr) return fail(SK_EINVAL, "joint march: too many raw transforms");
    js.c[t] = c->c;
    js.tail_tr[t] = js.ntr;
    for (int q = 0; q < ntr_needed; ++q) {
      JointTr& tr = js.tr[js.ntr++];
      tr.sdf_kind = a.sdf_kind;
      std::memcpy(tr.prm, c->dc.sdf_prm, sizeof(tr.prm));
      tr.xw = a.derivative ? 1 : 0;                      // p = p_base + 1 (src/adaptive.jl:42)
      tr.logmode = a.logw ? (q == 0 ? 1 : 2) : 0;
      tr.imag = a.logw ? q : (a.derivative ? 1 : 0);     // :sin for K' and for slot B (src/quadrature.jl:177,198)
      tr.out = (int)t;
      tr.base = js.ntr - 1;
      for (int e = 0; e < js.ntr - 1; ++e)
        if (js.tr[e].sdf_kind == tr.sdf_kind && std::memcmp(js.tr[e].prm, tr.prm, 7 * sizeof(double)) == 0) {
          tr.base = e;
          break;
        }
    }
  }
  if (n == 0) return SK_OK;
  if (n > (size_t)0x7fffffff) return fail(SK_EINVAL, "at most 2^31-1 lags per call");
  HIPCHK(hipSetDevice(c0->device));
  hipStream_t st = c0->own_stream;
  const int nt = js.nt, ntr = js.ntr;
  const size_t extra_b = a256(extra_bytes);
  const int64_t n1 = c0->dc.n1;
  // ---- values at x == 0 (src/adaptive.jl:134-145)
  JointBufs jb;
  std::memset(&jb, 0, sizeof(jb));
  for (int t = 0; t < nt; ++t) {
    double v = k0;
    if (cfgs[t]->args.derivative) v = 0.0;
    else if (param_derivative && param_derivative[t]) {
      int rc = sk_compute_k0(cfgs[t], &v);
      if (rc) return rc;
    }
    jb.ks0[t] = v;
  }
  // ---- device buffers (grow-only, owned by the first config)
  const size_t a8 = a256(n * 8);
  const size_t tb = sort_unique_temp_bytes((long long)n);
  const size_t fft_b = a256((size_t)ntr * 4 * n1 * sizeof(double2));
  // Hankel route: (strength, node) pairs of one transform, raw panel sums of all transforms (2 rules x n each)
  const bool hk_route = joint_hankel_wanted(c0, n);
  const size_t hksrc_b = hk_route ? a256((size_t)3 * c0->M * sizeof(double2)) : 0;
  const size_t hkraw_b = hk_route ? (size_t)(2 * ntr) * a8 : 0;
  // exactly what the pointer bumps below consume
  const size_t need = extra_b + 3 * a8 + 256 + (a256(tb) + 256) + (size_t)(8 * nt) * a8 + (size_t)(2 * nt) * a8 + 2 * fft_b +
                      hksrc_b + hkraw_b;
  int rc = ensure_dev_buf(&c0->d_joint, &c0->joint_bytes, need);
  if (rc) return rc;
  char* p = (char*)c0->d_joint;
  if (extra) *extra = p;
  p += extra_b;
  double* d_xs = (double*)p; p += a8;
  double* uxs = (double*)p; p += a8;
  long long* inv = (long long*)p; p += a8;
  long long* d_nu = (long long*)p; p += 256;
  void* tmp = p; p += a256(tb) + 256;
  double* big = (double*)p; p += (size_t)(8 * nt) * a8;
  double* gath = (double*)p; p += (size_t)(2 * nt) * a8;  // results in input order: nt x n values, then nt x n errors
  jb.fft = (double2*)p; p += fft_b;
  jb.fft_out = (double2*)p; p += fft_b;
  double2* hk_src = (double2*)p; p += hksrc_b;
  double* hk_raw = (double*)p;
  // ---- lags: staged upload, one sort + dedup for all integrands (src/adaptive.jl:99-120)
  int order = SK_ORDER_UNSORTED;  // device lags (pairwise distances): sorted and deduplicated below
  if (xs) {
    rc = h2d_doubles(c0, d_xs, xs, n, st, &order);
    if (rc) return rc;
  } else {
    d_xs = const_cast<double*>(d_lags_in);
    launch_check_sorted(d_xs, (long long)n, c0->db.flags, st);  // invalid pairs surface as NaN lags (k_pair_lags)
  }
  if (order == SK_ORDER_INVALID) return fail(SK_EINVAL, "lags must be non-negative and finite");
  if (order == SK_ORDER_UNKNOWN) order = SK_ORDER_UNSORTED;  // pinned host input of unknown order: sorting is harmless
  long long nu = (long long)n;
  const bool sorted = (order == SK_ORDER_SORTED);
  const double* lags = d_xs;
  if (!sorted) {
    if (sort_unique_sync(d_xs, (long long)n, uxs, inv, d_nu, c0->h_nuniq, tmp, tb, st, &nu, nullptr, nullptr, true))
      return fail(SK_EHIP, "sort failed");
    lags = uxs;
  }
  const size_t sn = (size_t)nu;
  double* q = big;
  jb.accK[0] = q; q += (size_t)nt * sn; jb.accK[1] = q; q += (size_t)nt * sn;
  jb.accE[0] = q; q += (size_t)nt * sn; jb.accE[1] = q; q += (size_t)nt * sn;
  jb.P[0] = q; q += (size_t)nt * sn; jb.P[1] = q; q += (size_t)nt * sn;
  jb.ks = q; q += (size_t)nt * sn; jb.errs = q; q += (size_t)nt * sn;
  // ---- rocFFT plan for this number of transforms
  if (!c0->plan_joint[ntr]) {
    size_t len[1] = {(size_t)n1};
    FFTCHK(rocfft_plan_create(&c0->plan_joint[ntr], rocfft_placement_notinplace, rocfft_transform_type_complex_inverse,
                              rocfft_precision_double, 1, len, (size_t)4 * ntr, nullptr));
    size_t wb = 0;
    rocfft_plan_get_work_buffer_size(c0->plan_joint[ntr], &wb);
    FFTCHK(rocfft_execution_info_create(&c0->info_joint[ntr]));
    if (wb) {
      HIPCHK(hipMalloc(&c0->fft_work_joint[ntr], wb));
      FFTCHK(rocfft_execution_info_set_work_buffer(c0->info_joint[ntr], c0->fft_work_joint[ntr], wb));
    }
  }
  FFTCHK(rocfft_execution_info_set_stream(c0->info_joint[ntr], st));
  // ---- the march (device-resident; the host sees the state once per round of R steps, as run_core)
  DevConst dcj = c0->dc;
  dcj.joint_nt = nt;
  DevBuffers db = c0->db;
  db.xs = lags;
  db.ks = jb.ks;
  db.errs = jb.errs;
  db.check_input = 0;
  std::memset(&c0->stats, 0, sizeof(c0->stats));
  c0->stats.n_unique = nu;
  const DevState* hs = &c0->h_status->state;
  // dim >= 2: direct Bessel sums with shared J_nu evaluations (sk_joint_kernels.inc, "dim >= 2"); callback SDFs: the
  // host evaluates every integrand at the panel's nodes, so it needs the state before each step (one step per round)
  const bool bessel = c0->args.dim > 1;
  const bool cb = c0->args.sdf_kind == SK_SDF_CALLBACK;
  const size_t M3 = (size_t)3 * c0->M;
  const double* d_fv = nullptr;
  if (cb) {
    if (c0->jfv_count < (size_t)ntr * M3) {
      if (c0->d_jfv) HIPCHK(hipFree(c0->d_jfv));
      if (c0->h_jfv) HIPCHK(hipHostFree(c0->h_jfv));
      c0->d_jfv = nullptr;
      c0->h_jfv = nullptr;
      c0->jfv_count = 0;
      HIPCHK(hipMalloc((void**)&c0->d_jfv, sizeof(double) * (size_t)kJointTr * M3));
      HIPCHK(hipHostMalloc((void**)&c0->h_jfv, sizeof(double) * (size_t)kJointTr * M3, hipHostMallocDefault));
      c0->jfv_count = (size_t)kJointTr * M3;
    }
    d_fv = c0->d_jfv;
  }
  int t = 0, mode = SK_MODE_INIT, R = cb ? 1 : 2;
  // dim 2, lag sets of hankel_min_lags and more (the unique count decides, as in run_core): every raw transform through
  // the Hankel kernels of the single-integrand march in their raw mode, then the joint epilogue (sk_joint_kernels.inc)
  const int hk_min = c0->args.hankel_min_lags == 0 ? kHkDefaultMinLags : c0->args.hankel_min_lags;
  const bool hankel = hk_route && nu >= hk_min;
  if (hk_route && !hankel) return kJointHkFallback;  // duplicates shrank the set below the threshold: the caller decides
  if (hankel) {
    const int nu0 = c0->args.dim / 2 - 1;
    for (;;) {
      const unsigned int seq = ++c0->seq;
      launch_advance(dcj, db, mode, t, nu, k0, jb.ks0[0], seq, st);
      // the node generation needs the device's state only: it runs while the status block travels
      launch_gen(dcj, db, SK_MODE_ASIS, t, nu, k0, jb.ks0[0], SK_G_WEIGHT, st);
      rc = joint_wait_status(c0, seq, st);
      if (rc) return rc;
      if (!xs && (c0->h_status->flags[1] & 1)) {  // device lags: the validity check ran on the unsorted input
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemsetAsync(c0->db.flags, 0, 2 * sizeof(int), st));
        return fail(SK_EINVAL, "lags must be non-negative and finite (index pair out of range?)");
      }
      if (hs->done) break;
      launch_joint_strengths(dcj, db, js, jb, t & 1, nullptr, st);  // + the integrands' tail fits and the logw boundary factor
      // one plan per Bessel order in use: J_nu0 for the plain kernel and the parameter derivatives, J_nu0+1 for K' and
      // slot B of a logw integrand (src/quadrature.jl:176-180,198)
      HkInner hin[2];
      bool have[2] = {false, false};
      for (int r = 0; r < ntr; ++r) {
        const int o = js.tr[r].imag ? 1 : 0;
        if (!have[o]) {
          bool boxes = false;
          rc = hk_plan_boxes(c0, db, hs, nu0 + o, o ? c0->hk_a2 : c0->hk_a, SK_RAW_A, st, hin[o], boxes);
          if (rc) return rc;
          if (!boxes) {
            if (t == 0) return kJointHkFallback;  // nothing evaluated yet: one march per integrand instead
            return fail(SK_EUNSUPPORTED, "joint march (dim 2): a step does not fit the Hankel buffers");
          }
          have[o] = true;
        }
      }
      // (the two plans differ in the series only -- same r_c, boxes and buffers -- and the transforms run one after the
      // other on the march's stream, each with its boxes on the side streams)
      for (int r = 0; r < ntr; ++r) {
        const int o = js.tr[r].imag ? 1 : 0;
        launch_joint_pack(dcj, db, jb, r, hk_src, st);
        DevBuffers dbt = db;
        dbt.src = hk_src;
        dbt.lw_raw = hk_raw + (size_t)(2 * r) * sn;
        rc = hk_launch_boxes(c0, dbt, hin[o], t, nu, SK_RAW_A, st);
        if (rc) return rc;
      }
      launch_joint_hk_epilogue(dcj, db, js, jb, hk_raw, t & 1, nu, st);
      ++t;
      mode = SK_MODE_TRANSITION;
      c0->stats.n_steps_launched++;
    }
  } else
  for (;;) {
    if (cb) {
      const unsigned int seq = ++c0->seq;
      launch_advance(dcj, db, mode, t, nu, k0, jb.ks0[0], seq, st);
      rc = joint_wait_status(c0, seq, st);
      if (rc) return rc;
      if (hs->done) break;
      // every integrand at the nodes of this (sub)panel, and |f_t(b)| for the tail fits (src/adaptive.jl:204-220)
      joint_host_nodes(c0, hs->pp, c0->h_nodes);
      for (int r = 0; r < ntr; ++r) cfgs[js.tr[r].out]->sdf.batch(c0->h_nodes.data(), c0->h_jfv + (size_t)r * M3, M3);
      HIPCHK(hipMemcpyAsync(c0->d_jfv, c0->h_jfv, sizeof(double) * (size_t)ntr * M3, hipMemcpyHostToDevice, st));
      if (hs->tail_pending) {
        for (int tt = 0; tt < nt; ++tt) {
          c0->h_scal[16 + tt] = std::fabs(cfgs[tt]->sdf(hs->b));
          HIPCHK(hipMemcpyAsync(&db.state[t & 1].jtail[tt].t1, &c0->h_scal[16 + tt], sizeof(double), hipMemcpyHostToDevice, st));
        }
      }
      mode = SK_MODE_ASIS;
    }
    for (int r = 0; r < R; ++r) {
      launch_gen(dcj, db, mode, t, nu, k0, jb.ks0[0], SK_G_WEIGHT, st);
      if (bessel) {
        launch_joint_strengths(dcj, db, js, jb, t & 1, d_fv, st);
        launch_bessel_joint(dcj, db, js, jb, t & 1, nu, st);
      } else {
        launch_joint_spread(dcj, c0->d_ec, db, js, jb, t & 1, d_fv, st);
        {
          void* in[1] = {jb.fft};
          void* out[1] = {jb.fft_out};
          FFTCHK(rocfft_execute(c0->plan_joint[ntr], in, out, c0->info_joint[ntr]));
        }
        launch_joint_interp(dcj, c0->d_ec, db, js, jb, t & 1, nu, st);
      }
      ++t;
      mode = SK_MODE_TRANSITION;
      c0->stats.n_steps_launched++;
    }
    if (cb) continue;
    const unsigned int seq = ++c0->seq;
    launch_advance(dcj, db, mode, t, nu, k0, jb.ks0[0], seq, st);
    rc = joint_wait_status(c0, seq, st);
    if (rc) return rc;
    if (!xs && (c0->h_status->flags[1] & 1)) {  // device lags: the validity check ran on the unsorted input
      HIPCHK(hipStreamSynchronize(st));
      HIPCHK(hipMemsetAsync(c0->db.flags, 0, 2 * sizeof(int), st));
      return fail(SK_EINVAL, "lags must be non-negative and finite (index pair out of range?)");
    }
    if (hs->done) break;
    mode = SK_MODE_ASIS;
    R = R < 16 ? 2 * R : 16;
  }
  HIPCHK(hipGetLastError());
#ifdef SK_STAMPS
  if (const char* f = getenv("SK_STAMP_FILE")) {  // tools/diag/stamps_joint.py
    std::vector<unsigned long long> h(8 * 8192);
    (void)hipMemcpy(h.data(), db.stamps, h.size() * 8, hipMemcpyDeviceToHost);
    if (FILE* fp = fopen((std::string(f) + ".joint").c_str(), "wb")) { fwrite(h.data(), 8, h.size(), fp); fclose(fp); }
  }
#endif
  c0->stats.n_outer = hs->n_outer;
  c0->stats.n_panels = hs->n_panels;
  c0->stats.n_accepted = hs->n_accepted;
  c0->stats.sum_active = hs->sum_active;
  c0->stats.last_b = hs->b;
  c0->trace_len = (size_t)hs->trace_len;
  c0->trace_valid = false;
  if (hs->done < 0) {
    const int code = -hs->done;
    return fail(code, code == SK_ESUBDIVIDE ? "The sub-interval (a, b) has been split too many times (b - a < 1e-16). Exiting to avoid infinite splitting."
                      : (code == SK_ENAN ? "NaN detected in panel integral..." : "device-side failure in the joint march"));
  }
  // ---- values at x == 0 of the other integrands (integrand 0's were written by the march's first step)
  if (hs->ix1 == 1) {
    for (int tt = 1; tt < nt; ++tt) {
      c0->h_scal[2 * tt] = jb.ks0[tt];
      c0->h_scal[2 * tt + 1] = std::nan("");
      HIPCHK(hipMemcpyAsync(jb.ks + (size_t)tt * sn, &c0->h_scal[2 * tt], sizeof(double), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(jb.errs + (size_t)tt * sn, &c0->h_scal[2 * tt + 1], sizeof(double), hipMemcpyHostToDevice, st));
    }
  }
  // ---- results in input order on the device (sorted input: they already are)
  if (!xs && c0->h_status->flags[0]) HIPCHK(hipMemsetAsync(c0->db.flags, 0, 2 * sizeof(int), st));  // "unsorted" is expected
  if (sorted) {
    *d_vals = jb.ks;
    *d_errs = jb.errs;
  } else {
    for (int tt = 0; tt < nt; ++tt)
      launch_gather(jb.ks + (size_t)tt * sn, jb.errs + (size_t)tt * sn, inv, (long long)n, gath + (size_t)tt * n,
                    gath + (size_t)(nt + tt) * n, st);
    *d_vals = gath;
    *d_errs = gath + (size_t)nt * n;
  }
  return SK_OK;
}

}  // namespace

extern "C" int sk_kernel_values_joint(sk_config* const* cfgs, size_t ncfg, const double* xs, size_t n, double k0,
                                      const int* param_derivative, double* vals_out, double* errs_out) {
  if (!xs && n) return fail(SK_EINVAL, "null lags");
  if (cfgs && ncfg && cfgs[0] && cfgs[0]->args.dim > 1) {
    // dim >= 2: the joint march evaluates direct Bessel sums (O(lags x nodes) per panel, the reference's own branch
    // for its 136-lag Jacobian tests); big lag sets go through the fast Hankel transform, one march per integrand
    const int hk_min = cfgs[0]->args.hankel_min_lags == 0 ? kHkDefaultMinLags : cfgs[0]->args.hankel_min_lags;
    if (hk_min > 0 && n >= (size_t)hk_min) {
      const int rcc = joint_check_args(cfgs, ncfg);
      if (rcc) return rcc;
      // dim 2: ONE march with every transform through the Hankel kernels (joint_core); otherwise, or when a step does
      // not fit the Hankel buffers, one fast-Hankel march per integrand, side by side
      if (!joint_hankel_wanted(cfgs[0], n)) {
        std::vector<double> k0s(ncfg, k0);
        return sk_kernel_values_multi(cfgs, ncfg, xs, n, k0s.data(), param_derivative, vals_out, errs_out);
      }
    }
  }
  static const bool timing = getenv("SK_JOINT_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  double *dv = nullptr, *de = nullptr;
  int rc = joint_core(cfgs, ncfg, xs, nullptr, n, k0, param_derivative, &dv, &de, 0, nullptr);
  if (rc == kJointHkFallback) {
    std::vector<double> k0s(ncfg, k0);
    return sk_kernel_values_multi(cfgs, ncfg, xs, n, k0s.data(), param_derivative, vals_out, errs_out);
  }
  if (rc || n == 0) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  sk_config* c0 = cfgs[0];
  for (size_t t = 0; t < ncfg; ++t) {
    rc = d2h_doubles(c0, vals_out + t * n, dv + t * n, n, c0->own_stream);
    if (!rc) rc = d2h_doubles(c0, errs_out + t * n, de + t * n, n, c0->own_stream);
    if (rc) return rc;
  }
  if (timing)
    fprintf(stderr, "[sk joint] upload + march %.3f ms, download %.3f ms\n",
            std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  return SK_OK;
}

// gen_kernel + gen_kernel_jacobian for point pairs (src/model.jl:53-77, src/derivatives.jl:86-112): see the header.
extern "C" int sk_kernel_jacobian_pairs(sk_config* const* cfgs, size_t ncfg, const int* role, const int* col, const double* pts,
                                        size_t npts, int pdim, const int64_t* pairs, size_t npairs, double k0,
                                        const double* dlag, int nwarp, const int* warp_cols, int ncols, double* vals_out,
                                        double* jac_out) {
  if (!cfgs || ncfg == 0 || !cfgs[0] || !role || !col || !pts || !pairs) return fail(SK_EINVAL, "null argument");
  if (pdim < 1 || npts < 1 || ncols < 0 || nwarp < 0 || (nwarp && (!dlag || !warp_cols))) return fail(SK_EINVAL, "bad argument");
  if (npairs == 0) return SK_OK;
  sk_config* c0 = cfgs[0];
  HIPCHK(hipSetDevice(c0->device));
  hipStream_t st = c0->own_stream;
  std::vector<int> pd(ncfg, 0);
  int t_warp = -1;
  for (size_t t = 0; t < ncfg; ++t) {
    if (role[t] == 1) {
      if (col[t] < 0 || col[t] >= ncols) return fail(SK_EINVAL, "Jacobian column out of range");
      pd[t] = 1;  // kernel_sdf_derivatives / kernel_singularity_derivative pass param_derivative=true (:68,80)
    } else if (role[t] == 2) {
      t_warp = (int)t;
    } else if (role[t] != 0) {
      return fail(SK_EINVAL, "role must be 0 (K), 1 (a Jacobian column) or 2 (K' for the warping columns)");
    }
  }
  if (nwarp && t_warp < 0) return fail(SK_EINVAL, "warping columns need a config with role 2 (derivative=true)");
  for (int q = 0; q < nwarp; ++q)
    if (warp_cols[q] < 0 || warp_cols[q] >= ncols) return fail(SK_EINVAL, "Jacobian column out of range");
  // scratch ahead of the march's buffers: points, pairs, lags, d lag / d warp, the Jacobian, column maps
  const size_t b_pts = a256(npts * pdim * 8), b_pairs = a256(2 * npairs * 8), b_lags = a256(npairs * 8),
               b_dlag = a256((size_t)npairs * (nwarp > 0 ? nwarp : 1) * 8), b_jac = a256((size_t)npairs * (ncols > 0 ? ncols : 1) * 8),
               b_map = a256((ncfg * 2 + (size_t)nwarp + 8) * sizeof(int));
  // the lags must exist before the march: a first (cheap) pass sizes the buffers with n = npairs
  void* extra = nullptr;
  {
    // allocate by running the size computation of joint_core on an empty call is not possible; reserve directly
    const size_t a8 = a256(npairs * 8);
    const size_t tb = sort_unique_temp_bytes((long long)npairs);
    JointSpec dummy;  // only ntr matters for sizing: at most two transforms per integrand
    (void)dummy;
    const size_t fft_b = a256((size_t)(2 * ncfg) * 4 * c0->dc.n1 * sizeof(double2));
    const size_t hk_b = joint_hankel_wanted(c0, npairs) ? a256((size_t)3 * c0->M * sizeof(double2)) + (size_t)(4 * ncfg) * a8 : 0;
    const size_t need = 3 * a8 + tb + 512 + (size_t)(10 * ncfg) * a8 + 2 * fft_b + hk_b + b_pts + b_pairs + b_lags + b_dlag + b_jac + b_map + 4096;
    int rc0 = ensure_dev_buf(&c0->d_joint, &c0->joint_bytes, need);
    if (rc0) return rc0;
  }
  const size_t extra_bytes = b_pts + b_pairs + b_lags + b_dlag + b_jac + b_map;
  char* e = (char*)c0->d_joint;  // joint_core places `extra` at the head of the (already large enough) buffer
  double* d_pts = (double*)e; e += b_pts;
  int64_t* d_pairs = (int64_t*)e; e += b_pairs;
  double* d_lags = (double*)e; e += b_lags;
  double* d_dlag = (double*)e; e += b_dlag;
  double* d_jac = (double*)e; e += b_jac;
  int* d_map = (int*)e;
  int rc = h2d_doubles(c0, d_pts, pts, npts * pdim, st, nullptr);
  if (!rc) rc = hipStreamSynchronize(st) == hipSuccess ? SK_OK : fail(SK_EHIP, "stream synchronisation failed");
  if (!rc) rc = h2d_doubles(c0, (double*)d_pairs, (const double*)pairs, 2 * npairs, st, nullptr);
  if (!rc && nwarp) {
    if (hipStreamSynchronize(st) != hipSuccess) rc = fail(SK_EHIP, "stream synchronisation failed");
    if (!rc) rc = h2d_doubles(c0, d_dlag, dlag, npairs * (size_t)nwarp, st, nullptr);
  }
  if (rc) return rc;
  std::vector<int> hmap;
  for (size_t t = 0; t < ncfg; ++t) { hmap.push_back(role[t]); hmap.push_back(col[t]); }
  for (int q = 0; q < nwarp; ++q) hmap.push_back(warp_cols[q]);
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpy(d_map, hmap.data(), hmap.size() * sizeof(int), hipMemcpyHostToDevice));
  launch_pair_lags(d_pts, (long long)npts, pdim, (const long long*)d_pairs, (long long)npairs, d_lags, st);
  double *dv = nullptr, *de = nullptr;
  const int hk_min = c0->args.hankel_min_lags == 0 ? kHkDefaultMinLags : c0->args.hankel_min_lags;
  const bool big = c0->args.dim > 1 && hk_min > 0 && npairs >= (size_t)hk_min;
  bool separate = big && !joint_hankel_wanted(c0, npairs);
  if (big) {
    rc = joint_check_args(cfgs, ncfg);
    if (rc) return rc;
  }
  if (!separate) {
    // the joint march: dim 1 (type-3 NUFFT), small lag sets in dim >= 2 (direct Bessel sums), dim 2 above the threshold
    // (every transform through the Hankel kernels)
    rc = joint_core(cfgs, ncfg, nullptr, d_lags, npairs, k0, pd.data(), &dv, &de, extra_bytes, &extra);
    if (rc == kJointHkFallback) separate = true;
    else if (rc) return rc;
    else if (extra != (void*)c0->d_joint) return fail(SK_EHIP, "internal error: joint buffers moved");
  }
  if (separate) {
    // one fast-Hankel march per integrand, side by side
    rc = multi_device_core(cfgs, ncfg, d_lags, npairs, k0, pd.data(), &dv, &de);
    if (rc) return rc;
  }
  if (ncols > 0) {
    launch_jac_assemble(dv, (long long)npairs, (int)ncfg, d_map, d_dlag, nwarp, t_warp, ncols, d_jac, st);
    rc = d2h_doubles(c0, jac_out, d_jac, npairs * (size_t)ncols, st);
    if (rc) return rc;
  }
  for (size_t t = 0; t < ncfg; ++t)
    if (role[t] == 0 && vals_out) return d2h_doubles(c0, vals_out, dv + t * npairs, npairs, st);
  return SK_OK;
}

