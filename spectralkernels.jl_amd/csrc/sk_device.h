// sk_device.h -- device-resident state of the adaptive panel march and kernel launch API.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sk {

constexpr int kMaxWidth = 16;
constexpr int kMaxNco = kMaxWidth / 2 + 1;
// Coefficients per half (even / odd part) of the ES kernel's piecewise polynomials: degree 2 nco - 1.  The approximation
// error bottoms out at the kernel's own cutoff jump e^-beta once the degree reaches w - 1 (w = 12: 1.6e-12 at degree 11
// and at 13; 5e-11 at 9) -- one coefficient per half less than rounds 1-3 carried, 13 FP64 instructions per lag and per
// staged source less; the narrow kernels (w <= 8, tol >= 1e-4) keep degree w + 1, where the last coefficient still buys 2.5x.
#ifdef SK_OLD_NCO  // diagnostic builds: rounds 1-3's degree w + 1 everywhere (same-box A/B)
__host__ __device__ constexpr int es_nco(int w) { return w / 2 + 1; }
#else
__host__ __device__ constexpr int es_nco(int w) { return w >= 10 ? w / 2 : w / 2 + 1; }
#endif
constexpr int kMaxCheb = 24;
constexpr int kStackCap = 5000;  // src/quadrature.jl:7
constexpr int kTraceCap = 4096;
// nonuniform fast Hankel transform (dim >= 2): level-split asymptotic expansion
constexpr int kHkTerms = 16;     // most terms of the Hankel asymptotic series (valid for z >= kHkZ0)
// Terms actually used with kernel width w: the NUFFT error target is ~10^-(w-1), and the k-th term of the series at
// z0 = 24 is 5e-3, 1e-4, 5e-6, 3e-7, 3e-8, 3e-9, 4e-10, 6e-11, 9e-12, 2e-12, 4e-13, 8e-14, 2e-14, 6e-15, 2e-15, 5e-16.
__host__ __device__ constexpr int hk_terms(int w) { return w <= 6 ? 4 : w <= 8 ? 6 : w <= 10 ? 8 : w <= 12 ? 10 : w <= 14 ? 13 : 16; }
constexpr int kHkNodes = 160;    // Chebyshev nodes of the near-field interpolant per lag level
constexpr int kHkSplit = 16;     // workgroups sharing one near-field node sum
constexpr int kHkMaxLevels = 12; // lag levels (ratio kHkBeta) incl. the innermost near-only one
constexpr double kHkZ0 = 24.0;
// dim >= 2: lag sets of at least this many lags take the fast Hankel transform (hankel_min_lags = 0).  Round 3: 8 (was 2048):
// with the transform by boxes the direct Bessel sums lose at every size -- 46 ms against 0.38 ms per call at 100 lags on the
// default rule (tools/diag/hankel_cutoff.py); they remain for the smallest sets and as the reference's own slow branch.
constexpr int kHkDefaultMinLags = 8;
constexpr double kHkBeta = 8.0;

// One step of the Hankel path: lag levels l = 0..L (level L is near-only).  Level l < L covers lags in
// (edge[l+1], edge[l]], treats sources w >= om[l] by the asymptotic series and w < om[l] by the
// Chebyshev interpolant on that lag interval.
struct HankelPlan {
  int L;                       // number of far levels
  int nu;                      // Bessel order
  double edge[kHkMaxLevels + 2];  // edge[0] = r_hi (inclusive top), edge[L+1] = 0
  double om[kHkMaxLevels + 1];    // far/near source threshold per level (om[L] = +inf)
  double a[kHkTerms];          // asymptotic coefficients a_k(nu)
  // innermost level with at most kHkNodes lags: its (near-only) sums are taken directly at those lags
  // [inner_lo, inner_lo + inner_cnt) instead of at Chebyshev nodes
  int direct_inner;
  long long inner_lo, inner_cnt;
  double pa, pb;               // the (sub)panel, for bounding the near prefix by whole sub-panels
  int near_levels;             // levels k_hk_near / k_hk_cheb cover: L + 1, or L when the inner region (HkInner) takes the rest
};

// The Hankel path by boxes (sk_hankel_inner.inc): per panel one asymptotic-series NUFFT for the lags (r_c, r_hi] and
// the sources w >= om0 (z >= kHkZ0 there), and two smooth boxes of the kernel J_nu(2 pi w r):
//   near box   sources w < om0 (a prefix of the sorted sources; none once the panel starts beyond om0) at the lags
//              (r_c, r_hi]: Qn lag intervals of length elln, direct sums at kHkNodes Chebyshev nodes each;
//   inner box  all sources at the lags (0, r_c]: Chebyshev moments of the sources in `nb` bands of `gs` sub-panels
//              (`ngm` per band), Q lag intervals of length ell.
constexpr int kHkNgCap = 1024;   // most moments per band (LDS of k_hk_eqw)
constexpr int kHkQCap = 2048;    // most lag intervals of both boxes together
struct HkSeries { double a[kHkTerms]; double cs, sn; };  // asymptotic coefficients a_k(nu); cos, sin of (2 nu + 1) pi / 4
struct HkInner {
  int nu, nb, gs, Q, ngm;
  int Qn;                      // near box intervals (0: no near sources, or no far field at all)
  int has_far;                 // 0: the whole panel is one smooth box (r_c = r_hi, no NUFFT)
  int nt;                      // series terms of this step (<= hk_terms(w))
  double escale;               // exponent e of the r^-e factor folded into the node values (dim/2 - 1; 0 in the raw passes)
  double r_c, ell, pa, pb;
  double om0, elln, r_hi;
  long long lag_lo, lag_cnt;   // the inner lags: indices [lag_lo, lag_lo + lag_cnt) of the sorted lag array
  HkSeries ser;                // Hankel's expansion of J_nu: the series NUFFT and the large-argument kernel values
  // zterm[K], 2 <= K < nt: smallest z = 2 pi om0 r from which K series terms suffice (first omitted term no larger than the
  // full count's at z0).  The lags are sorted: a workgroup of the far kernel whose smallest lag lies beyond it sums K terms
  // only (round 4: most lags of a panel sit far above r_c, where half the terms are below the truncation level anyway)
  double zterm[kHkTerms + 1];
};

// Joint march of several integrands on one lag set (sk_kernel_values_joint; src/derivatives.jl:51-112 calls
// kernel_values once per integrand on the same warp_lags).  Integrand t has its own running sums, tail fit and
// outputs; a raw transform is one set of source strengths through the shared spread / FFT / interpolation.
constexpr int kJointMax = 8;   // integrands per march
constexpr int kJointTr = 10;   // raw transforms per (sub)panel (a logw integrand takes two)
struct JointTr {
  int sdf_kind;    // catalogue SDF (prm[7]: parameter-derivative index, prm[4]: half-integer fast path)
  int xw;          // extra factor w (the K' integrand: p = p_base + 1, src/adaptive.jl:42)
  int logmode;     // 0: none; 1 / 2: slots A / B of a logw integrand (src/quadrature.jl:186-228, :243)
  int imag;        // take the imaginary part (sin kernel) instead of the real part (cos kernel)
  int out;         // integrand this transform feeds
  int base;        // first transform with the same SDF and parameters prm[0..6] (<= own index): S(w) is evaluated once
  double prm[8];
};
struct JointSpec {
  int nt, ntr;
  double c[kJointMax];        // scale of integrand t (src/adaptive.jl:43-45)
  int tail_tr[kJointMax];     // transform whose plain SDF |f(b)| drives integrand t's tail fit (src/adaptive.jl:204-220)
  JointTr tr[kJointTr];
};
// per-call buffers of a joint march, integrand-major with stride n (lags)
struct JointBufs {
  double* accK[2]; double* accE[2]; double* P[2];  // nt x n each
  double* ks; double* errs;                        // nt x n: outputs (sorted-unique order)
  double2* fft; double2* fft_out;                  // ntr x 4 x n1 spread grids / their transforms
  double ks0[kJointMax];                           // values stored at x == 0
};

// Per-config constants, passed to kernels by value.
struct DevConst {
  // quadrature
  int64_t m, k, M;     // quadspec and M = m*k
  int dim, derivative, crit0, has_tail, logw;
  double alpha, p, c, tol, tail;
  int sdf_kind;        // 0 = values supplied by host callback
  int joint_nt;        // > 0: joint march with this many integrands (tail fits per integrand)
  double sdf_prm[8];
  // NUFFT geometry (fixed per config): spread grid n1, FFT length n2, width w
  int64_t n1, n2;
  int w, nco, ncheb;
  double theta_max, delta;  // delta = 2/(n1-w): grid spacing in u = (omega-Cx)/X
  int log2m;           // log2(m) if m is a power of two, else -1
  double inv_k;        // 1/k if k is a power of two (exact), else 0
  double inv_theta_max;
  double tau_fac, theta_fac;  // n2/(n1-w), 2 pi/(n1-w): panel-independent factors of PanelParams::tau / theta_scale
  // tables (device pointers)
  const double* leg_no1; const double* leg_wt1; const double* leg_no2; const double* leg_wt2;
  const double* jac_no1; const double* jac_wt1; const double* jac_no2; const double* jac_wt2;
  const double* pos[2][2];      // [variant][rule] source positions in grid units, P_j in [w/2, n1-w/2]
  const int* cellstart[2][2];   // [variant][rule] first source with floor(P_j) >= cell, n1+1 entries
  const double* deconv;         // n1/2+1 values of 1/phi_hat(2 pi l/n2)
  const double2* otw;           // n1/2+1 values of e^{2 pi i l/n2}: the twiddle of the odd output samples (k_spread_b)
  // spread work list per variant: {rule << 30 | first cell, cells <= 64, first source, end source}
  // sorted by decreasing source count, so that the dense sub-panel-edge tiles start first
  const int4* tiles[2];
  int ntiles[2];
  int K;                        // batched march: most panels per step (1, 2, 4 or 8)
  int Kv;                       // K = 1 chosen by the automatic batch width: the full width, whose batching decisions the
                                // one-panel march re-derives (after its tiles) to notice an input that would batch; 0: no check
  int guess_mode;               // 0: runs extended on the certain criterion or the guess; 1: always extended
                                // (SK_BATCH_GUESS=always: tests drive the re-run and output-fix paths); 2: certain only
  int hybrid;                   // K = 1 launch inside a full-width march (the peel: one fresh outer panel per step): a state at
                                // which the full width might batch is skipped, not evaluated (BatchInfo::skip)
};

// ES-kernel tables (device memory; every workgroup stages them in LDS):
// piecewise-polynomial coefficients and the Chebyshev series of 1/phi_hat.
struct EsConst {
  double poly[(kMaxWidth / 2) * 2 * kMaxNco];
  double cheb[16];
};

struct PanelParams {
  double pa, pb;        // current (sub)panel
  double Cx, X;         // centre, half-width
  double r_hi;          // largest active lag
  double inv_rc;        // 2 / r_hi
  double kappa;         // X * r_hi/2: pre-phase turns per unit u
  double tau;           // fine-grid position of target: t = tau * v, v = 2 r / r_hi - 1
  double theta_scale;   // theta = theta_scale * v
  int jacobi;           // Gauss-Jacobi rule on the first sub-panel (p != 0 and pa == 0)
};

// Target-independent pieces of the convergence test of one outer panel
// (src/adaptive.jl:204-233): truncation bound = min(t1, t2num / x^((dim+1)/2)).
struct TailScalars {
  double t1, t2num;
  int crit;             // effective criteria for this panel (:panel after a NaN tail fit)
  int pad;
};

// Batched steps of the device-resident dim-1 march (catalogue SDFs): one launch triple evaluates up to K panels that
// the reference would evaluate one after the other PROVIDED each is accepted and none retires a lag -- then they all
// have the same width M/(2 r_hi) (src/adaptive.jl:152) or are the two children of a rejected panel
// (src/quadrature.jl:268-270), and share the spread geometry, ES weights, tap positions and 1/phi_hat; only the source
// strengths and the post-phase differ.  Slot j has centre Cx0 + j dCx and half-width pp.X.
//   kind 0  the current (sub)panel, plus (kb = 2) its left sibling, which sits on top of the bisection stack;
//   kind 1  a run of outer panels.  Slot j+1 follows slot j in the reference iff slot j is accepted and retires no
//           lag.  That is CERTAIN while the truncation bound of the largest active lag at pb[j] is >= tol |k0| / 2
//           (the lag fails check_convergence, src/adaptive.jl:231-233, so the scan from the top, :183-198, stops at
//           once); beyond that it is a guess, made while the previous outer panel retired nothing (spec_ok) and
//           checked afterwards from slot j's own convergence scan.
// The outcome is applied in order by the next step's prologue: the longest prefix v of slots that the reference would
// have evaluated.  The running sums are stored after slot 0 and after the last slot, so v == kb and v <= 1 are applied
// directly and any other v re-runs the batch capped at v slots.  Lags that retire at slot 0 of a longer batch take
// their outputs from slot 0's sums in the next launch (fix_*).
constexpr int kBatchMax = 8;
struct BatchInfo {
  int kb;              // panels evaluated by this step (1..K)
  int kind;
  int cmask;           // bit j: accepting slot j completes an outer panel (convergence scan with tail[j])
  int force_kb;        // > 0: cap of the next batch (re-run)
  // the entry this transition pushed on the bisection stack, if any: every workgroup derives the state redundantly but
  // only one writes the stack, so the others must not read that entry back from memory in the same launch
  int push_valid;
  int tails_pending;   // the tail fits of this batch's slots are left to the end of the spread launch (off the critical path)
  // sibling pair whose LEFT child starts at 0 with p != 0: slot 1's first sub-panel is Gauss-Jacobi (src/quadrature.jl:
  // 61-78) -- its own source positions and tile list (variant 1) in the spread launch, everything behind the FFT alike
  int mixed;
  int probe;           // one-panel march under the automatic batch width: the full width could batch at this step -- its
                       // decision is re-derived after the tiles (batch_probe_full), off the critical path
  // one-panel kernels inside a full-width march (DevConst::hybrid, sk_api.cpp): peel = this step is a single fresh outer
  // panel because the march is retiring lags -- the host launches the NEXT step with the lighter K = 1 instantiations;
  // skip = such a launch met a state at which the full width might batch: nothing is evaluated or published, the host
  // launches the step again at the full width (only ever set in the LDS copy of a skipping launch)
  int peel, skip;
  double push_a, push_b;
  double Cx0, dCx;     // centre of slot 0, centre step per slot (-2X: siblings, +2X: outer run)
  double pa[kBatchMax], pb[kBatchMax];  // the reference's bounds of every slot (trace, bisection)
  TailScalars tail[kBatchMax];          // of the outer panel that slot j completes (src/adaptive.jl:204-229)
  double g[kBatchMax + 1];              // outer run: |w^p S(w)| at pa[0], pb[0], pb[1], ...: size of a panel's
                                        // contribution at the largest lag (batch_build's guess)
};

struct DevState {
  PanelParams pp;
  double a, b;          // outer panel (src/adaptive.jl:152)
  double fb;            // S(b) for the tail fit (src/adaptive.jl:204-220)
  double k0abs;         // |k0|
  double ks0;           // value stored at x == 0
  long long n, ix1, hi, hi_prev;  // active lags are [ix1, hi)
  long long rhi_for;    // value of hi for which pp.r_hi / pp.inv_rc were loaded (-1: none)
  TailScalars tail;     // of the current outer panel
  TailScalars jtail[kJointMax];  // joint march: per integrand (jcrit: its effective criteria so far)
  int jcrit[kJointMax];
  double jB0;           // joint march: boundary factor of the logw integrand on the current [0, b] panel
  // confirmed running sums live in acc buffer `conf` (valid iff has_acc); every evaluation writes
  // the optimistic sums into the other buffer, an accepted panel flips `conf`.  P (the current
  // outer panel's accepted sub-panel sums, src/adaptive.jl:183-198 uses them) is only
  // materialised while an outer panel is being bisected: buffer `pc`, valid iff p_valid.
  int conf, has_acc, pc, p_valid;
  int crit, done, stack_len, tail_pending;
  unsigned long long max_err_bits;  // atomicMax over the bit pattern of |I2-I1| >= 0
  unsigned long long hi_new;        // atomicMax over i+1 of unconverged lags
  int nan1, nan2;
  long long n_outer, n_panels, n_accepted, sum_active;
  int trace_len;
  // batched march only (sk_batch_kernels.inc); there conf and pc run over three buffers
  long long n_steps, n_rerun;   // steps whose outcome was applied / discarded for a capped re-run
  long long sum_active_steps;   // sum over applied steps of active lags
  long long n_fix;              // steps that left an output fix behind (a guessed run cut short at its first panel)
  int spec_ok, fix_conf;        // the last completed outer panel retired nothing; buffer of the pending output fix
  long long fix_lo, fix_hi;     // lags [fix_lo, fix_hi) retired at slot 0 of a longer batch: outputs = sums in fix_conf
  TailScalars fix_tail;         //   + 2 x the truncation bound of that slot (src/adaptive.jl:194)
  // automatic batch width (sk_api.cpp): n_multi = steps built with more than one slot (applied or not); would_batch: a
  // one-panel march (K = 1 instantiation, dc.Kv > 1) reached a state at which the full width would have built one
  int n_multi, would_batch;
  int step_id, pad_step;         // the step this state describes: k_interp_b(t) evaluates nothing on any other (a skipped launch)
  BatchInfo bt;
};

// integrand handed to the panel rule (src/quadrature.jl:184-247): S(w) [times w^p log(w) factors as the
// panel requires], or the two integrands of the integration-by-parts formula for logw on [0, b]
enum { SK_G_DEFAULT = 0, SK_G_LOGA = 1, SK_G_LOGB = 2, SK_G_WEIGHT = 3 /* joint march: Gauss weight w^p only, stored with the node */ };
// what an evaluation kernel does with the panel sums: the usual epilogue, or store them raw (logw passes A, B)
enum { SK_RAW_NONE = 0, SK_RAW_A = 1, SK_RAW_B = 2 };

// how k_gen / k_advance obtain the state of their step (see step_state in sk_kernels.hip)
enum { SK_MODE_INIT = 0, SK_MODE_TRANSITION = 1, SK_MODE_ASIS = 2 };

struct TraceRec { double a, b, max_err, n_active, accepted; };

// Reductions of one k_interp launch, spread over 64 addresses to keep the atomics uncontended.
struct Reductions {
  unsigned long long max_bits[64];  // max |I2-I1| (bit pattern of a non-negative double)
  unsigned int hi[64];              // highest lag failing the convergence predicate, +1
  int nan;                          // bit 0: NaN in I1, bit 1: NaN in I2
  int pad[15];
};

// The same per slot of a batched step.
struct ReductionsB {
  unsigned long long max_bits[kBatchMax][64];
  unsigned int hi[kBatchMax][64];   // highest lag failing the convergence predicate of slot j, +1
  int nan[kBatchMax];               // per slot: bit 0 NaN in I1, bit 1 NaN in I2
  int pad[8];
  // the VALUE of that lag (bit pattern of a positive double; the lags ascend, so the maximum over the workgroups is the
  // lag with the highest index): the next panel's r_hi (src/adaptive.jl:152) without a dependent load in the transition
  unsigned long long rhi_bits[kBatchMax][64];
};

// Pinned, host-mapped block through which k_advance reports the state (polled by the host).
struct HostStatus {
  DevState state;
  int flags[2];       // k_check_sorted: [0] not strictly increasing, [1] negative or NaN lag
  // Hankel path: number of active lags with r <= r_hi / kHkBeta^l, l = 0..kHkMaxLevels (level planning on the host)
  long long lvl_cnt[kHkMaxLevels + 1];
  unsigned int seq;
  // streaming phase of the batched march (no k_advance): the spread launch of step t stores t + 1 in `progress` once it
  // has derived state_t, and when that state says "done" copies it into `state` and stores the call's tag in `done_seq`
  unsigned int progress;
  unsigned int done_seq;
  // one-panel launches inside a full-width march: peel_hint = BatchInfo::peel of the step that stored `progress` last
  // (written ahead of it); need_wide = t + 1: the one-panel launch of step t skipped it (BatchInfo::skip)
  unsigned int peel_hint;
  unsigned int need_wide;
};

struct DevBuffers {
  DevState* state;          // 2 slots (ping-pong)
  double2* stack;           // kStackCap (a,b) pairs
  TraceRec* trace;          // kTraceCap
  double2* src;             // 3M pre-phased source strengths (rule 1: M, rule 2: 2M)
  double2* fft;             // 2 x n2 complex (rule 1, rule 2)
  double2* fft_out;         // the transformed grids (out-of-place rocFFT plan: ~1 us faster than in place at this size)
  const double* fvals;      // callback mode: S(omega_j) for rule 1 (M) then rule 2 (2M)
  const double* xs;         // sorted unique lags
  double* ks; double* errs; // outputs (sorted-unique order): always the latest optimistic totals
  double* accK[2]; double* accE[2];  // ping-pong running sums of the active lags
  double* P[2];             // ping-pong outer-panel partial sums (bisection only)
  // per-workgroup partial reductions of k_interp (no same-address atomics): max |I2-I1| bits,
  // highest unconverged lag + 1, NaN flags (bit 0: I1, bit 1: I2)
  Reductions* red;          // 2 slots (step parity)
  HostStatus* host_status;  // device view of the pinned status block
  int* flags;               // input-check flags: [0] not strictly increasing, [1] negative or NaN lag
  int check_input;          // 1: the first k_gen of this call checks the lags (speculative sorted path)
  // Hankel path (dim >= 2)
  double2* hk_fft;          // kHkTerms x 2 x n2 complex grids of the current level
  double* hk_nodes;         // (kHkMaxLevels+1) x 2 x kHkNodes x kHkSplit partial near sums at the Chebyshev nodes
  double* hk_cheb;          // same shape: Chebyshev coefficients
  double* hk_dct;           // kHkNodes x kHkNodes exact-cosine DCT table [node][coefficient], then the kHkNodes node abscissae
  double* hk_mpart;         // inner box: per workgroup of k_hk_moments, ngm partial Chebyshev moments
  double* hk_mom;           // 2 x nb x ngm moments
  double2* hk_w;            // Q x 2 x nb x ngm equivalent sources (weight, node w) of the inner box
  double* hk_v;             // (Qn + Q) x 2 x kHkNodes node values: near box intervals, then inner box intervals
  double* hk_c2;            // the same shape: Chebyshev coefficients in r
  // logw first-panel passes (src/quadrature.jl:186-228): raw panel sums A (rules 1, 2) then B (rules 1, 2), n each
  double* lw_raw;
  // batched march: running sums / outer-panel partial sums over three buffers each (confirmed, after slot 0, after the
  // last slot), the spread grids of K slots ([slot][rule][even/odd][n1]) and their transforms
  double* bK[3]; double* bE[3]; double* bP[3];
  double2* bfft; double2* bfft_out;
  ReductionsB* redb;        // 2 slots (step parity)
  unsigned long long* stamps;  // diagnostic builds only (SK_STAMPS): 8 words per k_spread workgroup
  unsigned long long* stamps2; // diagnostic builds only: 16 words per k_interp2 workgroup
};

// launchers (sk_kernels.hip)
void launch_module_warmup(hipStream_t st);
void launch_empty(unsigned grid, hipStream_t st);
void launch_tail(const DevConst& dc, const DevBuffers& db, int slot, hipStream_t st);  // callback SDFs, dim > 1
void launch_gen(const DevConst& dc, const DevBuffers& db, int mode, int t, long long n, double k0, double ks0, int g,
                hipStream_t st);
void launch_spread(const DevConst& dc, const EsConst* ec, const DevBuffers& db, int slot, hipStream_t st);
void launch_bessel(const DevConst& dc, const DevBuffers& db, int slot, long long n, int nu, int raw, hipStream_t st);
// logw first panel: combine the raw sums of passes A and B with the boundary term (src/quadrature.jl:189,224-225)
void launch_lw_finish(const DevConst& dc, const DevBuffers& db, int slot, long long n, double B0, hipStream_t st);
// far sources (w >= om) of one level: all series terms spread in one launch
void launch_hk_spread(const DevConst& dc, const EsConst* ec, const DevBuffers& db, int slot, double om, int nt, hipStream_t st);
void launch_hk_near(const DevConst& dc, const DevBuffers& db, const HankelPlan& hp, hipStream_t st);
void launch_hk_boxes(const DevConst& dc, const DevBuffers& db, const HkInner& hi, hipStream_t st, int which = 3);  // bit 0 near, bit 1 inner
void hk_series_fill(HkSeries& ser, const double* a, int nu);
void launch_hk_far(const DevConst& dc, const EsConst* ec, const DevBuffers& db, const HkInner& hi, int slot, long long n, int raw,
                   hipStream_t st);
void launch_hk_iinterp(const DevConst& dc, const DevBuffers& db, const HkInner& hi, int slot, long long n, int raw, hipStream_t st);
int hk_inner_waves(const DevConst& dc, int nb);
void launch_hk_interp(const DevConst& dc, const EsConst* ec, const DevBuffers& db, const HankelPlan& hp, int level, int slot,
                      long long n, int raw, hipStream_t st);
void launch_interp(const DevConst& dc, const EsConst* ec, const DevBuffers& db, int slot, long long n, int raw, hipStream_t st);
void launch_advance(const DevConst& dc, const DevBuffers& db, int mode, int t, long long n, double k0, double ks0,
                    unsigned int seq, hipStream_t st);

// batched march (sk_batch_kernels.inc): node generation + spread of up to dc.K panels / interpolation + epilogue /
// round boundary
unsigned launch_spread_b_grid(const DevConst& dc);
void launch_spread_b(const DevConst& dc, const EsConst* ec, const DevBuffers& db, int mode, int t, long long n, double k0,
                     double ks0, unsigned int stream_seq, hipStream_t st);
// INVARIANT of every interpolation launch (k_interp2, k_interp_b, k_joint_interp): db.xs holds at least n valid doubles.
// The kernels request xs[min(i, n-1)] and the block's first / last lags AHEAD of the state (to overlap the two round
// trips), so even a launch whose state says "done" or belongs to another lag set reads them.
void launch_interp_b(const DevConst& dc, const EsConst* ec, const DevBuffers& db, int t, long long n, hipStream_t st);
void launch_advance_b(const DevConst& dc, const DevBuffers& db, int mode, int t, long long n, double k0, double ks0,
                      unsigned int seq, hipStream_t st);
// outputs of lags [lo, hi) from the running sums in buffer `conf` (a march that ended with a pending fix)
void launch_fix_b(const DevConst& dc, const DevBuffers& db, long long lo, long long hi, int conf, TailScalars tail, hipStream_t st);

// joint march (sk_joint_kernels.inc)
void launch_joint_spread(const DevConst& dc, const EsConst* ec, const DevBuffers& db, const JointSpec& js, const JointBufs& jb,
                         int slot, const double* fv /* host-callback integrand values, ntr x 3M, or nullptr */, hipStream_t st);
void launch_joint_interp(const DevConst& dc, const EsConst* ec, const DevBuffers& db, const JointSpec& js, const JointBufs& jb,
                         int slot, long long n, hipStream_t st);

// dim >= 2: strengths of every raw transform (fv != nullptr: integrand values from the host callbacks, ntr x 3M),
// then the direct Bessel sums of all transforms in one pass over the sources
void launch_joint_strengths(const DevConst& dc, const DevBuffers& db, const JointSpec& js, const JointBufs& jb, int slot,
                            const double* fv, hipStream_t st);
void launch_bessel_joint(const DevConst& dc, const DevBuffers& db, const JointSpec& js, const JointBufs& jb, int slot, long long n,
                         hipStream_t st);

// joint march through the fast Hankel transform (dim 2): (strength of transform tr, node) pairs for the Hankel kernels, and
// the joint epilogue on the raw panel sums raw[(tr 2 + rule) n + i] that those kernels stored
void launch_joint_pack(const DevConst& dc, const DevBuffers& db, const JointBufs& jb, int tr, double2* out, hipStream_t st);
void launch_joint_hk_epilogue(const DevConst& dc, const DevBuffers& db, const JointSpec& js, const JointBufs& jb, const double* raw,
                              int slot, long long n, hipStream_t st);

void launch_jac_assemble(const double* vals, long long n, int ncfg, const int* map, const double* dlag, int nwarp, int t_warp,
                         int ncols, double* jac, hipStream_t st);

// seam kernels for sk_panel_sums
void launch_seam_spread(const DevConst& dc, const double* nodes, const double* coef, int64_t M, double Cx, double X,
                        double kappa, double2* fft, hipStream_t st);
void launch_seam_deconv(const DevConst& dc, double2* fft, hipStream_t st);
void launch_seam_interp(const DevConst& dc, const EsConst* ec, const double2* fft, const double* xs, int64_t n, double Cx, double r_hi,
                        double tau, double theta_scale, double* out_re, double* out_im, hipStream_t st);
void launch_direct(const double* nodes, const double* coef, int64_t M, const double* xs, int64_t n, double* out_re,
                   double* out_im, hipStream_t st);

// callers' side: pairwise lags and the dense covariance matrix (src/model.jl:64-65, src/utils.jl:41-64)
void launch_pair_lags(const double* pts, long long npts, int pdim, const long long* pairs, long long npairs, double* lags,
                      hipStream_t st, long long e0 = 0);
void launch_fill_cov(const double* vals, long long npts, double* out, hipStream_t st);

// sort / dedup helpers
void launch_check_sorted(const double* xs, long long n, int* flags, hipStream_t st);  // flags[0]=not strictly increasing, flags[1]=invalid (<0 or NaN)
size_t sort_unique_temp_bytes(long long n);
// returns number of unique values in *n_unique (device int64) and fills uxs, inv
void sort_unique(const double* xs, long long n, double* uxs, long long* inv, long long* d_n_unique, void* temp,
                 size_t temp_bytes, hipStream_t st, bool use_lib);
// sort_unique + the count on the host (synchronises `st`; h_nu pinned); the library sort when the bucket sort gave up
int sort_unique_sync(const double* xs, long long n, double* uxs, long long* inv, long long* d_n_unique, long long* h_nu,
                     void* temp, size_t temp_bytes, hipStream_t st, long long* nu, int* was_unsorted = nullptr,
                     int* invalid = nullptr, bool early = false);  // h_nu: 3 pinned words
void launch_gather(const double* ks, const double* errs, const long long* inv, long long n, double* vals, double* errs_out,
                   hipStream_t st);

}  // namespace sk
