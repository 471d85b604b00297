// sk_config.h -- the config handle behind the C ABI (private to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <atomic>
#include <string>
#include <vector>
#include "sk_device.h"
#include "sk_internal.h"

#define HIPCHK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return sk::fail(SK_EHIP, std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);    \
  } while (0)
#define FFTCHK(expr)                                                                       \
  do {                                                                                     \
    rocfft_status s_ = (expr);                                                             \
    if (s_ != rocfft_status_success)                                                       \
      return sk::fail(SK_EHIP, std::string("rocFFT error ") + std::to_string((int)s_) + " at " #expr); \
  } while (0)

using namespace sk;

struct sk_config {
  sk_config_args args;
  int64_t m, k, M;
  double p, c;
  int device;
  int warned;
  SdfHost sdf, dsdf;
  // host rules
  std::vector<double> leg[4], jac[4];
  EsTables tab;
  // device
  DevConst dc;
  EsConst ec;
  EsConst* d_ec = nullptr;
  DevBuffers db;
  std::vector<void*> owned;  // device allocations freed at destroy
  rocfft_plan plan = nullptr;
  rocfft_plan plan_b = nullptr;     // batched march: 4 K transforms of length n1 (bfft -> bfft_out)
  rocfft_execution_info info_b = nullptr;
  void* fft_work_b = nullptr;
  // automatic batch_panels only: a config whose last full-width march never built a step of more than one panel (each
  // panel retires lags or is bisected on its own) runs the K = 1 instantiations -- lighter kernels, the same bits; the
  // one-panel march notices an input that would batch (DevState::would_batch) and is then repeated at the full width
  int k_full = 0;                   // dc.K of the full batch
  bool k_adaptive = false;
  bool narrow = false;              // the last full-width march built one-panel steps only
  rocfft_plan plan_b1 = nullptr;    // 4 transforms of length n1 (K = 1)
  rocfft_execution_info info_b1 = nullptr;
  void* fft_work_b1 = nullptr;
  rocfft_plan plan_seam = nullptr;  // sk_panel_sums
  rocfft_execution_info info_seam = nullptr;
  void* fft_work_seam = nullptr;
  rocfft_plan plan_hk = nullptr;   // batch 2*hk_terms(w), Hankel path
  void* fft_work_hk = nullptr;
  rocfft_execution_info info_hk = nullptr;
  // the same transform with fewer series terms (batch 2 nt), created on first use
  rocfft_plan plan_hk_nt[kHkTerms + 1] = {};
  rocfft_execution_info info_hk_nt[kHkTerms + 1] = {};
  void* fft_work_hk_nt[kHkTerms + 1] = {};
  double hk_a[kHkTerms], hk_a2[kHkTerms];  // a_k(nu) of the main order and of dim/2 (logw pass B)
  int hk_nb = 0;                   // 1: the Hankel path runs by boxes (sk_hankel_inner.inc); 0: level split only (SK_HK_LEVELS)
  int hk_ngmax = 0;                // moments per band the buffers hold
  size_t hk_qbytes = 0;            // bytes of hk_qbuf (grow only)
  void* hk_qbuf = nullptr;         // hk_w / hk_v / hk_c2
  rocfft_execution_info info = nullptr;
  void* fft_work = nullptr;
  double* d_fvals = nullptr;
  // per-call workspace (grow only)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  double* d_io = nullptr;          // staging of sk_kernel_values (host arrays)
  size_t io_bytes = 0;
  // pinned host staging of the host-array entries (sk_hostio.cpp): lags in, values and error estimates out
  double* h_in = nullptr;
  size_t hin_bytes = 0;
  double* h_out = nullptr;
  size_t hout_bytes = 0;
  std::vector<std::atomic<int>> chunk_done;  // threads done with chunk c of the staged upload
  std::vector<size_t> chunk_edge;            // chunk boundaries of the transfer in progress
  std::vector<hipEvent_t> cp_events;         // one per download chunk
  double host_timing[4] = {0, 0, 0, 0};      // last sk_kernel_values call: upload, march, download, total (ms)
  bool expect_unsorted = false;              // the last device-resident input of unknown order (expect_n lags) was unsorted
  long long expect_n = -1;
  long long last_n_steps = 0;                // steps of the last batched march on this config (0: none yet)
  int copy_threads = 0;                      // staging threads of this config (0: SK_COPY_THREADS / 8); sharded: cores / shards
  // joint march (sk_joint.cpp): grow-only buffers, one rocFFT plan per number of raw transforms
  void* d_joint = nullptr;
  size_t joint_bytes = 0;
  double* d_jfv = nullptr;         // joint march on callback SDFs: integrand values of a panel, ntr x 3M (device / pinned)
  double* h_jfv = nullptr;
  size_t jfv_count = 0;
  rocfft_plan plan_joint[kJointTr + 1] = {};
  rocfft_execution_info info_joint[kJointTr + 1] = {};
  void* fft_work_joint[kJointTr + 1] = {};
  // grow-only device scratch of the multi / pairs / dense entries (points, index pairs, per-config results)
  void* d_aux = nullptr;
  size_t aux_bytes = 0;
  double* d_lags = nullptr;        // values / errors staging of the pair entries (3 n doubles)
  size_t lags_bytes = 0;
  double* d_plags = nullptr;       // pairwise lags generated on the device
  size_t pl_bytes = 0;
  HostStatus* h_status = nullptr;  // pinned + device-mapped, written by k_advance
  unsigned int seq = 0;
  size_t trace_len = 0;
  bool trace_valid = false;
  long long* h_nuniq = nullptr;
  long long hyb_narrow = 0, hyb_skips = 0;  // last call: steps launched with the one-panel kernels inside a full-width march / skipped by them
  double* h_scal = nullptr;        // pinned scalars handed to small asynchronous uploads
  double* h_fvpin[2] = {nullptr, nullptr};  // callback SDFs: pinned copies of the integrand values of a pass
  hipStream_t own_stream = nullptr;
  hipStream_t copy_stream = nullptr;  // bulk upload of the host-array entry, beside the first step (up_event)
  hipEvent_t up_event = nullptr;
  hipStream_t side_stream = nullptr;  // Hankel path: the boxes run beside the NUFFT (fork / join events): inner box
  hipStream_t side_stream2 = nullptr; //   ... and the near box, which is all the far kernel waits for
  hipEvent_t fork_event = nullptr, join_event = nullptr, join_event2 = nullptr;
  // last call
  sk_stats stats;
  std::vector<TraceRec> trace;
  // callback scratch
  std::vector<double> h_nodes, h_fvals, h_fvals2, h_dfvals;
  // optional per-kernel event timing
  int profile = 0;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct EvRec { int kid; hipEvent_t a, b; };
  std::vector<EvRec> ev_recs;
  sk_kernel_times ktimes;
};

namespace sk {
// what the upload pass found out about the lags (src/adaptive.jl:99,113): strictly increasing (hence unique), not,
// or containing a negative / NaN entry; UNKNOWN (pinned or device-resident input) leaves the check to the device
enum { SK_ORDER_UNKNOWN = 0, SK_ORDER_SORTED = 1, SK_ORDER_UNSORTED = 2, SK_ORDER_INVALID = 3 };
// sk_hostio.cpp: transfers between caller-owned host arrays and the device (pinned staging, chunked and overlapped)
int h2d_doubles(sk_config* cfg, double* d_dst, const double* src, size_t cnt, hipStream_t st, int* order);
int d2h_doubles2(sk_config* cfg, double* dst_a, double* dst_b, const double* d_src, size_t cnt_each, hipStream_t st);
int d2h_doubles(sk_config* cfg, double* dst, const double* d_src, size_t cnt, hipStream_t st);
// sk_api.cpp: plan and launches of one step of the Hankel path by boxes (shared by run_core and the joint march)
int hk_plan_boxes(sk_config* cfg, DevBuffers& db, const DevState* hs, int nu, const double* a_nu, int raw, hipStream_t st,
                  HkInner& hin, bool& boxes);
int hk_launch_boxes(sk_config* cfg, DevBuffers& db, const HkInner& hin, int t, long long n, int raw, hipStream_t st);
// sk_api.cpp: one march per config on device-resident lags of any order (ncfg x n results on the device, input order)
int multi_device_core(sk_config* const* cfgs, size_t ncfg, const double* d_lags, size_t n, double k0,
                      const int* param_derivative, double** d_vals, double** d_errs);
}  // namespace sk
