"""Diagnostic: one joint march (5 integrands, C2 lags) with the SK_STAMPS build; phase times of k_spread_joint's
workgroups (one tile each).  Ticks of s_memtime."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SPECTRALKERNELS_HIP_LIB"] = os.path.join(ROOT, "tools", "diag", "libsk_stamps.so")
os.environ["SK_STAMP_FILE"] = "/tmp/stamps.bin"
import spectralkernels_jl_amd as sk
f = sk.SDF.matern(1.0, 1.0, 1.5)
cfgs = [sk.AdaptiveKernelConfig(f), sk.AdaptiveKernelConfig(f, derivative=True), sk.AdaptiveKernelConfig(f.d(0)),
        sk.AdaptiveKernelConfig(f.d(1)), sk.AdaptiveKernelConfig(f.d(2))]
pds = [False, False, True, True, True]
rs = 10.0 ** np.linspace(-6, 0, 10 ** 6)
for _ in range(3):
    sk.kernel_values_joint(cfgs, rs, np.pi / 2, param_derivative=pds)
a = np.fromfile("/tmp/stamps.bin.joint", dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] > 0].astype(np.int64)
t0 = a[:, 0].min()
d = np.diff(a[:, :6], axis=1)
print("workgroups", len(a), "kernel span %d ticks" % (a[:, 5].max() - t0))
print("entry quantiles", np.percentile(a[:, 0] - t0, [0, 25, 50, 75, 90, 100]))
print("phase means: header+tables %.0f | stage batch 0 %.0f | gather batch 0 %.0f | other batches %.0f | output %.0f | total %.0f" % (
    *d.mean(axis=0), (a[:, 5] - a[:, 0]).mean()))
for lo, hi in ((0, 64), (64, 129), (129, 257), (257, 10 ** 6)):
    m = (a[:, 6] >= lo) & (a[:, 6] < hi)
    if m.any():
        print("  sources in [%d,%d): n=%d, phases %s, total %.0f" % (lo, hi, m.sum(), d[m].mean(axis=0).round(0), (a[m, 5] - a[m, 0]).mean()))
print("end quantiles", np.percentile(a[:, 5] - t0, [0, 25, 50, 75, 90, 100]))
