"""Host-array entry on BASELINE config 2: time split of sk_kernel_values (staged upload / march / download)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5))
rs = 10.0 ** np.linspace(-6, 0, n)
k0 = np.pi / 2
vals, errs = np.zeros(n), np.zeros(n)
for name, xs, v, e in (("pageable", rs, vals, errs),):
    for _ in range(5):
        sk.kernel_values_into(cfg, xs, k0, v, e)
    ts = []
    for _ in range(30):
        t = time.perf_counter(); sk.kernel_values_into(cfg, xs, k0, v, e); ts.append(time.perf_counter() - t)
    print(name, "threads", os.environ.get("SK_COPY_THREADS", "8"), "median %.3f ms min %.3f ms" % (1e3 * np.median(ts), 1e3 * min(ts)), sk.host_timing(cfg))
px, pk, pe = sk.pinned_empty(n), sk.pinned_empty(n), sk.pinned_empty(n)
px[:] = rs
for _ in range(5):
    sk.kernel_values_into(cfg, px, k0, pk, pe)
ts = []
for _ in range(30):
    t = time.perf_counter(); sk.kernel_values_into(cfg, px, k0, pk, pe); ts.append(time.perf_counter() - t)
print("pinned median %.3f ms min %.3f ms" % (1e3 * np.median(ts), 1e3 * min(ts)), sk.host_timing(cfg), np.array_equal(pk, vals))
truth = k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
print("err", np.abs(vals - truth).max() / k0)
