"""Host-array entry (sk_kernel_values: H2D + march + D2H) timing on the C2 workload (diagnostics)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5))
k0 = np.pi / 2
for n in (10 ** 4, 10 ** 5, 10 ** 6, 10 ** 7):
    rs = 10.0 ** np.linspace(-6, 0, n)
    for _ in range(3):
        ks, errs = sk.kernel_values(cfg, rs, k0=k0)
    t = time.perf_counter()
    reps = 20
    for _ in range(reps):
        ks, errs = sk.kernel_values(cfg, rs, k0=k0)
    dt = (time.perf_counter() - t) / reps
    truth = k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
    print("n=%.0e host arrays: %.3f ms/call = %.2e lags/s (max err %.1e)" % (n, dt * 1e3, n / dt, np.abs(ks - truth).max() / k0), flush=True)
