"""Throughput of the C2 workload at larger lag counts (diagnostics): python tools/diag/bign.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5))
k0 = np.pi / 2
for n in (10 ** 5, 10 ** 6, 10 ** 7, 3 * 10 ** 7, 10 ** 8):
    rs = 10.0 ** np.linspace(-6, 0, n)
    d = sk.DeviceArray.from_host(rs); v = sk.DeviceArray(n); e = sk.DeviceArray(n)
    f = lambda: sk.kernel_values_device(cfg, d.ptr, n, k0, v.ptr, e.ptr)
    for _ in range(3): f()
    sk.synchronize(); t = time.perf_counter()
    reps = 20 if n <= 10 ** 7 else 5
    for _ in range(reps): f()
    sk.synchronize(); dt = (time.perf_counter() - t) / reps
    cfg.profile(True); f(); kt = cfg.kernel_times(); cfg.profile(False)
    it = kt["interp"]; us = 1e3 * it["total_ms"] / it["launches"]
    truth = k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
    err = np.abs(v.to_host() - truth).max() / k0
    print("n=%.0e: %.3f ms/call, %.2e lags/s; k_interp %.1f us/launch = %.2f TB/s algorithmic (40 B/lag); max err %.1e; %s" % (
        n, dt * 1e3, n / dt, us, 40.0 * n / (us * 1e-6) / 1e12, err, cfg.last_stats()["n_panels"]), flush=True)
    d.free(); v.free(); e.free()
