"""NUFFT width against accuracy and time on BASELINE config 2 (the default width follows tol / 1000)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
n = 10 ** 6
rs = 10.0 ** np.linspace(-6, 0, n)
k0 = np.pi / 2
truth = k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
ref = None
for w in (16, 12, 10, 8):
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5), nufft_width=w)
    step = lambda: sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    for _ in range(20):
        step()
    sk.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        step()
    sk.synchronize()
    dt = (time.perf_counter() - t) / 200
    ks = d_ks.to_host()
    if ref is None:
        ref = ks.copy()
    st = cfg.last_stats()
    print("w=%2d: %.4f ms/step, panels %d, sum_active %d, max err vs closed form %.2e, vs w=16 %.2e (all over K0), max errs %.2e" % (
        w, 1e3 * dt, st["n_panels"], st["sum_active"], np.abs(ks - truth).max() / k0, np.abs(ks - ref).max() / k0, d_es.to_host().max() / k0), flush=True)
