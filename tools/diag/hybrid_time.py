"""Diagnostic: ms per call of a few multi-step marches (run with and without SK_NO_HYBRID=1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
CASES = [
    ("default rule, 1e6 log lags to 10", (2.05, 2.6, 0.67), dict(tol=1e-8), lambda: 10.0 ** np.linspace(-5, 1, 10 ** 6)),
    ("default rule, 1e6 uniform lags to 8, tol 1e-10", (1.2, 1.9, 0.8), dict(tol=1e-10), lambda: np.linspace(0, 8, 10 ** 6)),
    ("(128,8) tol 1e-10, 15k log lags", (2.99, 2.61, 0.68), dict(tol=1e-10, quadspec=(128, 8)), lambda: 10.0 ** np.linspace(-5, 1, 15000)),
    ("(512,16) tol 1e-12, 45k log lags", (1.22, 2.06, 0.9), dict(tol=1e-12, quadspec=(512, 16)), lambda: 10.0 ** np.linspace(-5, 1, 45000)),
]
for name, prm, kw, mk in CASES:
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), **kw)
    k0 = sk.compute_k0(cfg)
    rs = mk(); n = rs.size
    d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
    for _ in range(5):
        sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    sk.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    sk.synchronize()
    st = cfg.last_stats()
    print("%-48s %.4f ms/call  steps %d panels %d %s" % (name, 1e3 * (time.perf_counter() - t) / 100, st["n_steps"], st["n_panels"], sk.step_kinds(cfg)), flush=True)
