"""Diagnostic: many configs created (and destroyed) in one process -- ms per call of each (a fitting loop rebuilds its
config every iteration, gen_new_sdf_config).  usage: cfg_churn.py [keep|del] [count]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
mode = sys.argv[1] if len(sys.argv) > 1 else "del"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = 10 ** 5
rs = np.linspace(0.0, 5.1, n)
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
keep = []
for i in range(count):
    t0 = time.perf_counter()
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(2.14, 0.97, 0.89), tol=1e-12, quadspec=(256, 16))
    t_create = time.perf_counter() - t0
    k0 = sk.compute_k0(cfg)
    step = lambda: sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    t0 = time.perf_counter()
    step()
    sk.synchronize()
    t_first = time.perf_counter() - t0
    per = []
    for _ in range(6):
        t0 = time.perf_counter()
        step()
        sk.synchronize()
        per.append(1e3 * (time.perf_counter() - t0))
    st = cfg.last_stats()
    print("config %2d: create %.1f ms, first call %.2f ms, then %s | syncs %d launched %d steps %d" % (
        i, 1e3 * t_create, 1e3 * t_first, " ".join("%.3f" % p for p in per), st["n_syncs"], st["n_steps_launched"], st["n_steps"]), flush=True)
    if mode == "keep":
        keep.append(cfg)
    else:
        del cfg
