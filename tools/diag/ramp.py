"""Per-step time of the first device-resident steps after set-up (how long the GPU takes to reach its steady clocks):
usage: ramp.py [C2|C5] [idle_ms]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
w = bench.WORKLOADS[wl]
kind, prm = w["sdf"]
cfg = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), **w["kw"])
k0 = sk.compute_k0(cfg)
rs = bench.shard_grid(w, 0, 1)
n = rs.size
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
step = lambda: sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
for rep in range(3):
    time.sleep(idle / 1e3)
    ts = []
    for _ in range(80):
        t0 = time.perf_counter(); step(); ts.append(1e6 * (time.perf_counter() - t0))
    print("%s rep %d after %.0f ms idle, us per step: first 10 %s | 10-20 %.1f | 20-40 %.1f | 40-80 %.1f" % (
        wl, rep, idle, [round(x) for x in ts[:10]], np.mean(ts[10:20]), np.mean(ts[20:40]), np.mean(ts[40:])))
