"""Per-call wall time of a dim-2 march, 300 calls: prints the outliers (a ~90 ms stall was seen once in a while)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(0.7, 1.3, 1.8, 2.0), tol=1e-12, dim=2, quadspec=(256, 16))
k0 = sk.compute_k0(cfg)
rs = np.linspace(0.0, 3.0, 200001)[1:]
n = rs.size
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
ts = []
for i in range(300):
    t = time.perf_counter()
    sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    sk.synchronize()
    ts.append(1e3 * (time.perf_counter() - t))
ts = np.array(ts)
print("median %.3f ms, max %.3f at call %d; calls above 3x median: %s" % (np.median(ts), ts.max(), int(ts.argmax()),
      [(int(i), round(float(ts[i]), 2)) for i in np.where(ts > 3 * np.median(ts))[0]]))
