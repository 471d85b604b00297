"""BASELINE config 5, a few device-resident calls and nothing else (run under rocprofv3 --kernel-trace for a timeline)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(2.14, 0.97, 0.89), tol=1e-12, quadspec=(256, 16))
k0 = sk.compute_k0(cfg)
rs = np.linspace(0.0, 5.1, 10 ** 5)
n = rs.size
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
for _ in range(8):
    sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
sk.synchronize()
print(cfg.last_stats(), sk.step_kinds(cfg))
