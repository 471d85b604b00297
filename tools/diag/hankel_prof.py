"""C4 (or log-spaced lags, argv[1] = log) through the Hankel path, a few calls: run under rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
import bench
prm = (0.7, 1.3, 1.8, 2.0)
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), tol=1e-8, dim=2)
k0 = sk.compute_k0(cfg)
rs = 10.0 ** np.linspace(-6, 0, 10 ** 6) if len(sys.argv) > 1 and sys.argv[1] == "log" else bench.shard_grid(bench.WORKLOADS["C4"], 0, 1)
n = rs.size
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
for _ in range(10):
    sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
sk.synchronize()
print(cfg.last_stats())
