"""C4 through the host-array entry (pageable lags in, pinned-pool results out), a few calls: run under rocprofv3."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
import bench
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(0.7, 1.3, 1.8, 2.0), tol=1e-8, dim=2)
k0 = sk.compute_k0(cfg)
rs = bench.shard_grid(bench.WORKLOADS["C4"], 0, 1)
for _ in range(10):
    ks, es = sk.kernel_values(cfg, rs, k0=k0)
print(sk.host_timing(cfg), cfg.last_stats())
