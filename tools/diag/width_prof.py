"""One NUFFT width on C2 for rocprofv3 (kernel times per width): width_prof.py <w>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
w = int(sys.argv[1])
n = 10 ** 6
rs = 10.0 ** np.linspace(-6, 0, n)
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5), nufft_width=w)
for _ in range(150):
    sk.kernel_values_device(cfg, d_rs.ptr, n, np.pi / 2, d_ks.ptr, d_es.ptr)
sk.synchronize()
