"""Diagnostic: the dim-1 joint march (5 integrands: K, dK/dsigma, dK/drho, dK/dnu, K') on BASELINE config 2's lags, ms per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
prm = (1.0, 1.0, 1.5)
base = sk.SDF.matern(*prm)
cfgs = [sk.AdaptiveKernelConfig(base)] + [sk.AdaptiveKernelConfig(base.d(j)) for j in (0, 1, 2)] + [sk.AdaptiveKernelConfig(base, derivative=True)]
rs = 10.0 ** np.linspace(-6, 0, 10 ** 6)
k0 = np.pi / 2
for _ in range(3):
    v, e = sk.kernel_values_joint(cfgs, rs, k0)
t = time.perf_counter()
for _ in range(10):
    v, e = sk.kernel_values_joint(cfgs, rs, k0)
dt = (time.perf_counter() - t) / 10
print("joint march, 5 integrands, 1e6 lags: %.3f ms per call (host arrays), checksum %.12g" % (1e3 * dt, float(np.nansum(v))), cfgs[0].last_stats()["n_panels"])
cfgs[0].profile(True)
sk.kernel_values_joint(cfgs, rs, k0)
print({k: round(1e3 * x["total_ms"] / max(1, x["launches"]), 1) for k, x in cfgs[0].kernel_times().items() if x["launches"]})
