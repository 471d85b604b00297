"""Joint march of 5 integrands on the C2 lags (run under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
f = sk.SDF.matern(1.0, 1.0, 1.5)
cfgs = [sk.AdaptiveKernelConfig(f), sk.AdaptiveKernelConfig(f, derivative=True), sk.AdaptiveKernelConfig(f.d(0)),
        sk.AdaptiveKernelConfig(f.d(1)), sk.AdaptiveKernelConfig(f.d(2))]
pds = [False, False, True, True, True]
n = 10 ** 6
rs = 10.0 ** np.linspace(-6, 0, n)
k0 = np.pi / 2
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name, fn in (("joint", lambda: sk.kernel_values_joint(cfgs, rs, k0, param_derivative=pds)),
                 ("multi", lambda: sk.kernel_values_multi(cfgs, rs, [k0] * 5, param_derivative=pds))):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print(name, "%.3f ms per call" % (1e3 * (time.perf_counter() - t0) / reps), cfgs[0].last_stats())
