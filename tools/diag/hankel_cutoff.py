"""dim 2: direct Bessel sums against the Hankel path by boxes over the number of lags, for two rule sizes (the cut-over
`hankel_min_lags` defaults to)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
prm = (0.7, 1.3, 1.8, 2.0)
for qs in ((4096, 16), (256, 16), (64, 8)):
    for n in (100, 300, 1000, 3000, 10000, 30000):
        rs = np.sort(np.random.default_rng(n).uniform(0.0, 1.4, n))
        out = []
        for hk in (-1, 1):
            cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), tol=1e-8, dim=2, quadspec=qs, hankel_min_lags=hk)
            k0 = sk.compute_k0(cfg)
            d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
            for _ in range(3):
                sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
            sk.synchronize()
            reps = 10
            t = time.perf_counter()
            for _ in range(reps):
                sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
            sk.synchronize()
            out.append((1e3 * (time.perf_counter() - t) / reps, d_ks.to_host(), cfg.last_stats()["n_panels"]))
        print("quadspec %s n %6d: direct %8.3f ms  boxes %8.3f ms  panels %d/%d  max diff %.1e" % (
            qs, n, out[0][0], out[1][0], out[0][2], out[1][2], np.abs(out[0][1] - out[1][1]).max() / abs(k0)), flush=True)
