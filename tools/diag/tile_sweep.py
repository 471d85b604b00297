"""Diagnostic: spread tile geometry (SK_TILE_CELLS / SK_TILE_SRC, read by build_tiles in sk_api.cpp when a config is
created) against ms per call, over quadspecs and tolerances.  usage: tile_sweep.py [n_lags]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 5
rs = np.linspace(0.0, 5.1, n)
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
cells = [int(x) for x in os.environ.get("CELLS", "64 48 32 24 16").split()]
srcs = [int(x) for x in os.environ.get("SRCS", "256 192 128 96 64").split()]
for tol in (1e-8, 1e-12):
    for qs in ((64, 4), (128, 8), (256, 16), (512, 16), (1024, 16), (2048, 16), (4096, 16)):
        res = {}
        for c in cells:
            for s in srcs:
                os.environ["SK_TILE_CELLS"], os.environ["SK_TILE_SRC"] = str(c), str(s)
                cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(2.14, 0.97, 0.89), tol=tol, quadspec=qs)
                k0 = sk.compute_k0(cfg)
                step = lambda: sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
                for _ in range(4):
                    step()
                sk.synchronize()
                t = time.perf_counter()
                for _ in range(20):
                    step()
                sk.synchronize()
                res[(c, s)] = 1e3 * (time.perf_counter() - t) / 20
                st = cfg.last_stats()
                del cfg
        best = min(res, key=res.get)
        print("tol %g quadspec %s steps %d: default(64,256) %.4f  best %s %.4f | " % (tol, qs, st["n_steps"], res[(64, 256)], best, res[best])
              + " ".join("%d/%d:%.3f" % (c, s, res[(c, s)]) for c in cells for s in srcs), flush=True)
