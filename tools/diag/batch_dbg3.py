import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralkernels_jl_amd as sk
from oracle import sk_oracle as oracle
prm = (2.14, 0.97, 0.89)
rs = np.linspace(0, 5.1, 100000)
k0 = float(oracle.matern_cov(0.0, prm))
for bp in (4, 1, 0, 4):
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), batch_panels=bp, tol=1e-12, quadspec=(256, 16))
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        ks, errs = sk.kernel_values(cfg, rs, k0=k0)
        ts.append(1e3 * (time.perf_counter() - t0))
    print("K", bp, ["%.2f" % t for t in ts], flush=True)
    cfg.profile(True)
    sk.kernel_values(cfg, rs, k0=k0)
    print("   ", {k: (round(v["total_ms"], 3), v["launches"]) for k, v in cfg.kernel_times().items()}, flush=True)
    cfg.profile(False)
    del cfg
