import sys, os, numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5))
for n in (1000, 100000, 10 ** 6):
    rs = 10.0 ** np.linspace(-6, 0, n)
    ks, errs = sk.kernel_values(cfg, rs, k0=np.pi / 2)
    truth = np.pi / 2 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
    print(n, np.abs(ks - truth).max(), cfg.last_stats(), flush=True)
