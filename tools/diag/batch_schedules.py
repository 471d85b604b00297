"""Panel schedules of the bench workloads under batch_panels = 1, 2, 4, 8 (what batching can and cannot save)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralkernels_jl_amd as sk
from oracle import sk_oracle as oracle
rng = np.random.default_rng(11)
rs = np.sort(rng.uniform(0.0, 5.1, 300))
parms = (2.14, 0.97, 0.89)
tol = 1e-12
k0 = float(oracle.matern_cov(0.0, parms))
ocfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms), tol=tol, quadspec=(256, 16), nthreads=8)
oks, oerrs, otr = oracle.kernel_values(ocfg, rs, k0=k0, return_trace=True)
for bp in (1, 2, 4):
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*parms), tol=tol, quadspec=(256, 16), batch_panels=bp)
    ks, errs = sk.kernel_values(cfg, rs, k0=k0)
    tr = cfg.last_trace()
    print("K", bp, "max|hip-oracle|/k0 %.3e" % (np.abs(ks - oks).max() / k0), "panels hip", len(tr), "oracle", len(otr), cfg.last_stats())
    for i in range(max(len(tr), len(otr))):
        a = tr[i] if i < len(tr) else None
        b = otr[i] if i < len(otr) else None
        flag = "" if (a and b and a["n_active"] == b["n_active"] and a["b"] == b["b"]) else "   <<<"
        print("   %2d hip %s | oracle %s%s" % (i, ("[%.6g,%.6g] act %d" % (a["a"], a["b"], a["n_active"])) if a else "-",
                                              ("[%.6g,%.6g] act %d" % (b["a"], b["b"], b["n_active"])) if b else "-", flag))
