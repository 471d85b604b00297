"""Trace of one fixture with batch_panels = 1, 2, 4 (round-3 bring-up of the batched march)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralkernels_jl_amd as sk
z = np.load(os.path.join(ROOT, "tests", "golden", "bisect_narrow_matern.npz"))
k0 = float(z["k0"])
for bp in (1, 2, 4):
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1.0, 1e-3, 1.5), quadspec=(64, 4), batch_panels=bp)
    ks, errs = sk.kernel_values(cfg, z["rs"], k0=k0)
    tr = cfg.last_trace()
    print("batch_panels", bp, "max|ks-fix|/k0 %.2e" % (np.nanmax(np.abs(ks - z["ks"])) / k0), cfg.last_stats())
    for i, r in enumerate(tr):
        ref = z["trace"][i] if i < len(z["trace"]) else None
        print("  %2d [%.6g, %.6g] err %.4e acc %d   | ref err %s acc %s" % (i, r["a"], r["b"], r["max_err"], r["accepted"],
              ("%.4e" % ref[2]) if ref is not None else "-", int(ref[4]) if ref is not None else "-"))
