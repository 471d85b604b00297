"""Diagnostic: marches in which the one-panel launches inside a full-width march (sk_api.cpp, hybrid) occur and are
skipped -- random catalogue SDFs / tolerances / small rules / lag sets; every case is compared bitwise with the explicit
full width.  usage: hybrid_search.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(5)
bad = 0
for c in range(ncase):
    qs = [(64, 4), (128, 8), (256, 16), (512, 16), (4096, 16)][rng.integers(0, 5)]
    tol = 10.0 ** -rng.integers(6, 13)
    prm = (float(rng.uniform(0.5, 3)), float(rng.uniform(0.2, 3)), float(rng.uniform(0.6, 2.5)))
    n = int(rng.integers(2000, 60000))
    kind = rng.integers(0, 3)
    rs = np.sort(rng.uniform(0, rng.uniform(0.5, 8), n)) if kind == 0 else (10.0 ** np.sort(rng.uniform(-5, 1, n)) if kind == 1 else np.linspace(0, float(rng.uniform(1, 6)), n))
    rs = np.unique(rs)
    a = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), tol=tol, quadspec=qs)
    k0 = sk.compute_k0(a)
    kf = 2 if qs[0] * qs[1] > 40000 else 4
    b = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), tol=tol, quadspec=qs, batch_panels=kf)
    try:
        ka, ea = sk.kernel_values(a, rs, k0=k0)
        sa, ha = a.last_stats(), sk.step_kinds(a)
        kb, eb = sk.kernel_values(b, rs, k0=k0)
    except Exception as e:
        print("case", c, "error", str(e)[:80])
        continue
    same = np.array_equal(ka, kb) and np.array_equal(ea, eb, equal_nan=True)
    bad += not same
    print("case %2d qs %s tol %.0e n %5d kind %d: steps %3d launched %3d one-panel %3d skipped %d %s" % (
        c, qs, tol, rs.size, kind, sa["n_steps"], sa["n_steps_launched"], ha["one_panel_launches"], ha["skipped"], "same bits" if same else "DIFFERENT"),
        "prm", tuple(round(x, 4) for x in prm), flush=True)
print("different:", bad)
