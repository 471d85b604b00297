"""Steps / re-runs of the batched march on C5- and C3-like configurations for batch_panels = 1, 2, 4, 8."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralkernels_jl_amd as sk
from oracle import sk_oracle as oracle
cases = {
    "C5": (dict(tol=1e-12, quadspec=(256, 16)), (2.14, 0.97, 0.89), np.linspace(0, 5.1, 100000), None),
    "C2": (dict(), (1.0, 1.0, 1.5), 10.0 ** np.linspace(-6, 0, 1000000), np.pi / 2),
}
rho, nu, al = 10.0, 0.5, 0.4
phi = 1.0 / oracle.sing_matern_k0(al, (1.0, rho, nu))
cases["C3"] = (dict(alpha=al, quadspec=(256, 16)), (phi, rho, nu), 10.0 ** np.linspace(-8, -2, 1000000), 1.0)
for name, (kw, prm, rs, k0) in cases.items():
    if k0 is None:
        k0 = float(oracle.matern_cov(0.0, prm))
    ref = None
    for bp in (1, 2, 4, 8):
        cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), batch_panels=bp, **kw)
        ks, errs = sk.kernel_values(cfg, rs, k0=k0)
        t0 = time.perf_counter()
        for _ in range(20):
            ks, errs = sk.kernel_values(cfg, rs, k0=k0)
        dt = (time.perf_counter() - t0) / 20
        st = cfg.last_stats()
        if ref is None:
            ref = (ks, errs)
        print(name, "K", bp, "host-to-host %.3f ms" % (1e3 * dt), {k: st[k] for k in ("n_outer", "n_panels", "n_steps", "n_rerun", "n_steps_launched", "n_syncs")},
              "max|dks|/k0 %.2e max|derrs|/k0 %.2e" % (np.nanmax(np.abs(ks - ref[0])) / k0, np.nanmax(np.abs(errs - ref[1])) / k0), flush=True)
