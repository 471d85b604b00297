import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spectralkernels_jl_amd as sk
from oracle import sk_oracle as o

print("device", sk.device_available())
t = time.time()
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5), quadspec=(256, 16))
print("config create %.2fs" % (time.time() - t), "w", cfg.nufft_width)
# seam test
rng = np.random.default_rng(1)
M = 4096
lo, hi = 100.0, 100.0 + M / 2.0
nodes = np.sort(rng.uniform(lo, hi, 3000))
coef = rng.standard_normal(3000)
xs = np.concatenate([[0.0], 10.0 ** np.linspace(-6, 0, 200)])
ref = o.direct_sums(nodes, coef, xs)
a = sk.panel_sums(cfg, nodes, coef, lo, hi, xs, method="direct")
b = sk.panel_sums(cfg, nodes, coef, lo, hi, xs, method="nufft")
nrm = np.abs(coef).sum()
print("direct vs oracle %.2e  nufft vs oracle %.2e" % (np.abs(a - ref).max() / nrm, np.abs(b - ref).max() / nrm))

for quadspec, n in (((256, 16), 1000), ((4096, 16), 1000), ((4096, 16), 10 ** 6)):
    t = time.time()
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5), quadspec=quadspec)
    print("config create %.2fs" % (time.time() - t))
    rs = 10.0 ** np.linspace(-6, 0, n)
    k0 = sk.compute_k0(cfg)
    for rep in range(3):
        t = time.time()
        ks, errs = sk.kernel_values(cfg, rs, k0=k0)
        dt = time.time() - t
    truth = np.pi / 2 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
    print(quadspec, n, "k0", k0, "time %.4fs" % dt, "max err/k0 %.3e" % (np.abs(ks - truth).max() / k0),
          "max est %.3e" % (np.nanmax(errs) / k0))
    print(cfg.last_stats())
    for r in cfg.last_trace():
        print("  ", r)
    if n <= 1000:
        ocfg = o.AdaptiveKernelConfig(o.SDF.matern(1, 1, 1.5), quadspec=quadspec)
        oks, oerrs, otr = o.kernel_values(ocfg, rs, k0=k0, return_trace=True)
        print("  vs oracle: %.3e (errs %.3e)" % (np.abs(ks - oks).max() / k0, np.abs(errs - oerrs).max() / k0))
        for r in otr:
            print("   o", r)
