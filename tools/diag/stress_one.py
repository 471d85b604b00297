"""Details of one stress seed: python tools/diag/stress_one.py SEED"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralkernels_jl_amd as sk
from oracle import sk_oracle as oracle
seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
dim = [1, 1, 2][seed % 3]
phi, rho, nu = rng.uniform(0.5, 3.0), 10.0 ** rng.uniform(-1.0, 0.7), rng.uniform(0.4, 2.2)
mode = ["plain", "deriv", "logw", "plain"][(seed // 3) % 4]
alpha = 0.0
if mode == "logw" or seed % 5 == 0:
    alpha = (dim - 1) + rng.uniform(0.1, 0.8)
if mode == "deriv" and alpha >= dim - 0.05:
    alpha = 0.0
tol = [1e-4, 1e-6, 1e-8, 1e-10][seed % 4]
quadspec = [(256, 16), (128, 8), (64, 4), (512, 4)][(seed // 2) % 4]
n = int(rng.integers(30, 500)) if dim == 1 or seed % 2 else int(rng.integers(2100, 4000))
scale = 10.0 ** rng.uniform(-2, 0.7)
rs = scale * (rng.uniform(0, 1, n) if seed % 2 else 10.0 ** rng.uniform(-3, 0, n))
if seed % 4 == 0:
    rs = np.concatenate([rs, rs[: n // 5], [0.0]]); rs = rs[rng.permutation(rs.size)]
else:
    rs = np.sort(rs)
kw = dict(dim=dim, alpha=alpha, tol=tol, quadspec=quadspec, derivative=(mode == "deriv"), logw=(mode == "logw"))
base = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu, dim), dim=dim, alpha=alpha, tol=tol, quadspec=quadspec)
k0 = oracle.compute_k0(base)
print("params", phi, rho, nu, kw, "k0", k0, "n", rs.size, "r range", rs.min(), rs.max())
for w in (0, 16):
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(phi, rho, nu, dim), nufft_width=w, **kw)
    ocfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu, dim), **kw)
    ks, errs = sk.kernel_values(cfg, rs, k0=k0, param_derivative=(mode == "logw"))
    oks, oerrs, otr = oracle.kernel_values(ocfg, rs, k0=k0, param_derivative=(mode == "logw"), return_trace=True)
    tr = cfg.last_trace()
    d = np.abs(ks - oks) / abs(k0)
    print("width", w or "auto", "max diff", d.max(), "at r", rs[d.argmax()], "panels hip/oracle", len(tr), len(otr), "max|ks|/k0", np.abs(oks).max() / abs(k0))
    for i, (a, b) in enumerate(zip(tr, otr)):
        if (a["a"], a["b"], a["accepted"], a["n_active"]) != (b["a"], b["b"], b["accepted"], b["n_active"]):
            print("  first divergence at panel", i, a, b)
            break
    else:
        print("  common prefix identical; tails:", tr[len(otr):][:3] if len(tr) > len(otr) else otr[len(tr):][:3])
