"""Re-run one configuration of tools/diag/stress.py (seed, offset) with timings of both sides."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "16")
import functools
print = functools.partial(print, flush=True)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sk_oracle as oracle
seed, OFFSET = int(sys.argv[1]), int(sys.argv[2])
side = sys.argv[3] if len(sys.argv) > 3 else "both"
rng = np.random.default_rng(5000 + seed + OFFSET)
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
    rs = np.concatenate([rs, rs[: n // 5], [0.0]])
    rs = rs[rng.permutation(rs.size)]
else:
    rs = np.sort(rs)
kw = dict(dim=dim, alpha=alpha, tol=tol, quadspec=quadspec, derivative=(mode == "deriv"), logw=(mode == "logw"))
print("config", dict(phi=phi, rho=rho, nu=nu), kw, "n", rs.size, "r in", rs.min(), rs.max())
base = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu, dim), dim=dim, alpha=alpha, tol=tol, quadspec=quadspec)
k0 = oracle.compute_k0(base)
pd = mode == "logw"
if side in ("both", "oracle"):
    ocfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu, dim), **kw)
    t = time.time()
    oks, oerrs = oracle.kernel_values(ocfg, rs, k0=k0, param_derivative=pd)
    print("oracle %.1f s" % (time.time() - t), getattr(ocfg, "last_stats", lambda: None)())
if side in ("both", "hip"):
    import spectralkernels_jl_amd as sk
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(phi, rho, nu, dim), **kw)
    t = time.time()
    ks, errs = sk.kernel_values(cfg, rs, k0=k0, param_derivative=pd)
    print("hip %.3f s" % (time.time() - t), cfg.last_stats())
    if side == "both":
        print("max diff / k0", np.nanmax(np.abs(ks - oks)) / abs(k0))
