"""One-off randomised stress run against the oracle (wider than tests/test_gpu_random.py): dims 1 and 2,
derivative, logw, sorted / unsorted / duplicated lags.  python tools/diag/stress.py [n_seeds]"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "16")  # the oracle: 256 logical CPUs are visible on the GPU box, a fraction is ours
import functools
print = functools.partial(print, flush=True)  # gpurun takes seven silent minutes for a hang
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralkernels_jl_amd as sk
from oracle import sk_oracle as oracle

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
OFFSET = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for seed in range(nseeds):
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
    try:
        base = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu, dim), dim=dim, alpha=alpha, tol=tol, quadspec=quadspec)
        k0 = oracle.compute_k0(base)
        cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(phi, rho, nu, dim), **kw)
        ocfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu, dim), **kw)
        pd = mode == "logw"
        ks, errs = sk.kernel_values(cfg, rs, k0=k0, param_derivative=pd)
        oks, oerrs = oracle.kernel_values(ocfg, rs, k0=k0, param_derivative=pd)
        d = np.nanmax(np.abs(ks - oks)) / abs(k0)
        ok = np.array_equal(np.isnan(errs), np.isnan(oerrs)) and d <= max(0.2 * tol, 5e-12)
    except Exception as e:  # both sides should agree on errors too
        ok, d = False, repr(e)
    if not ok:
        bad += 1
    print("seed %d dim %d %s alpha %.2f tol %.0e qs %s n %d -> %s %s" % (seed, dim, mode, alpha, tol, quadspec, rs.size,
                                                                        "ok" if ok else "MISMATCH", d), flush=True)
print("done: %d seeds, %d mismatches, %.0f s" % (nseeds, bad, time.time() - t0))
sys.exit(1 if bad else 0)
