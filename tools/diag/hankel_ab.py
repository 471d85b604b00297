"""dim >= 2: the Hankel path with the inner region by moments against the level-split transform (SK_HK_LEVELS=1),
same box: values, errors against the closed form, ms per call, event-timed launch kinds.
usage: hankel_ab.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
import bench
from oracle import sk_oracle as o

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def mk(levels, sdf, **kw):
    if levels:
        os.environ["SK_HK_LEVELS"] = "1"
    else:
        os.environ.pop("SK_HK_LEVELS", None)
    return sk.AdaptiveKernelConfig(sdf, **kw)


def run(cfg, rs, k0):
    n = rs.size
    d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
    step = lambda: sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    for _ in range(3):
        step()
    sk.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        step()
    sk.synchronize()
    dt = (time.perf_counter() - t) / steps
    ks, es = d_ks.to_host(), d_es.to_host()
    st = cfg.last_stats()
    cfg.profile(True)
    acc = {}
    for _ in range(5):
        step()
        for k, v in cfg.kernel_times().items():
            a = acc.setdefault(k, [0.0, 0])
            a[0] += v["total_ms"]; a[1] += v["launches"]
    cfg.profile(False)
    ev = {k: (round(1e3 * a[0] / 5, 1), a[1] // 5) for k, a in acc.items() if a[1]}
    return ks, es, 1e3 * dt, st, ev


cases = []
X = bench.shard_grid(bench.WORKLOADS["C4"], 0, 1)
prm = (0.7, 1.3, 1.8, 2.0)
cases.append(("C4 pairwise 1e6", sk.SDF.matern(*prm), dict(tol=1e-8, dim=2), X, lambda r: o.matern_cov(r, prm[:3], d=2)))
lg = 10.0 ** np.linspace(-6, 0, 10 ** 6)
cases.append(("log-spaced 1e6", sk.SDF.matern(*prm), dict(tol=1e-8, dim=2), lg, lambda r: o.matern_cov(r, prm[:3], d=2)))
un = np.linspace(0.0, 3.0, 200001)[1:]
cases.append(("uniform 2e5 tol 1e-12 (256,16)", sk.SDF.matern(*prm), dict(tol=1e-12, dim=2, quadspec=(256, 16)), un,
              lambda r: o.matern_cov(r, prm[:3], d=2)))
cases.append(("derivative log 3e5", sk.SDF.matern(*prm), dict(tol=1e-8, dim=2, derivative=True), 10.0 ** np.linspace(-5, 0, 300000), None))
cases.append(("dim 4 log 3e5", sk.SDF.matern(0.7, 1.3, 1.8, 4.0), dict(tol=1e-8, dim=4), 10.0 ** np.linspace(-4, 0, 300000),
              lambda r: o.matern_cov(r, (0.7, 1.3, 1.8), d=4)))
cases.append(("exponential sdf log 3e5", sk.SDF.exponential(0.8), dict(tol=1e-9, dim=2), 10.0 ** np.linspace(-5, 0.5, 300000), None))
for name, sdf, kw, rs, truth in cases:
    res = {}
    for levels in (1, 0):
        cfg = mk(levels, sdf, **kw)
        k0 = sk.compute_k0(cfg)
        ks, es, ms, st, ev = run(cfg, rs, k0)
        res[levels] = (ks, es)
        err = float(np.abs(ks - truth(rs)).max() / abs(k0)) if truth else None
        print("%-34s %s: %.4f ms  err %s  panels %d  syncs %d  events(us/call, launches) %s" % (
            name, "levels " if levels else "moments", ms, err, st["n_panels"], st["n_syncs"], ev), flush=True)
    k0 = abs(k0)
    print("    moments vs levels: max |dK| / K0 %.3e   max |derr| / K0 %.3e   nan %d / %d" % (
        np.nanmax(np.abs(res[0][0] - res[1][0])) / k0, np.nanmax(np.abs(res[0][1] - res[1][1])) / k0,
        int(np.isnan(res[0][0]).sum()), int(np.isnan(res[1][0]).sum())), flush=True)
