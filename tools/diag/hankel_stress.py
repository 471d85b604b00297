"""Random dim >= 2 configurations: the Hankel path by boxes against the device's direct Bessel sums (hankel_min_lags=-1),
same lags.  Prints every seed; exit code 1 on a mismatch.  usage: hankel_stress.py [first seed] [count] [fewest lags] [most lags]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run_seed(sk, seed, nlo=2100, nhi=12000):
    """(description, ok, detail) of one random configuration."""
    g = np.random.default_rng(seed)
    dim = int(g.choice([2, 2, 2, 4, 6]))
    m = int(g.choice([16, 32, 64, 128, 256, 512]))
    k = int(g.choice([1, 2, 3, 4, 6, 8, 16]))
    tol = float(10.0 ** g.integers(-12, -5))
    derivative = bool(g.random() < 0.25)
    alpha = float(g.choice([0.0, 0.0, 0.5, 1.5])) if dim == 2 else 0.0
    prm = (float(g.uniform(0.3, 2.0)), float(10.0 ** g.uniform(-1.5, 0.7)), float(g.uniform(0.4, 2.5)))
    n = int(g.integers(nlo, nhi))
    top = float(10.0 ** g.uniform(-2, 1))
    kind = int(g.integers(0, 3))
    if kind == 0:
        rs = top * 10.0 ** np.sort(g.uniform(-6, 0, n))
    elif kind == 1:
        rs = np.sort(g.uniform(0, top, n))
    else:
        rs = np.sort(np.abs(top * (0.3 + 0.02 * g.standard_normal(n))))
    rs = np.unique(rs)
    kw = dict(dim=dim, tol=tol, quadspec=(m, k), derivative=derivative, alpha=alpha)
    desc = "seed %d dim %d (%d,%d) tol %.0e der %d alpha %.1f rho %.3g nu %.2f n %d top %.3g kind %d" % (
        seed, dim, m, k, tol, derivative, alpha, prm[1], prm[2], rs.size, top, kind)
    try:
        sdf = sk.SDF.matern(*prm, dim)
        boxes = sk.AdaptiveKernelConfig(sdf, hankel_min_lags=1, **kw)
        direct = sk.AdaptiveKernelConfig(sdf, hankel_min_lags=-1, **kw)
        k0 = sk.compute_k0(sk.AdaptiveKernelConfig(sdf, dim=dim, tol=tol, quadspec=(m, k), alpha=alpha))
    except Exception as e:  # rejected at creation (both would be)
        return desc, True, str(e)[:80]
    res = []
    for cfg in (boxes, direct):
        try:
            ks, es = sk.kernel_values(cfg, rs, k0=k0)
            res.append((ks, es, cfg.last_stats()["n_panels"], None))
        except sk.SpectralKernelsError as e:
            res.append((None, None, -1, e.code))
    (kb, eb, pb, cb), (kd, ed, pd, cd) = res
    if cb is not None or cd is not None:
        return desc, cb == cd, "errors %s / %s" % (cb, cd)
    # (dim > 2: compute_k0, like the reference's, integrates w^(dim/2) S without the (pi w)^nu / nu! of the small-r limit and
    # can be orders below K(0); the values themselves set the scale then)
    scale = max(abs(k0), float(np.nanmax(np.abs(kd))))
    d = float(np.nanmax(np.abs(kb - kd)) / scale)
    de = float(np.nanmax(np.abs(eb - ed)) / scale)
    ok = pb == pd and d <= max(1e-3 * tol, 2e-12) and np.array_equal(np.isnan(eb), np.isnan(ed))
    return desc, bool(ok), "panels %d / %d  dK %.2e  derr %.2e" % (pb, pd, d, de)


if __name__ == "__main__":
    import spectralkernels_jl_amd as sk
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    nlo = int(sys.argv[3]) if len(sys.argv) > 3 else 2100
    nhi = int(sys.argv[4]) if len(sys.argv) > 4 else 12000
    bad = 0
    for seed in range(s0, s0 + cnt):
        desc, ok, detail = run_seed(sk, seed, nlo, nhi)
        print("%s: %s %s" % (desc, detail, "ok" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
    print("mismatches: %d" % bad)
    sys.exit(1 if bad else 0)
