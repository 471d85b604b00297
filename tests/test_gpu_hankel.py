"""dim >= 2 through the Hankel path by boxes (csrc/sk_hankel_inner.inc): one asymptotic-series NUFFT per panel, a directly
summed near box, the inner box by Chebyshev moments -- against the oracle's direct Bessel sums (the reference's own slow
branch, src/quadrature.jl:145-160), against the level-split transform it replaces (SK_HK_LEVELS=1 at config creation)
and against the device's direct kernel.  All three follow the same panel schedule, so the comparisons are at the
NUFFT's accuracy, far below the user tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PARMS = (0.7, 1.3, 1.8)


def three_ways(sk, monkeypatch, sdf, rs, k0, strict=True, **kw):
    """(boxes, level split, direct sums) on the same lags; boxes and direct sums must follow the same schedule."""
    monkeypatch.delenv("SK_HK_LEVELS", raising=False)
    boxes = sk.AdaptiveKernelConfig(sdf, hankel_min_lags=1, **kw)
    monkeypatch.setenv("SK_HK_LEVELS", "1")
    levels = sk.AdaptiveKernelConfig(sdf, hankel_min_lags=1, **kw)
    monkeypatch.delenv("SK_HK_LEVELS", raising=False)
    direct = sk.AdaptiveKernelConfig(sdf, hankel_min_lags=-1, **kw)
    out = []
    for cfg in (boxes, levels, direct):
        ks, es = sk.kernel_values(cfg, rs, k0=k0)
        out.append((ks, es, cfg.last_stats()["n_panels"], cfg.last_stats()["n_outer"]))
    assert not strict or out[0][2:] == out[2][2:]
    return out


@pytest.mark.parametrize("name,kw,rs", [
    # default quadspec (M = 65536): near box + inner box on the first panel, no near box afterwards
    ("log-spaced, default rule", dict(dim=2, tol=1e-8), 10.0 ** np.linspace(-6, 0.1, 5000)),
    ("derivative (J_1)", dict(dim=2, tol=1e-8, derivative=True), 10.0 ** np.linspace(-5, 0.1, 5000)),
    # many panels: the later ones start beyond om0
    ("uniform lags, tol 1e-12, (256,16)", dict(dim=2, tol=1e-12, quadspec=(256, 16)), np.linspace(0.0, 3.0, 6001)[1:]),
    # small rule: panels without any far field (2 pi pb r_hi <= 192), bisection
    ("tiny rule (16,2)", dict(dim=2, tol=1e-8, quadspec=(16, 2)), np.sort(np.random.default_rng(3).uniform(0.0, 1.0, 4000))),
    ("rule (64,6): three bands", dict(dim=2, tol=1e-9, quadspec=(64, 6)), 10.0 ** np.linspace(-4, 0.3, 4000)),
    # singular SDF: Gauss-Jacobi first panel
    ("alpha 1.5", dict(dim=2, tol=1e-8, alpha=1.5), np.linspace(0.0, 1.1, 5000)[1:]),
])
def test_boxes_vs_level_split_vs_direct_sums(sk, monkeypatch, name, kw, rs):
    sdf = sk.SDF.matern(*PARMS, 2)
    k0 = sk.compute_k0(sk.AdaptiveKernelConfig(sdf, **kw))
    (kb, eb, *sb), (kl, el, *sl), (kd, ed, *_) = three_ways(sk, monkeypatch, sdf, rs, k0, **kw)
    assert sb == sl
    a0 = abs(k0)
    assert np.array_equal(np.isnan(eb), np.isnan(ed))
    # the direct sums are exact to rounding: the box path must be as close to them as the level split is (1e-13 class)
    assert np.nanmax(np.abs(kb - kd)) / a0 <= 5e-13, name
    assert np.nanmax(np.abs(eb - ed)) / a0 <= 5e-13, name
    assert np.nanmax(np.abs(kb - kl)) / a0 <= 5e-13, name


@pytest.mark.parametrize("dim", [4, 6])
def test_boxes_fold_the_radial_factor(sk, oracle, monkeypatch, dim):
    # dim > 2: K = c r^-(dim/2-1) sum_j c_j J_nu(2 pi w_j r).  The boxes interpolate J_nu(z) / (z/2)^e with the factor w^e in
    # the strengths, so that lags of 1e-5 keep the accuracy of the sums (dividing an interpolant of absolute accuracy by
    # r^e afterwards lost 1e-12 at r = 1e-4 in dim 4).  dim 6 needs J_2 accurate in relative terms at small arguments
    # (bessel_j: ascending series below 2; the library's forward recurrence made the direct kernel bisect its way
    # through 22 panels of rounding noise here, where the exact integrand takes 4).
    parms = (1.3, 0.8, 1.2)
    kw = dict(dim=dim, tol=1e-8, quadspec=(256, 16))
    sdf = sk.SDF.matern(*parms, dim)
    k0 = float(oracle.matern_cov(0.0, parms, d=dim))
    rs = np.concatenate([10.0 ** np.linspace(-5, -1, 2000), np.linspace(0.1, 2.0, 2000)[1:]])
    (kb, eb, pb_, ob_), _, (kd, ed, *_) = three_ways(sk, monkeypatch, sdf, rs, k0, **kw)
    assert pb_ == ob_
    assert np.abs(kb - kd).max() / k0 <= 5e-13
    assert np.abs(kb - oracle.matern_cov(rs, parms, d=dim)).max() / k0 <= 10 * 1e-8


def test_boxes_against_the_oracle_on_pairwise_lags(sk, oracle):
    # BASELINE config 4 in small: unsorted pairwise distances with duplicates, default rule; the oracle's direct Bessel
    # sums on the same lag set (same r_hi, hence the same panels)
    rng = np.random.default_rng(11)
    X, Y = rng.uniform(0, 1, (60, 2)), rng.uniform(0, 1, (50, 2))
    rs = np.sqrt(((X[:, None, :] - Y[None, :, :]) ** 2).sum(-1)).ravel()
    rs = np.concatenate([rs, rs[:100]])
    kw = dict(dim=2, tol=1e-8, quadspec=(256, 16))
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*PARMS, 2), hankel_min_lags=1, **kw)
    ocfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*PARMS, 2), nthreads=8, **kw)
    k0 = float(oracle.matern_cov(0.0, PARMS, d=2))
    ks, es = sk.kernel_values(cfg, rs, k0=k0)
    oks, oes = oracle.kernel_values(ocfg, rs, k0=k0)
    assert np.abs(ks - oks).max() / k0 <= 5e-13 and np.abs(es - oes).max() / k0 <= 5e-13
    assert np.abs(ks - oracle.matern_cov(rs, PARMS, d=2)).max() / k0 <= 10 * 1e-8


def test_boxes_are_deterministic_and_handle_callbacks(sk, monkeypatch):
    # two calls give the same bits (fixed summation orders, no floating-point atomics); a closure SDF takes the
    # host-stepped march through the same kernels
    rs = 10.0 ** np.linspace(-5, 0.0, 4000)
    kw = dict(dim=2, tol=1e-8, quadspec=(256, 16), hankel_min_lags=1)
    cat = sk.AdaptiveKernelConfig(sk.SDF.matern(*PARMS, 2), **kw)
    k0 = sk.compute_k0(cat)
    a = sk.kernel_values(cat, rs, k0=k0)
    b = sk.kernel_values(cat, rs, k0=k0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    phi, rho, nu = PARMS
    cb = sk.AdaptiveKernelConfig(lambda w: phi * (rho * rho + w * w) ** (-nu - 1.0), sdf="callback", **kw)
    c = sk.kernel_values(cb, rs, k0=k0)
    assert np.abs(c[0] - a[0]).max() / abs(k0) <= 5e-13
    assert cb.last_stats()["n_panels"] == cat.last_stats()["n_panels"]


@pytest.mark.parametrize("seed", range(24))
def test_random_configs_boxes_vs_direct_sums(sk, seed):
    # tools/diag/hankel_stress.py: random rule, tolerance, dimension (2, 4, 6), order, singularity, SDF and lag distribution
    # (log-spaced / uniform / clustered): same panel count as the direct Bessel sums, values to 1e-3 of the tolerance
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "diag"))
    from hankel_stress import run_seed
    desc, ok, detail = run_seed(sk, seed)
    assert ok, desc + ": " + detail


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 40, 300])
def test_default_cutover_and_tiny_lag_sets(sk, n):
    # hankel_min_lags = 0: eight lags and more take the transform by boxes (the direct sums lose at every size:
    # tools/diag/hankel_cutoff.py); same values as the direct kernel either side of the cut-over, also for one or two lags
    kw = dict(dim=2, tol=1e-8, quadspec=(256, 16))
    sdf = sk.SDF.matern(*PARMS, 2)
    rs = np.sort(np.random.default_rng(n).uniform(0.01, 1.2, n))
    auto = sk.AdaptiveKernelConfig(sdf, **kw)
    direct = sk.AdaptiveKernelConfig(sdf, hankel_min_lags=-1, **kw)
    boxes = sk.AdaptiveKernelConfig(sdf, hankel_min_lags=1, **kw)
    k0 = sk.compute_k0(auto)
    ka, ea = sk.kernel_values(auto, rs, k0=k0)
    kd, ed = sk.kernel_values(direct, rs, k0=k0)
    kb, eb = sk.kernel_values(boxes, rs, k0=k0)
    assert np.abs(ka - kd).max() / k0 <= 5e-13 and np.abs(kb - kd).max() / k0 <= 5e-13
    assert auto.last_stats()["n_panels"] == direct.last_stats()["n_panels"] == boxes.last_stats()["n_panels"]
    # which path ran: the direct kernel reproduces itself bit for bit, the transform does not reproduce the direct sums
    same = np.array_equal(ka, kd)
    assert same == (n < 8) or np.array_equal(kb, kd)
