"""Unsorted lag arrays (src/adaptive.jl:99-120: unique + sort, evaluate, gather back): the device bucket sort (csrc/
sk_kernels.hip, k_bk_*) and its fall-back to the library sort must give, bit for bit, what the march gives on the sorted
unique lags."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def check(sk, rs, **kw):
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1.0, 1.0, 1.5), quadspec=(64, 4), tol=1e-7, **kw)
    k0 = sk.compute_k0(cfg)
    ux, inv = np.unique(np.where(rs == 0.0, 0.0, rs), return_inverse=True)
    ks_u, es_u = sk.kernel_values(cfg, ux, k0=k0)
    ks, es = sk.kernel_values(cfg, rs, k0=k0)
    assert cfg.last_stats()["n_unique"] == ux.size
    assert np.array_equal(ks, ks_u[inv]) and np.array_equal(es, es_u[inv], equal_nan=True)
    # device-resident entry too
    n = rs.size
    d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
    sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    sk.synchronize()
    assert np.array_equal(d_ks.to_host(), ks)


@pytest.mark.parametrize("name,make", [
    ("uniform, 10 % duplicates", lambda g: np.r_[g.uniform(0, 2, 180000), g.uniform(0, 2, 20000).repeat(2)[:20000]]),
    ("pairwise distances", lambda g: np.sqrt(((g.uniform(0, 1, (500, 1, 2)) - g.uniform(0, 1, (1, 400, 2))) ** 2).sum(-1)).ravel()),
    ("50 distinct values (buckets overflow: library sort)", lambda g: g.choice(g.uniform(0, 1, 50), 100000)),
    ("all equal", lambda g: np.full(70000, 0.37)),
    ("exponential: mass at the bottom of the range", lambda g: g.exponential(0.01, 150000)),
    ("zeros of both signs and one huge lag", lambda g: np.r_[g.uniform(0, 1, 60000), [0.0, -0.0, 0.0, 50.0]]),
    ("below the bucket-sort size", lambda g: g.uniform(0, 1, 5000).repeat(2)),
])
def test_unsorted_lags_match_sorted_unique_bit_for_bit(sk, name, make):
    g = np.random.default_rng(12)
    rs = make(g)
    g.shuffle(rs)
    check(sk, rs)


def test_unsorted_invalid_lags_are_rejected(sk):
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1.0, 1.0, 1.5), quadspec=(64, 4))
    rs = np.random.default_rng(1).uniform(0, 1, 50000)
    for bad in (-0.25, float("nan")):
        x = rs.copy()
        x[1234] = bad
        with pytest.raises(sk.SpectralKernelsError):
            sk.kernel_values(cfg, x, k0=1.0)


def test_device_resident_unsorted_lags_are_sorted_without_asking_from_the_second_call_on(sk):
    # the first call of a config finds out (check + speculative first step); the next ones with as many lags sort at once,
    # and the sort reports what the check would have: an invalid lag, or an input that was sorted after all
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1.0, 1.0, 1.5), quadspec=(64, 4), tol=1e-7)
    k0 = sk.compute_k0(cfg)
    g = np.random.default_rng(5)
    n = 60000
    ref_cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1.0, 1.0, 1.5), quadspec=(64, 4), tol=1e-7)

    def run(rs):
        d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
        sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
        sk.synchronize()
        return d_ks.to_host(), d_es.to_host(), cfg.last_stats()

    def expect(rs):
        ux, inv = np.unique(rs, return_inverse=True)
        ks, es = sk.kernel_values(ref_cfg, ux, k0=k0)
        return ks[inv], es[inv]

    syncs = []
    for trial in range(3):  # unsorted three times: the second and third take the sort-first path
        rs = g.uniform(0, 2, n)
        ks, es, st = run(rs)
        eks, ees = expect(rs)
        assert np.array_equal(ks, eks) and np.array_equal(es, ees, equal_nan=True)
        syncs.append(st["n_syncs"])
    srt = np.sort(g.uniform(0, 2, n))           # sorted after all: still right, and the expectation ends
    ks, es, st = run(srt)
    eks, ees = expect(srt)
    assert np.array_equal(ks, eks)
    ks2, _, st2 = run(srt)                      # ... so this one is marched as it is
    assert np.array_equal(ks2, eks) and st2["n_unique"] == n
    rs = g.uniform(0, 2, n)
    run(rs)                                     # unsorted again: found out by the check
    bad = g.uniform(0, 2, n)
    bad[777] = -1.0
    with pytest.raises(sk.SpectralKernelsError):  # sort-first path: the sort reports the invalid lag
        run(bad)
    ks, es, _ = run(rs)                         # and the config is still usable
    eks, _ = expect(rs)
    assert np.array_equal(ks, eks)
