"""Joint multi-integrand march: sk_kernel_values_joint against the separate marches -- HIP's own and the
oracle's -- for the integrands of gen_kernel_jacobian (src/derivatives.jl:86-112): K, dK/dr (derivative=true),
dK/dalpha (logw=true) and dS/dtheta_j (kernel_sdf_derivatives, src/derivatives.jl:61-72)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _separate(sk, cfgs, rs, k0, pds):
    return [sk.kernel_values(c, rs, k0=k0, param_derivative=pd) for c, pd in zip(cfgs, pds)]


def test_catalogue_parameter_derivatives_on_the_device(sk, oracle):
    # dS/dtheta_j of the catalogue (sdf_dparam) as a device integrand == the same function through the host callback
    # and the oracle's march with that callback; test/derivatives/sdf_params.jl:8-24 parameters
    parms = (2.3, 0.1, 1.75)
    rs = np.linspace(0.0, 3.5, 30)
    f = sk.SDF.matern(*parms)
    k0 = sk.compute_k0(sk.AdaptiveKernelConfig(f, quadspec=(256, 16)))
    for j in range(3):
        dev = sk.AdaptiveKernelConfig(f.d(j), quadspec=(256, 16), tol=1e-10)
        cb = sk.AdaptiveKernelConfig(lambda w, j=j: f.d(j)(w), quadspec=(256, 16), tol=1e-10, sdf="callback")
        kd, ed = sk.kernel_values(dev, rs, k0=k0, param_derivative=True)
        kc, ec = sk.kernel_values(cb, rs, k0=k0, param_derivative=True)
        assert np.abs(kd - kc).max() <= 1e-11 * k0 and dev.last_stats()["n_panels"] == cb.last_stats()["n_panels"]
        ocfg = oracle.AdaptiveKernelConfig(lambda w, j=j: float(f.d(j)(w)), quadspec=(256, 16), tol=1e-10)
        ok, _ = oracle.kernel_values(ocfg, rs, k0=k0, param_derivative=True)
        assert np.abs(kd - ok).max() <= max(0.1 * 1e-10 * k0, 1e-12 * np.abs(ok).max())
        # and against d/dtheta_j of the closed-form Matern pair by central differences (sdf_params.jl:13-18 use AD)
        h = 1e-6 * max(1.0, abs(parms[j]))
        pp, pm = list(parms), list(parms)
        pp[j] += h
        pm[j] -= h
        fd = (oracle.matern_cov(rs, pp) - oracle.matern_cov(rs, pm)) / (2 * h)
        assert np.abs(kd - fd).max() <= 1e-5 * max(1.0, np.abs(fd).max())  # the reference's own bar (sdf_params.jl:22-24)


@pytest.mark.parametrize("alpha", [0.0, 0.5])
def test_joint_march_matches_separate_marches(sk, oracle, alpha):
    tol = 1e-8
    parms = (2.14, 0.97, 0.89)
    rng = np.random.default_rng(5)
    rs = np.sort(np.r_[0.0, rng.uniform(0.0, 1.1, 5000), 10.0 ** rng.uniform(-5, 0, 2000)])
    f = sk.SDF.matern(*parms)
    kw = dict(alpha=alpha, quadspec=(256, 16), tol=tol)
    cfgs = [sk.AdaptiveKernelConfig(f, **kw), sk.AdaptiveKernelConfig(f, derivative=True, **kw)]
    pds = [False, False]
    if alpha != 0.0:  # kernel_singularity_derivative (src/derivatives.jl:74-81)
        cfgs.append(sk.AdaptiveKernelConfig(f, logw=True, **kw))
        pds.append(True)
    for j in range(3):  # kernel_sdf_derivatives
        cfgs.append(sk.AdaptiveKernelConfig(f.d(j), **kw))
        pds.append(True)
    k0 = sk.compute_k0(cfgs[0])
    vj, ej = sk.kernel_values_joint(cfgs, rs, k0, param_derivative=pds)
    st = cfgs[0].last_stats()
    sep = _separate(sk, cfgs, rs, k0, pds)
    for t, (v, e) in enumerate(sep):
        scale = abs(k0)
        assert np.abs(vj[t] - v).max() <= tol * scale, (t, np.abs(vj[t] - v).max() / scale)
        assert np.isnan(ej[t][0]) and np.all(np.isfinite(ej[t][1:]))
        assert vj[t][0] == v[0]  # x == 0 (src/adaptive.jl:134-145)
    assert st["n_panels"] >= max(c.last_stats()["n_panels"] for c in cfgs[1:])
    # the oracle's separate marches (its own closures for the parameter derivatives)
    okw = dict(alpha=alpha, quadspec=(256, 16), tol=tol)
    of = oracle.SDF.matern(*parms)
    ocfgs = [oracle.AdaptiveKernelConfig(of, **okw), oracle.AdaptiveKernelConfig(of, derivative=True, **okw)]
    if alpha != 0.0:
        ocfgs.append(oracle.AdaptiveKernelConfig(of, logw=True, **okw))
    for j in range(3):
        ocfgs.append(oracle.AdaptiveKernelConfig(lambda w, j=j: float(f.d(j)(w)), **okw))
    sub = np.r_[0, np.arange(1, rs.size, 23), rs.size - 1]
    for t, (oc, pd) in enumerate(zip(ocfgs, pds)):
        ov, _ = oracle.kernel_values(oc, rs[sub], k0=k0, param_derivative=pd)
        # two marches with different schedules (the oracle's on a thinned lag set), each within tol-ish of the truth
        assert np.abs(vj[t][sub] - ov).max() <= 3 * tol * abs(k0), (t, np.abs(vj[t][sub] - ov).max() / abs(k0))
    # closed form of K itself (test/matern_sdf.jl acceptance)
    truth = oracle.matern_cov(rs, parms) if alpha == 0.0 else np.r_[k0, oracle.sing_matern_cov(rs[1:], -alpha, parms)]
    assert np.abs(vj[0] - truth).max() / abs(k0) <= 10 * tol


def test_joint_march_unsorted_duplicates_and_cost(sk):
    tol = 1e-8
    f = sk.SDF.matern(1.0, 1.0, 1.5)
    kw = dict(tol=tol)
    cfgs = [sk.AdaptiveKernelConfig(f, **kw), sk.AdaptiveKernelConfig(f, derivative=True, **kw),
            sk.AdaptiveKernelConfig(f.d(0), **kw), sk.AdaptiveKernelConfig(f.d(1), **kw), sk.AdaptiveKernelConfig(f.d(2), **kw)]
    pds = [False, False, True, True, True]
    n = 10 ** 6
    rs = 10.0 ** np.linspace(-6, 0, n)
    k0 = np.pi / 2
    vj, ej = sk.kernel_values_joint(cfgs, rs, k0, param_derivative=pds)  # warm-up (plans, buffers)
    t0 = time.perf_counter()
    vj, ej = sk.kernel_values_joint(cfgs, rs, k0, param_derivative=pds)
    t_joint = time.perf_counter() - t0
    sk.kernel_values_multi(cfgs, rs, [k0] * 5, param_derivative=pds)
    t0 = time.perf_counter()
    vm, em = sk.kernel_values_multi(cfgs, rs, [k0] * 5, param_derivative=pds)
    t_multi = time.perf_counter() - t0
    t0 = time.perf_counter()
    sk.kernel_values(cfgs[0], rs, k0=k0)
    t_one = time.perf_counter() - t0
    print("\\njoint %.2f ms, separate marches %.2f ms, one integrand %.2f ms (host arrays, 1e6 lags x 5 integrands)" % (
        1e3 * t_joint, 1e3 * t_multi, 1e3 * t_one))
    for t in range(5):
        assert np.abs(vj[t] - vm[t]).max() <= tol * k0
    truth = k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
    assert np.abs(vj[0] - truth).max() / k0 <= 10 * tol
    # any order, duplicates, a zero: one sort + dedup for all integrands
    rng = np.random.default_rng(9)
    sub = rng.permutation(n)[:20000]
    xs = np.r_[rs[sub], rs[sub[:500]], 0.0]
    rng.shuffle(xs)
    vu, eu = sk.kernel_values_joint(cfgs, xs, k0, param_derivative=pds)
    ref = sk.kernel_values_joint(cfgs, np.unique(xs), k0, param_derivative=pds)[0]
    idx = np.searchsorted(np.unique(xs), xs)
    assert np.array_equal(vu, ref[:, idx])
    assert cfgs[0].last_stats()["n_unique"] == np.unique(xs).size
