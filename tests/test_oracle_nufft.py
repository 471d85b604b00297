"""The oracle's CPU type-3 NUFFT mode (used only for bench.py's cpu_baseline) agrees with its exact
direct sums, so the timed baseline computes the same thing as the checker."""
import numpy as np


def test_oracle_nufft_mode_matches_direct(oracle):
    rs = np.concatenate([[0.0], 10.0 ** np.linspace(-6, 0, 700)])
    k0 = np.pi / 2
    a = oracle.AdaptiveKernelConfig(oracle.SDF.matern(1, 1, 1.5), quadspec=(256, 16))
    b = oracle.AdaptiveKernelConfig(oracle.SDF.matern(1, 1, 1.5), quadspec=(256, 16), use_nufft=True)
    ka, ea, ta = oracle.kernel_values(a, rs, k0=k0, return_trace=True)
    kb, eb, tb = oracle.kernel_values(b, rs, k0=k0, return_trace=True)
    assert np.nanmax(np.abs(ka - kb)) / k0 < 1e-10
    assert [(r["a"], r["b"], r["accepted"]) for r in ta] == [(r["a"], r["b"], r["accepted"]) for r in tb]
    # singular + derivative variants share the code path
    parms = (2.14, 0.97, 0.89)
    for kw in (dict(alpha=0.4), dict(derivative=True)):
        a = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms), quadspec=(256, 16), **kw)
        b = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms), quadspec=(256, 16), use_nufft=True, **kw)
        xs = np.linspace(0.01, 1.1, 500)
        ka, _ = oracle.kernel_values(a, xs, k0=4.0)
        kb, _ = oracle.kernel_values(b, xs, k0=4.0)
        assert np.abs(ka - kb).max() / 4.0 < 1e-10
