"""Seeded randomised parity: random Matern parameters, singularity, tolerance, quadspec and lag sets
(sorted / unsorted / duplicated / with zero) against the oracle through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(8))
def test_random_configs_vs_oracle(sk, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    phi, rho, nu = rng.uniform(0.5, 3.0), 10.0 ** rng.uniform(-1.5, 1.0), rng.uniform(0.3, 2.5)
    alpha = [0.0, 0.0, rng.uniform(0.05, 0.8)][seed % 3]
    tol = [1e-4, 1e-6, 1e-8, 1e-10][seed % 4]
    quadspec = [(256, 16), (128, 8), (64, 4), (512, 4)][(seed // 2) % 4]
    n = int(rng.integers(20, 400))
    scale = 10.0 ** rng.uniform(-3, 1)
    rs = scale * (rng.uniform(0, 1, n) if seed % 2 else 10.0 ** rng.uniform(-4, 0, n))
    if seed % 3 == 0:
        rs = np.concatenate([rs, rs[: n // 5], [0.0]])
        rs = rs[rng.permutation(rs.size)]
    else:
        rs = np.sort(rs)
    kw = dict(alpha=alpha, tol=tol, quadspec=quadspec)
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(phi, rho, nu), **kw)
    ocfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu), **kw)
    k0 = oracle.sing_matern_k0(alpha, (phi, rho, nu)) if alpha else float(oracle.matern_cov(0.0, (phi, rho, nu)))
    ks, errs = sk.kernel_values(cfg, rs, k0=k0)
    oks, oerrs, otr = oracle.kernel_values(ocfg, rs, k0=k0, return_trace=True)
    assert np.array_equal(np.isnan(errs), np.isnan(oerrs))
    assert np.nanmax(np.abs(ks - oks)) / k0 <= max(0.1 * tol, 5e-13)
    # near-threshold accept/convergence decisions may differ by NUFFT noise: allow a different
    # schedule only if the values still agree (checked above); usually it is identical
    tr = cfg.last_trace()
    if len(tr) == len(otr):
        assert [(r["a"], r["b"]) for r in tr] == [(r["a"], r["b"]) for r in otr]
