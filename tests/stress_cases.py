"""Configurations of tools/diag/stress.py (seed, offset) in the one regime where HIP and the oracle end up to 1e-6 K(0) apart:
derivative=true at tol 1e-10 (the algorithm as written stops ~1000 tol short there) -- shared by the CPU-tier
sensitivity test and the GPU parity tests.  (27, 48) is the round-2 stress sweep's mismatch; (27, 0) and (39, 0) showed
up in round 3's sweep."""
import numpy as np


def stress_case(oracle, seed, offset):
    rng = np.random.default_rng(5000 + seed + offset)
    dim = [1, 1, 2][seed % 3]
    phi, rho, nu = rng.uniform(0.5, 3.0), 10.0 ** rng.uniform(-1.0, 0.7), rng.uniform(0.4, 2.2)
    mode = ["plain", "deriv", "logw", "plain"][(seed // 3) % 4]
    assert dim == 1 and mode == "deriv" and seed % 5 != 0
    tol = [1e-4, 1e-6, 1e-8, 1e-10][seed % 4]
    quadspec = [(256, 16), (128, 8), (64, 4), (512, 4)][(seed // 2) % 4]
    n = int(rng.integers(30, 500))
    scale = 10.0 ** rng.uniform(-2, 0.7)
    rs = scale * (rng.uniform(0, 1, n) if seed % 2 else 10.0 ** rng.uniform(-3, 0, n))
    if seed % 4 == 0:
        rs = np.concatenate([rs, rs[: n // 5], [0.0]])
        rs = rs[rng.permutation(rs.size)]
    else:
        rs = np.sort(rs)
    kw = dict(tol=tol, quadspec=quadspec, derivative=True)
    base = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu, 1), tol=tol, quadspec=quadspec)
    k0 = oracle.compute_k0(base)
    mk = lambda mod, **extra: mod.AdaptiveKernelConfig(mod.SDF.matern(phi, rho, nu, 1), **kw, **extra)
    truth = oracle.matern_cov_deriv(rs, (phi, rho, nu), d=1)
    return rs, k0, mk, truth


def seed27_case(oracle):
    return stress_case(oracle, 27, 48)
