"""The configuration of tools/diag/repro_seed.py 27 48 (round-2 stress sweep): the one case where HIP and the oracle
ended 2.4e-7 K(0) apart -- shared by the CPU-tier sensitivity test and the GPU parity test."""
import numpy as np


def seed27_case(oracle):
    seed, offset = 27, 48
    rng = np.random.default_rng(5000 + seed + offset)
    phi, rho, nu = rng.uniform(0.5, 3.0), 10.0 ** rng.uniform(-1.0, 0.7), rng.uniform(0.4, 2.2)
    n = int(rng.integers(30, 500))
    scale = 10.0 ** rng.uniform(-2, 0.7)
    rs = np.sort(scale * rng.uniform(0, 1, n))
    kw = dict(tol=1e-10, quadspec=(128, 8), derivative=True)
    base = oracle.AdaptiveKernelConfig(oracle.SDF.matern(phi, rho, nu, 1), tol=1e-10, quadspec=(128, 8))
    k0 = oracle.compute_k0(base)
    mk = lambda mod, **extra: mod.AdaptiveKernelConfig(mod.SDF.matern(phi, rho, nu, 1), **kw, **extra)
    truth = oracle.matern_cov_deriv(rs, (phi, rho, nu), d=1)
    return rs, k0, mk, truth
