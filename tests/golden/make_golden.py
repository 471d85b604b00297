"""Generates the committed fixtures under tests/golden/ (run in the build container).

Sources of truth: closed-form Fourier pairs the reference's tests use (scripts/matern_pair.jl,
test/exponential_sdf_1d.jl), mpmath for the singular Matern 1F2 formula, scipy + long-double
Newton for Gauss rules, and the CPU oracle (oracle/sk_oracle.c) for values/errs/panel schedule.
The reference itself (Julia + FINUFFT) cannot run here; no reference source is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import sk_oracle as o  # noqa: E402


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name), **kw)
    print("wrote", name)


def rules():
    for m, p in [(256, 0.0), (256, -0.4), (256, -0.5), (256, 1.0), (4096, 0.0), (4096, -0.4)]:
        o._RULE_CACHE.pop((m, p), None)
        path = os.path.join(HERE, "rules_m%d_p%+.3f.npz" % (m, p))
        if os.path.exists(path):
            os.remove(path)
        no1, wt1, no2, wt2 = o.gauss_rules(m, p)
        save("rules_m%d_p%+.3f.npz" % (m, p), no1=no1, wt1=wt1, no2=no2, wt2=wt2)


def trace_arr(tr):
    return np.array([[r["a"], r["b"], r["max_err"], r["n_active"], float(r["accepted"])] for r in tr])


def case(name, sdf, rs, truth, k0, **kw):
    cfg = o.AdaptiveKernelConfig(sdf, **kw)
    ks, errs, tr = o.kernel_values(cfg, rs, k0=k0, return_trace=True)
    print(name, "panels", len(tr), "max|oracle-truth|/k0 = %.2e" % (np.nanmax(np.abs(ks - truth)) / abs(k0)))
    save(name, rs=rs, truth=truth, k0=k0, ks=ks, errs=errs, trace=trace_arr(tr))


def main():
    rules()
    # C1: README demo (README.md:21-33): Matern nu=3/2, 1000 log-spaced r, tol=1e-8, default quadspec
    rs = 10.0 ** np.linspace(-6, 0, 1000)
    k0 = np.pi / 2
    case("c1_matern32_default.npz", o.SDF.matern(1, 1, 1.5), rs, k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs), k0)
    # C2 thinned: same, small rule (fast oracle), includes r = 0 and duplicates, unsorted
    rng = np.random.default_rng(7)
    rs2 = np.concatenate([[0.0], 10.0 ** np.linspace(-6, 0, 300), [1e-3, 1e-3, 0.0]])
    rs2 = rs2[rng.permutation(rs2.size)]
    case("c2_matern32_small_rule_unsorted.npz", o.SDF.matern(1, 1, 1.5), rs2,
         k0 * (1 + 2 * np.pi * rs2) * np.exp(-2 * np.pi * rs2), k0, quadspec=(256, 16))
    # C3 thinned: singular long-memory SDF alpha=0.4, rho=10, nu=0.5, K(0)=1, (2^8,2^4)
    alpha = 0.4
    kk = o.sing_matern_k0(alpha, (1.0, 10.0, 0.5))
    phi = 1.0 / kk
    rs3 = 10.0 ** np.linspace(-8, -2, 400)
    truth3 = o.sing_matern_cov(rs3, -alpha, (phi, 10.0, 0.5))
    case("c3_singular_alpha04.npz", o.SDF.matern(phi, 10.0, 0.5), rs3, truth3, 1.0, alpha=alpha, quadspec=(256, 16))
    # C5 thinned: Matern (2.14,0.97,0.89), tol=1e-12, (2^8,2^4), r in [0,5.1] incl. 0
    parms = (2.14, 0.97, 0.89)
    rs5 = np.linspace(0.0, 5.1, 500)
    k05 = float(o.matern_cov(0.0, parms))
    case("c5_matern_tol1e-12.npz", o.SDF.matern(*parms), rs5, o.matern_cov(rs5, parms), k05, tol=1e-12,
         quadspec=(256, 16))
    # derivative=true on the exponential pair (test/exponential_sdf_1d.jl)
    rs6 = np.linspace(0.0, 5.1, 400)[1:]
    case("exp_derivative.npz", o.SDF.exponential(1.0), rs6, o.exponential_cov_deriv(rs6), 2.0, derivative=True,
         quadspec=(256, 16))
    # a narrow spectral peak that forces bisection of the first panel (src/quadrature.jl:268-270)
    rs7 = 10.0 ** np.linspace(-4, 0, 200)
    prm = (1.0, 1e-3, 1.5)
    k07 = float(o.matern_cov(0.0, prm))
    case("bisect_narrow_matern.npz", o.SDF.matern(*prm), rs7, o.matern_cov(rs7, prm), k07, quadspec=(64, 4))


if __name__ == "__main__":
    main()
