"""The oracle reproduces the committed fixtures (tests/golden/, made by make_golden.py) and the
fixtures agree with the closed-form truths they carry."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "c2_matern32_small_rule_unsorted.npz": dict(sdf=("matern", (1, 1, 1.5)), kw=dict(quadspec=(256, 16)), tol=1e-8),
    "c3_singular_alpha04.npz": dict(sdf=("matern", None), kw=dict(alpha=0.4, quadspec=(256, 16)), tol=1e-8),
    "exp_derivative.npz": dict(sdf=("exponential", (1.0,)), kw=dict(derivative=True, quadspec=(256, 16)), tol=1e-8),
    "bisect_narrow_matern.npz": dict(sdf=("matern", (1.0, 1e-3, 1.5)), kw=dict(quadspec=(64, 4)), tol=1e-8),
}


def make_sdf(mod, spec, oracle):
    kind, prm = spec
    if kind == "matern":
        if prm is None:  # C3: phi normalised so that K(0) = 1
            phi = 1.0 / oracle.sing_matern_k0(0.4, (1.0, 10.0, 0.5))
            prm = (phi, 10.0, 0.5)
        return mod.SDF.matern(*prm)
    return mod.SDF.exponential(*prm)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_fixture(oracle, name):
    z = np.load(os.path.join(GOLD, name))
    c = CASES[name]
    cfg = oracle.AdaptiveKernelConfig(make_sdf(oracle, c["sdf"], oracle), tol=c["tol"], **c["kw"])
    ks, errs, tr = oracle.kernel_values(cfg, z["rs"], k0=float(z["k0"]), return_trace=True)
    k0 = abs(float(z["k0"]))
    assert np.nanmax(np.abs(ks - z["ks"])) / k0 < 1e-13
    assert np.array_equal(np.isnan(errs), np.isnan(z["errs"]))
    assert len(tr) == z["trace"].shape[0]
    assert [r["accepted"] for r in tr] == [bool(a) for a in z["trace"][:, 4]]


@pytest.mark.parametrize("name", sorted(glob.glob(os.path.join(GOLD, "c*.npz")) + glob.glob(os.path.join(GOLD, "e*.npz"))
                                        + glob.glob(os.path.join(GOLD, "b*.npz"))))
def test_fixture_within_10tol_of_truth(name):
    z = np.load(name)
    tol = 1e-12 if "1e-12" in name else 1e-8
    err = np.abs(z["ks"] - z["truth"]) / abs(float(z["k0"]))
    assert np.nanmax(err) <= 10 * tol  # acceptance of test/matern_sdf.jl:27


def test_rule_fixtures_properties():
    for f in glob.glob(os.path.join(GOLD, "rules_*.npz")):
        z = np.load(f)
        p = float(os.path.basename(f).split("_p")[1][:-4])
        for no, wt in ((z["no1"], z["wt1"]), (z["no2"], z["wt2"])):
            assert np.all(np.diff(no) > 0) and no[0] > -1 and no[-1] < 1
            assert abs(wt.sum() - 2 ** (p + 1) / (p + 1)) < 1e-13
            # exactness on monomials up to degree 5 against the Jacobi moments int (1+x)^p x^k
            from math import comb
            for k in range(6):  # t = 1 + x: int_0^2 t^p (t-1)^k dt
                exact = sum(comb(k, i) * (-1) ** (k - i) * 2 ** (p + i + 1) / (p + i + 1) for i in range(k + 1))
                assert abs((wt * no ** k).sum() - exact) < 1e-13
