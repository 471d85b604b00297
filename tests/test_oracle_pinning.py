"""Pins the CPU oracle (oracle/sk_oracle.c) against the reference's own known-answer tests:
closed-form Fourier pairs with the acceptance |K~ - K| / K(0) <= 10*tol
(test/exponential_sdf_1d.jl:3-21, test/matern_sdf.jl:4-27 and :36-63, test/derivatives/warping.jl:21-23).
The reference holds no bit-level golden vectors for this path."""
import numpy as np
import pytest

TOLS = [1e-4, 1e-8, 1e-10, 1e-12]


@pytest.mark.parametrize("derivative", [False, True])
@pytest.mark.parametrize("tol", TOLS)
def test_exponential_sdf_1d(oracle, tol, derivative):
    # test/exponential_sdf_1d.jl: S = exp(-|w|), K = 2/(1+(2 pi r)^2), xgrid = range(0, 5.1, 1000)
    xgrid = np.linspace(0.0, 5.1, 1000)
    if derivative:
        xgrid = xgrid[1:]
    truth = oracle.exponential_cov_deriv(xgrid) if derivative else oracle.exponential_cov(xgrid)
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.exponential(1.0), tol=tol, derivative=derivative)
    ks, errs = oracle.kernel_values(cfg, xgrid)  # k0 by quadrature as in the reference test
    assert np.all(np.abs(ks - truth) / 2.0 <= 10 * tol)


@pytest.mark.parametrize("derivative,tol", [(False, 1e-4), (False, 1e-8), (False, 1e-10), (False, 1e-12),
                                            (True, 1e-4), (True, 1e-8)])
def test_matern_sdf_dim1(oracle, tol, derivative):
    # test/matern_sdf.jl:4-27, dim = 1 (grid thinned 4x to keep the direct-sum oracle fast).
    # derivative=true at tol <= 1e-10 is NOT pinned: the restated algorithm reaches ~90*tol there
    # (the tail fit ignores the extra factor w of the K' integrand, src/adaptive.jl:204-220), and the
    # upstream test that would decide it is disabled (test/runtests.jl:22-27).  See DESIGN.md.
    parms = (2.14, 0.97, 0.89)
    xgrid = np.linspace(0.0, 5.1, 1000)[::4]
    k0 = float(oracle.matern_cov(0.0, parms))
    truth = oracle.matern_cov_deriv(xgrid, parms) if derivative else oracle.matern_cov(xgrid, parms)
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms), tol=tol, derivative=derivative)
    ks, errs = oracle.kernel_values(cfg, xgrid, k0=k0 if derivative else None)
    assert np.all(np.abs(ks - truth) / k0 <= 10 * tol)
    assert np.isnan(errs[0]) and np.all(np.isfinite(errs[1:]))  # src/adaptive.jl:137-144


def test_matern_sdf_dim2_direct_bessel(oracle):
    # test/matern_sdf.jl dim = 2 (direct Bessel branch, src/quadrature.jl:145-160); small rule for speed
    parms = (2.14, 0.97, 0.89)
    xgrid = np.linspace(0.0, 5.1, 120)
    k0 = float(oracle.matern_cov(0.0, parms, d=2))
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms, d=2), dim=2, tol=1e-8, quadspec=(256, 16))
    ks, _ = oracle.kernel_values(cfg, xgrid, k0=k0)
    assert np.all(np.abs(ks - oracle.matern_cov(xgrid, parms, d=2)) / k0 <= 10 * 1e-8)


@pytest.mark.parametrize("tol", [1e-4, 1e-8])
def test_singular_matern_dim1(oracle, tol):
    # test/matern_sdf.jl:36-63, dim = 1, alpha = 0.5; truth = 1F2 formula in 50 digits
    parms = (2.14, 0.97, 0.89)
    alpha = 0.5
    xgrid = np.linspace(0.0, 1.1, 1000)[::8]
    k0 = oracle.sing_matern_k0(alpha, parms)
    truth = oracle.sing_matern_cov(xgrid, -alpha, parms)
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms), tol=tol, alpha=alpha)
    ks, _ = oracle.kernel_values(cfg, xgrid, k0=k0)
    assert np.all(np.abs(ks[1:] - truth[1:]) / k0 <= 10 * tol)
    assert abs(ks[0] - k0) == 0.0


@pytest.mark.parametrize("dim", [1, 2])
def test_singular_matern_derivative_in_alpha(oracle, dim):
    # test/matern_sdf.jl:36-44,69-85 ("derivative in α"): logw=true, param_derivative=true, k0 = K(0);
    # truth = d/d(alpha) of the 1F2 formula (mpmath).  Pins the integration-by-parts panel
    # (src/quadrature.jl:186-228) for dim 1 (:cis kernels) and dim 2 (J0 / J1 kernels).
    parms = (2.14, 0.97, 0.89)
    alpha = (dim - 1) + 0.5
    tol = 1e-8
    xgrid = np.linspace(0.0, 1.1, 1000)[1::25]
    sdf = oracle.SDF.matern(*parms, d=dim)
    k0 = oracle.compute_k0(oracle.AdaptiveKernelConfig(sdf, dim=dim, alpha=alpha, tol=tol))
    cfg = oracle.AdaptiveKernelConfig(sdf, dim=dim, alpha=alpha, tol=tol, logw=True)
    ks, _ = oracle.kernel_values(cfg, xgrid, k0=k0, param_derivative=True)
    truth = oracle.sing_matern_cov_dalpha(xgrid, alpha, parms, d=dim)
    assert np.all(np.abs(ks - truth) / abs(k0) <= 10 * tol)


def test_logw_dim1_boundary_term_is_the_reference_one(oracle):
    # src/quadrature.jl:189 evaluates besselj(dim/2-1, .) also for dim = 1, i.e. J_{-1/2}(z) = sqrt(2/(pi z)) cos z
    # where integration by parts gives cos z.  The term only matters when [0, b] is bisected (never with the
    # default quadspec); the oracle keeps the reference's expression, so a bisected dim-1 run is off by far
    # more than tol while dim 2 stays exact.  This test documents that the behaviour is intended.
    parms = (2.14, 0.97, 0.89)
    xs = np.linspace(0.0, 1.1, 50)[1:]
    out = {}
    for dim, alpha in ((1, 0.5), (2, 1.5)):
        sdf = oracle.SDF.matern(*parms, d=dim)
        k0 = oracle.compute_k0(oracle.AdaptiveKernelConfig(sdf, dim=dim, alpha=alpha))
        cfg = oracle.AdaptiveKernelConfig(sdf, dim=dim, alpha=alpha, logw=True, quadspec=(64, 2))
        ks, _, tr = oracle.kernel_values(cfg, xs, k0=k0, param_derivative=True, return_trace=True)
        assert sum(1 for r in tr if r["a"] == 0.0) > 1
        out[dim] = np.abs(ks - oracle.sing_matern_cov_dalpha(xs, alpha, parms, d=dim)).max() / abs(k0)
    assert out[2] <= 1e-7 and out[1] > 1e-3


def test_warping_lags(oracle):
    # test/derivatives/warping.jl:4-23: kernel_values(cfg, lags)[1] ~ closed form (rtol sqrt(eps))
    p1, p2 = 1 / 50.0, 1.1
    x = np.linspace(1.1, 2.0, 100)
    lags = np.abs((1.0 / p1) ** p2 - (x / p1) ** p2)
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.exponential(1.0), tol=1e-12)
    kv, _ = oracle.kernel_values(cfg, lags)
    should = oracle.exponential_cov(lags)
    assert np.linalg.norm(kv - should) <= np.sqrt(np.finfo(float).eps) * max(np.linalg.norm(kv), np.linalg.norm(should))


def test_ctor_errors(oracle):
    # src/adaptive.jl:29-35, src/quadrature.jl:40
    with pytest.raises(ValueError):
        oracle.AdaptiveKernelConfig(oracle.SDF.exponential(), convergence_criteria="nope")
    with pytest.raises(ValueError):
        oracle.AdaptiveKernelConfig(oracle.SDF.exponential(), alpha=1.0)
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.exponential(), tol=1e-13)
    assert cfg.warned and cfg.quadspec == (4096, 1)  # src/adaptive.jl:37-40
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.exponential(), tol=1e-12)
    assert not cfg.warned and cfg.quadspec == (4096, 16)  # strict '<'


def test_tail_fit_quirk_schedule(oracle):
    # C1 marches exactly two outer panels [0,32768], [32768,65536]
    rs = 10.0 ** np.linspace(-6, 0, 50)
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(1, 1, 1.5))
    _, _, tr = oracle.kernel_values(cfg, rs, k0=np.pi / 2, return_trace=True)
    assert [(r["a"], r["b"], r["accepted"]) for r in tr] == [(0.0, 32768.0, True), (32768.0, 65536.0, True)]


# ------------------------------------------------------------------ the remaining ladders of the reference's tests

@pytest.mark.parametrize("derivative,tol", [(False, 1e-4), (False, 1e-8), (False, 1e-10), (False, 1e-12),
                                            (True, 1e-4), (True, 1e-8)])
def test_matern_sdf_dim2_ladder(oracle, tol, derivative):
    # test/matern_sdf.jl:7-27 for dim = 2 (J0 / J1 kernels through the direct Bessel branch), value and r-derivative
    # over the tolerance ladder, default quadspec as in the reference; grid thinned 20x (direct sums).
    # derivative=true at tol <= 1e-10: see test_derivative_high_accuracy_stops_short_by_truncation.
    parms = (2.14, 0.97, 0.89)
    xgrid = np.linspace(0.0, 5.1, 1000)[::20]
    k0 = float(oracle.matern_cov(0.0, parms, d=2))
    truth = oracle.matern_cov_deriv(xgrid, parms, d=2) if derivative else oracle.matern_cov(xgrid, parms, d=2)
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms, d=2), dim=2, tol=tol, derivative=derivative)
    ks, errs = oracle.kernel_values(cfg, xgrid, k0=k0 if derivative else None)
    assert np.all(np.abs(ks - truth) / k0 <= 10 * tol)


def _sing_truth_deriv(oracle, xs, alpha, parms, d):
    import mpmath as mp
    out = []
    with mp.workdps(50):
        for x in xs:
            out.append(float(mp.diff(lambda t: oracle._sing_matern_mp(t, -alpha, parms, d), mp.mpf(float(x)))))
    return np.array(out)


@pytest.mark.parametrize("dim,derivative,tol", [(2, False, 1e-4), (2, False, 1e-8), (2, False, 1e-10), (2, True, 1e-4),
                                                (2, True, 1e-8), (1, False, 1e-10), (1, False, 1e-12), (1, True, 1e-4),
                                                (1, True, 1e-8)])
def test_singular_matern_ladder(oracle, dim, derivative, tol):
    # test/matern_sdf.jl:36-63: alpha = (dim-1) + 0.5, value and r-derivative; truth: the 1F2 formula of
    # scripts/matern_pair.jl:20-33 in 50 digits (its r-derivative by mpmath differentiation; ForwardDiff upstream)
    parms = (2.14, 0.97, 0.89)
    alpha = (dim - 1) + 0.5
    xgrid = np.linspace(0.0, 1.1, 1000)[1::40]
    sdf = oracle.SDF.matern(*parms, d=dim)
    k0 = float(oracle.sing_matern_cov([0.0], -alpha, parms, d=dim)[0]) if dim == 2 else oracle.sing_matern_k0(alpha, parms)
    k0q = oracle.compute_k0(oracle.AdaptiveKernelConfig(sdf, dim=dim, alpha=alpha, tol=1e-10))
    assert abs(k0q - k0) <= 1e-8 * abs(k0)  # compute_k0 (src/adaptive.jl:74-91) against the closed form at r -> 0
    truth = _sing_truth_deriv(oracle, xgrid, alpha, parms, dim) if derivative else oracle.sing_matern_cov(xgrid, -alpha, parms, d=dim)
    cfg = oracle.AdaptiveKernelConfig(sdf, dim=dim, alpha=alpha, tol=tol, derivative=derivative,
                                      quadspec=(2 ** 12, 2 ** 4) if dim == 1 else (2 ** 10, 2 ** 4))
    ks, _ = oracle.kernel_values(cfg, xgrid, k0=k0)
    assert np.all(np.abs(ks - truth) / abs(k0) <= 10 * tol)


def test_sdf_parameter_derivatives(oracle):
    # test/derivatives/sdf_params.jl:8-24: Matern (2.3, 0.1, 1.75), tol = 1e-12, 30 lags in [0, 3.5]; the integrands
    # dS/dtheta_j (ForwardDiff upstream, closed forms here) with param_derivative=true; truth = d/dtheta_j of the
    # closed-form pair (central differences; upstream ForwardDiff), the reference's own bar 1e-5
    parms = (2.3, 0.1, 1.75)
    xs = np.linspace(0.0, 3.5, 30)
    phi, rho, nu = parms
    S = lambda w: phi * (rho ** 2 + w * w) ** (-nu - 0.5)
    dS = [lambda w: S(w) / phi,
          lambda w: S(w) * (-nu - 0.5) * 2 * rho / (rho ** 2 + w * w),
          lambda w: -S(w) * np.log(rho ** 2 + w * w)]
    k0 = oracle.compute_k0(oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms), tol=1e-12))
    assert abs(k0 - float(oracle.matern_cov(0.0, parms))) <= 1e-9 * k0
    for j in range(3):
        cfg = oracle.AdaptiveKernelConfig(lambda w, j=j: float(dS[j](w)), tol=1e-12)  # gen_new_sdf_config: defaults
        ks, _ = oracle.kernel_values(cfg, xs, k0=k0, param_derivative=True)
        h = 1e-6 * max(1.0, abs(parms[j]))
        pp, pm = list(parms), list(parms)
        pp[j] += h
        pm[j] -= h
        fd = (oracle.matern_cov(xs, pp) - oracle.matern_cov(xs, pm)) / (2 * h)
        assert np.abs(ks - fd).max() < 1e-5 * max(1.0, np.abs(fd).max())


def test_nll_2d_lag_set(oracle):
    # test/derivatives/nll_2d.jl:7-45: 16 lattice points, warp(par, x) = (x1 p1, x1 x2 / p2), the 136 lags of
    # dense_index_pairs (src/model.jl:18-22, 64-65), dim = 2 SDF (range^2 + w^2)^(-nu-1), tol = 1e-12 -- what
    # gen_kernel hands to kernel_values (src/model.jl:73-77); truth: matern_cov(., (1, range, nu); d = 2)
    p1, p2, rng_, nu = 1.5, 0.1, 1.3, 0.8
    g1d = np.linspace(0.1, 1.0, 4)
    pts = np.array([(x, y) for y in g1d for x in g1d])
    wp = np.stack([pts[:, 0] * p1, pts[:, 0] * pts[:, 1] / p2], 1)
    jj, kk = np.triu_indices(16)
    lags = np.linalg.norm(wp[jj] - wp[kk], axis=1)
    assert lags.size == 136
    parms = (1.0, rng_, nu)
    k0 = float(oracle.matern_cov(0.0, parms, d=2))
    cfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*parms, d=2), dim=2, tol=1e-12, quadspec=(2 ** 10, 2 ** 4))
    assert abs(oracle.compute_k0(cfg) - k0) <= 1e-10 * k0
    ks, errs = oracle.kernel_values(cfg, lags)
    assert np.abs(ks - oracle.matern_cov(lags, parms, d=2)).max() / k0 <= 10 * 1e-12
    assert np.all(ks[lags == 0] == ks[0])


def test_derivative_high_accuracy_stops_short_by_truncation(oracle):
    # The one known-answer case of the reference that the restated algorithm does not meet: Matern,
    # derivative=true, tol <= 1e-10 (test/matern_sdf.jl:15-27; the upstream file is disabled, test/runtests.jl:22-27).
    # Independent evaluation: QUADPACK's oscillatory rule (QAWO) on the integral TRUNCATED at the b where the lag
    # retired agrees with the march far below tol -- the panel quadrature is right -- while the truncated integral
    # itself is several tol away from the closed form: the lag retired too early.  The missing tail matches the
    # leading term of integration by parts, 2 b S(b) |cos(2 pi b r)| / r: estimate_tail_decay fits S (src/adaptive.jl:
    # 204-220, f = cfg.f) although the K' integrand decays like w S(w), so truncation_error_estimate (:222-229) is
    # too small by a factor ~ b and check_convergence (:231-233) lets the lag go.
    from scipy import integrate
    parms = (2.14, 0.97, 0.89)
    k0 = float(oracle.matern_cov(0.0, parms))
    xs = np.linspace(0.0, 5.1, 1000)[1::100]
    S = oracle.SDF.matern(*parms)
    tol = 1e-10
    cfg = oracle.AdaptiveKernelConfig(S, tol=tol, derivative=True)
    ks, errs, tr = oracle.kernel_values(cfg, xs, k0=k0, return_trace=True)
    truth = oracle.matern_cov_deriv(xs, parms)
    for i in (xs.size // 2, xs.size - 3):
        b_i = max(r["b"] for r in tr if r["accepted"] and r["n_active"] > i)  # lags retire from the top
        val, _ = integrate.quad(lambda w: w * float(S(w)), 0.0, b_i, weight="sin", wvar=2 * np.pi * xs[i], limit=4000,
                                epsabs=1e-17, epsrel=1e-14)
        trunc = -4.0 * np.pi * val  # c = 2 * (-2 pi) (src/adaptive.jl:43-44)
        assert abs(ks[i] - trunc) / k0 <= 0.05 * tol           # the march integrates [0, b] correctly
        missing = abs(trunc - truth[i]) / k0
        assert missing >= 1.5 * tol                             # but stops short of the tolerance
        lead = 2.0 * b_i * float(S(b_i)) * abs(np.cos(2 * np.pi * b_i * xs[i])) / xs[i] / k0
        assert 0.5 * lead <= missing <= 2.0 * lead              # ... by the tail it did not see
        assert errs[i] / k0 < 0.01 * tol                        # and its own error estimate does not know


def test_nan_precedence_follows_the_reference(oracle):
    """src/quadrature.jl:165: `any(isnan, int1) || any(isnan, int2) && throw(...)` only throws when int1 is NaN-free.
    An SDF that is NaN everywhere poisons both rules: no throw, the panel is never accepted and is bisected down to
    the b - a < 1e-16 guard (src/utils.jl:28-36) -- the product's SK_ESUBDIVIDE, not SK_ENAN."""
    import pytest as _pt
    cfg = oracle.AdaptiveKernelConfig(lambda w: w * float("nan"), quadspec=(8, 2))
    with _pt.raises(RuntimeError) as e:
        oracle.kernel_values(cfg, np.array([0.5, 1.0]), k0=1.0)
    assert "split too many times" in str(e.value)
    # NaN in the 2m-rule only (a pole that only the finer rule's nodes hit is not constructible robustly; emulate
    # with a closure that is NaN on a set of measure zero hit by no node: stays finite) -> plain success
    ok = oracle.AdaptiveKernelConfig(lambda w: np.exp(-w), quadspec=(8, 2))
    ks, _ = oracle.kernel_values(ok, np.array([0.5, 1.0]), k0=2.0)
    assert np.all(np.isfinite(ks))


def test_derivative_high_accuracy_regime_is_unstable_in_the_oracle_itself(oracle):
    """stress seed 27 (derivative=true, tol 1e-10, quadspec (128, 8)) -- HIP and oracle end
    2.4e-7 K(0) apart.  The discriminating experiment on the CPU: shifting the oracle's own thresholds by 1e-4
    relative (5e-15 K(0) absolute in tol|k0|/2, below any NUFFT's accuracy) changes the ~12 000-panel schedule and
    moves the oracle's result by the same 2.4e-7 K(0), while its distance to the closed form stays 1.26e-7 K(0):
    in this regime (the algorithm stops short by ~1000 tol) every schedule is as good as any other."""
    from stress_cases import seed27_case
    rs, k0, mk, truth = seed27_case(oracle)
    res = []
    for eps in (0.0, 1e-4):
        ks, _, tr = oracle.kernel_values(mk(oracle, nthreads=8), rs, k0=k0 * (1 + eps), return_trace=True)
        res.append((ks, len(tr)))
    spread = np.abs(res[0][0] - res[1][0]).max() / abs(k0)
    assert res[0][1] != res[1][1]            # another schedule
    assert 1e-7 < spread < 1e-6              # ... and a result 2000 tol away
    for ks, _ in res:                        # both equally far from the truth
        assert 0.5e-7 < np.abs(ks - truth).max() / abs(k0) < 2e-7
