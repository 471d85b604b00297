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
