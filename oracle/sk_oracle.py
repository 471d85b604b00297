"""ctypes front-end of the CPU oracle (oracle/sk_oracle.c) + closed-form truths.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

Mirrors the reference API names (src/SpectralKernels.jl:12): AdaptiveKernelConfig,
kernel_values, compute_k0.  Gauss rules come from scipy (independent of the product's
C++ generator); K(0) uses QUADPACK (scipy.integrate.quad) on the reference's integrand
(src/adaptive.jl:74-91).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
from scipy import integrate, special

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SDF_CALLBACK, SDF_MATERN, SDF_EXPONENTIAL, SDF_GAUSSIAN = 0, 1, 2, 3
CRITERIA = {"both": 0, "panel": 1, "tails": 2}
ERRORS = {2: "sub-interval split too many times (b - a < 1e-16)", 3: "NaN detected in panel integral",
          4: "splitting stack overflow", 7: "singularity derivative not implemented in d > 2"}

_SDF_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)


class _Cfg(C.Structure):
    _fields_ = [("sdf_kind", C.c_int), ("sdf_params", C.c_double * 8), ("f", _SDF_FN), ("ctx", C.c_void_p),
                ("dim", C.c_int), ("alpha", C.c_double), ("tol", C.c_double), ("derivative", C.c_int),
                ("logw", C.c_int), ("criteria", C.c_int), ("has_tail", C.c_int), ("tail", C.c_double),
                ("m", C.c_long), ("k", C.c_long),
                ("leg_no1", C.c_void_p), ("leg_wt1", C.c_void_p), ("leg_no2", C.c_void_p), ("leg_wt2", C.c_void_p),
                ("jac_no1", C.c_void_p), ("jac_wt1", C.c_void_p), ("jac_no2", C.c_void_p), ("jac_wt2", C.c_void_p),
                ("nthreads", C.c_int), ("use_nufft", C.c_int), ("df", _SDF_FN)]


class _Rec(C.Structure):
    _fields_ = [("a", C.c_double), ("b", C.c_double), ("max_err", C.c_double), ("n_active", C.c_long),
                ("accepted", C.c_int)]


class _Trace(C.Structure):
    _fields_ = [("recs", C.POINTER(_Rec)), ("cap", C.c_long), ("len", C.c_long), ("n_outer", C.c_long)]


def build(force=False):
    """Compile oracle/sk_oracle.c -> oracle/libsk_oracle.so (gcc, OpenMP)."""
    so = os.path.join(_HERE, "libsk_oracle.so")
    src = os.path.join(_HERE, "sk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.sko_kernel_values_sorted.restype = C.c_int
        _LIB.sko_kernel_values_sorted.argtypes = [C.POINTER(_Cfg), C.c_void_p, C.c_long, C.c_double, C.c_double,
                                                  C.c_void_p, C.c_void_p, C.POINTER(_Trace)]
        _LIB.sko_direct_sums.restype = None
        _LIB.sko_direct_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                         C.c_void_p, C.c_int]
        _LIB.sko_sdf_eval.restype = C.c_double
        _LIB.sko_sdf_eval.argtypes = [C.POINTER(_Cfg), C.c_double]
        _LIB.sko_max_threads.restype = C.c_int
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class SDF:
    """A spectral density: either a catalogue entry (kind, params) or a Python callable."""

    def __init__(self, kind, params=(), fn=None):
        self.kind, self.params, self.fn = kind, tuple(float(p) for p in params), fn

    @staticmethod
    def matern(phi, rho, nu, d=1):
        """scripts/matern_pair.jl:17"""
        return SDF(SDF_MATERN, (phi, rho, nu, d))

    @staticmethod
    def exponential(a=1.0):
        """test/exponential_sdf_1d.jl:3"""
        return SDF(SDF_EXPONENTIAL, (a,))

    @staticmethod
    def gaussian(a=1.0):
        return SDF(SDF_GAUSSIAN, (a,))

    @staticmethod
    def callback(fn):
        return SDF(SDF_CALLBACK, (), fn)

    def __call__(self, w):
        w = np.asarray(w, dtype=float)
        if self.kind == SDF_MATERN:
            phi, rho, nu, d = self.params
            return phi * (rho * rho + w * w) ** (-nu - d / 2.0)
        if self.kind == SDF_EXPONENTIAL:
            return np.exp(-self.params[0] * np.abs(w))
        if self.kind == SDF_GAUSSIAN:
            return np.exp(-(w * w) / self.params[0] ** 2)
        return np.vectorize(self.fn, otypes=[float])(w)


class AdaptiveKernelConfig:
    """src/adaptive.jl:2-59 (struct + constructor), same keyword names."""

    def __init__(self, f, df=None, dim=1, alpha=0.0, tol=1e-8, derivative=False, logw=False,
                 convergence_criteria="both", tail=None, quadspec=(2 ** 12, 2 ** 4), nthreads=None,
                 use_nufft=False):
        if not isinstance(f, SDF):
            f = SDF.callback(f)
        if convergence_criteria not in CRITERIA:
            raise ValueError("Argument convergence_criteria must be one of :panel, :tails, :both.")
        if alpha >= dim:
            raise ValueError("alpha must be less than dim to be integrable.")
        self.warned = False
        if tol < 1e-12 and quadspec[0] * quadspec[1] > 2 ** 12:  # adaptive.jl:37-40
            self.warned = True
            quadspec = (2 ** 12, 1)
        self.f, self.df, self.dim, self.alpha, self.tol = f, df, int(dim), float(alpha), float(tol)
        self.derivative, self.logw = bool(derivative), bool(logw)
        self.convergence_criteria, self.tail, self.quadspec = convergence_criteria, tail, tuple(quadspec)
        self.p = -self.alpha + (0 if dim == 1 else dim / 2) + (1 if derivative else 0)
        self.c = (2.0 if dim == 1 else 2 * math.pi) * (-2 * math.pi if derivative else 1.0) * (-1 if logw else 1)
        m = self.quadspec[0]
        self.leg = gauss_rules(m, 0.0)
        if self.p != 0.0:
            if self.p <= -1.0:
                raise ValueError("p needs to be in (-1.0, Inf) to be integrable")
            self.jac = gauss_rules(m, self.p)
        else:
            self.jac = self.leg
        self.nthreads = nthreads if nthreads else lib().sko_max_threads()
        self.use_nufft = bool(use_nufft)  # CPU type-3 NUFFT for the panel sums (baseline timing only)
        self._keep = None

    def _cstruct(self):
        s = _Cfg()
        s.sdf_kind = self.f.kind
        for i, v in enumerate(self.f.params):
            s.sdf_params[i] = v
        if self.f.kind == SDF_CALLBACK:
            fn = self.f.fn
            cb = _SDF_FN(lambda w, ctx: float(fn(w)))
            s.f = cb
            self._keep = [cb]
            if self.df is not None:
                dfn = self.df
                dcb = _SDF_FN(lambda w, ctx: float(dfn(w)))
                self._keep.append(dcb)
                s.df = dcb
        s.dim, s.alpha, s.tol = self.dim, self.alpha, self.tol
        s.derivative, s.logw = int(self.derivative), int(self.logw)
        s.criteria = CRITERIA[self.convergence_criteria]
        s.has_tail, s.tail = (0, 0.0) if self.tail is None else (1, float(self.tail))
        s.m, s.k = self.quadspec
        (s.leg_no1, s.leg_wt1, s.leg_no2, s.leg_wt2) = [_ptr(a) for a in self.leg]
        (s.jac_no1, s.jac_wt1, s.jac_no2, s.jac_wt2) = [_ptr(a) for a in self.jac]
        s.nthreads = self.nthreads
        s.use_nufft = int(self.use_nufft)
        return s


_RULE_CACHE = {}


def _jacobi_eval_ld(n, b, x):
    """P_n^{(0,b)}(x) and derivative by the three-term recurrence in long double (vectorised)."""
    ld = np.longdouble
    x = x.astype(ld)
    b = ld(b)
    p0 = np.ones_like(x)
    p1 = ((b + 2) * x - b) / 2
    for k in range(2, n + 1):
        kk = ld(k)
        s = 2 * kk + b
        c1 = 2 * kk * (kk + b) * (s - 2)
        c2 = (s - 1) * (s * (s - 2) * x - b * b)
        c3 = 2 * (kk - 1) * (kk + b - 1) * s
        p0, p1 = p1, (c2 * p1 - c3 * p0) / c1
    nn = ld(n)
    s = 2 * nn + b
    dp = (nn * (-b - s * x) * p1 + 2 * nn * (nn + b) * p0) / (s * (1 - x * x))
    return p1, dp


def gauss_rules(m, p):
    """QuadRule (src/quadrature.jl:35-47): (no1, wt1, no2, wt2) for m and 2m nodes;
    gausslegendre for p == 0, gaussjacobi(n, 0.0, p) otherwise.  scipy supplies the nodes;
    scipy's weights lose up to 6 digits at the panel ends for n ~ 10^4, so nodes are
    Newton-polished and weights recomputed from w = 2^(p+1)/((1-x^2) P_n'(x)^2) in long double."""
    key = (int(m), float(p))
    gold = os.path.join(os.path.dirname(_HERE), "tests", "golden", "rules_m%d_p%+.3f.npz" % key)
    if key not in _RULE_CACHE and os.path.exists(gold):
        z = np.load(gold)
        _RULE_CACHE[key] = tuple(np.ascontiguousarray(z[k]) for k in ("no1", "wt1", "no2", "wt2"))
    if key not in _RULE_CACHE:
        out = []
        for n in (m, 2 * m):
            if p == 0.0:
                x, _ = special.roots_legendre(n)
            else:
                x, _ = special.roots_jacobi(n, 0.0, p)
            x = x.astype(np.longdouble)
            for _ in range(2):
                pn, dp = _jacobi_eval_ld(n, p, x)
                x = x - pn / dp
            pn, dp = _jacobi_eval_ld(n, p, x)
            w = np.longdouble(2.0) ** np.longdouble(p + 1) / ((1 - x * x) * dp * dp)
            out += [np.ascontiguousarray(x.astype(np.float64)), np.ascontiguousarray(w.astype(np.float64))]
        _RULE_CACHE[key] = tuple(out)
    return _RULE_CACHE[key]


def compute_k0(cfg):
    """src/adaptive.jl:74-91 with QUADPACK qagi standing in for QuadGK."""
    f, p = cfg.f, cfg.p
    L = 1.0
    while L ** p * abs(float(f(L))) > abs(float(f(0.0))) / 2:
        L *= 2
    if cfg.dim == 1:
        def integrand(w):
            wl = w * L
            return wl ** p * (math.log(wl) if cfg.logw else 1.0) * float(f(wl)) * L
    else:
        nu = cfg.dim / 2 - 1 + (1 if cfg.derivative else 0)

        def integrand(w):
            wl = w * L
            return (math.pi * w) ** nu / math.gamma(nu + 1) * wl ** p * (math.log(wl) if cfg.logw else 1.0) \
                * float(f(wl)) * L
    rtol = min(1e-8, 1e-2 * cfg.tol)
    val, _ = integrate.quad(integrand, 0.0, np.inf, epsabs=0.0, epsrel=max(rtol, 5e-14), limit=2000)
    return cfg.c * val


def kernel_values(cfg, xs, k0=None, param_derivative=False, return_trace=False):
    """src/adaptive.jl:95-108: dedup -> sorted core -> gather back in input order.
    (unique + sortperm + Dict lookup == np.unique with inverse.)"""
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    if k0 is None:
        k0 = compute_k0(cfg)
    if xs.size == 0:
        z = np.zeros(0)
        return (z, z.copy(), []) if return_trace else (z, z.copy())
    uxs, inv = np.unique(xs, return_inverse=True)
    n = uxs.size
    ks0 = 0.0
    if uxs[0] == 0.0:  # adaptive.jl:132-146
        ks0 = 0.0 if cfg.derivative else (compute_k0(cfg) if param_derivative else k0)
    ks, errs = np.zeros(n), np.zeros(n)
    cap = 20000
    recs = (_Rec * cap)()
    tr = _Trace(recs, cap, 0, 0)
    cs = cfg._cstruct()
    rc = lib().sko_kernel_values_sorted(C.byref(cs), _ptr(uxs), n, float(k0), float(ks0), _ptr(ks), _ptr(errs),
                                        C.byref(tr))
    if rc:
        raise RuntimeError(ERRORS.get(rc, "oracle error %d" % rc))
    out = (ks[inv], errs[inv])
    if return_trace:
        trace = [dict(a=recs[i].a, b=recs[i].b, max_err=recs[i].max_err, n_active=recs[i].n_active,
                      accepted=bool(recs[i].accepted)) for i in range(tr.len)]
        return out + (trace,)
    return out


def direct_sums(w, c, xs, nthreads=None):
    """finufft1d3(w, c, xs) semantics (src/utils.jl:10) by exact summation: complex vector."""
    w = np.ascontiguousarray(w, dtype=float)
    c = np.ascontiguousarray(c, dtype=float)
    xs = np.ascontiguousarray(xs, dtype=float)
    re, im = np.zeros(xs.size), np.zeros(xs.size)
    lib().sko_direct_sums(_ptr(w), _ptr(c), w.size, _ptr(xs), xs.size, _ptr(re), _ptr(im),
                          nthreads or lib().sko_max_threads())
    return re + 1j * im


# ---------------------------------------------------------------------------------------
# Closed-form truths (scripts/matern_pair.jl, test/exponential_sdf_1d.jl)
# ---------------------------------------------------------------------------------------

def matern_cov(t, parms, d=1):
    """scripts/matern_pair.jl:7-14.  K(r) for S = phi (rho^2+w^2)^(-nu-d/2)."""
    phi, rho, nu = parms
    t = np.abs(np.asarray(t, dtype=float))
    const = math.pi ** (d / 2) * phi / (2 ** (nu - 1) * math.gamma(nu + d / 2) * rho ** (2 * nu))
    arg = rho * 2 * math.pi * t
    with np.errstate(invalid="ignore", over="ignore"):
        val = const * special.kv(nu, arg) * arg ** nu
    lim = const * math.gamma(nu) * 2 ** (nu - 1)  # x^nu K_nu(x) -> Gamma(nu) 2^(nu-1)
    return np.where(arg == 0, lim, val)


def matern_cov_deriv(t, parms, d=1):
    """d/dr of matern_cov: d/dx [x^nu K_nu(x)] = -x^nu K_{nu-1}(x)."""
    phi, rho, nu = parms
    t = np.asarray(t, dtype=float)
    const = math.pi ** (d / 2) * phi / (2 ** (nu - 1) * math.gamma(nu + d / 2) * rho ** (2 * nu))
    arg = rho * 2 * math.pi * np.abs(t)
    with np.errstate(invalid="ignore", over="ignore"):
        val = -const * arg ** nu * special.kv(nu - 1, arg) * (rho * 2 * math.pi)
    return np.where(arg == 0, 0.0, val)


def exponential_cov(r, a=1.0):
    """test/exponential_sdf_1d.jl:4 generalised: S=exp(-a|w|) <-> K=2a/(a^2+(2 pi r)^2)."""
    r = np.asarray(r, dtype=float)
    return 2 * a / (a * a + (2 * math.pi * r) ** 2)


def exponential_cov_deriv(r, a=1.0):
    r = np.asarray(r, dtype=float)
    return -2 * a * 8 * math.pi ** 2 * r / (a * a + (2 * math.pi * r) ** 2) ** 2


def sing_matern_cov(t, p, parms, d=1, dps=50):
    """scripts/matern_pair.jl:20-33 (1F2 difference formula), evaluated with mpmath at `dps`
    digits because the Float64 formula is unstable for t*rho > 2 (:22)."""
    import mpmath as mp
    phi, a, b = parms
    out = []
    with mp.workdps(dps):
        phi, a, b, p_, d_ = [mp.mpf(float(v)) for v in (phi, a, b, p, d)]
        pi = mp.pi
        for ti in np.atleast_1d(np.asarray(t, dtype=float)):
            tt = mp.mpf(float(ti)) + mp.mpf("1e-30")
            z = a ** 2 * pi ** 2 * tt ** 2
            o = pi ** p_ * (a * tt) ** p_ * mp.gamma((d_ + p_) / 2) * mp.hyp1f2((d_ + p_) / 2, d_ / 2,
                                                                              (2 - 2 * b + p_) / 2, z) \
                / (mp.gamma(d_ / 2) * mp.gamma((2 - 2 * b + p_) / 2))
            o -= pi ** (2 * b) * (a * tt) ** (2 * b) * mp.gamma(b + d_ / 2) \
                * mp.hyp1f2(b + d_ / 2, 1 + b - p_ / 2, b + d_ / 2 - p_ / 2, z) \
                / (mp.gamma(1 + b - p_ / 2) * mp.gamma(b + d_ / 2 - p_ / 2))
            o *= phi * a ** (-2 * b) * pi ** (1 + d_ / 2 - p_) * tt ** (-p_) * mp.csc(b * pi - (p_ * pi) / 2) \
                / mp.gamma(b + d_ / 2)
            out.append(float(o))
    return np.array(out)


def sing_matern_cov_dalpha(t, alpha, parms, d=1, dps=50):
    """d/d(alpha) of sing_matern_cov(t, -alpha, parms) -- the truth of the reference's "derivative in α"
    test (test/matern_sdf.jl:69-85, ForwardDiff there; mpmath numerical differentiation here)."""
    import mpmath as mp
    out = []
    with mp.workdps(dps):
        for ti in np.atleast_1d(np.asarray(t, dtype=float)):
            def g(a, ti=ti):
                return _sing_matern_mp(mp.mpf(float(ti)) + mp.mpf("1e-30"), -a, parms, d)
            out.append(float(mp.diff(g, mp.mpf(float(alpha)))))
    return np.array(out)


def _sing_matern_mp(tt, p_, parms, d):
    import mpmath as mp
    phi, a, b = [mp.mpf(float(v)) for v in parms]
    d_ = mp.mpf(float(d))
    pi = mp.pi
    z = a ** 2 * pi ** 2 * tt ** 2
    o = pi ** p_ * (a * tt) ** p_ * mp.gamma((d_ + p_) / 2) * mp.hyp1f2((d_ + p_) / 2, d_ / 2, (2 - 2 * b + p_) / 2, z) \
        / (mp.gamma(d_ / 2) * mp.gamma((2 - 2 * b + p_) / 2))
    o -= pi ** (2 * b) * (a * tt) ** (2 * b) * mp.gamma(b + d_ / 2) \
        * mp.hyp1f2(b + d_ / 2, 1 + b - p_ / 2, b + d_ / 2 - p_ / 2, z) \
        / (mp.gamma(1 + b - p_ / 2) * mp.gamma(b + d_ / 2 - p_ / 2))
    o *= phi * a ** (-2 * b) * pi ** (1 + d_ / 2 - p_) * tt ** (-p_) * mp.csc(b * pi - (p_ * pi) / 2) / mp.gamma(b + d_ / 2)
    return o


def sing_matern_k0(alpha, parms, d=1):
    """K_alpha(0) = 2 int_0^inf w^-alpha phi (rho^2+w^2)^-s dw, s = nu + d/2 (dim 1 only):
    phi rho^(1-alpha-2s) B((1-alpha)/2, s-(1-alpha)/2)."""
    phi, rho, nu = parms
    s = nu + d / 2
    q = (1 - alpha) / 2
    return phi * rho ** (1 - alpha - 2 * s) * special.beta(q, s - q)
