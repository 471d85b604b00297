"""Numpy prototype of the level-split Hankel sums used by the planned GPU NUFHT:
   B(r) = sum_j c_j J_nu(2 pi w_j r)
 per lag level l (r in [r_hi/beta^(l+1), r_hi/beta^l]):
   far  : sources w >= Om_l = z0 beta^(l+1)/(2 pi r_hi): K-term Hankel asymptotic expansion, each term a
          type-3 exponential sum  T_k(r) = sum_j c_j (w_j/Om_l)^-(k+1/2) exp(2 pi i w_j r)
   near : sources w <  Om_l: Chebyshev interpolation in r on the level interval (direct sums at the nodes)."""
import numpy as np
from scipy.special import jv, roots_legendre
from numpy.polynomial import chebyshev as Ch

def asym_coeffs(nu, K):
    a = [1.0]
    for k in range(1, K):
        a.append(a[-1] * (4 * nu * nu - (2 * k - 1) ** 2) / (8.0 * k))
    return np.array(a)

def hankel_levels(w, c, r, r_hi, nu=0, z0=24.0, K=16, beta=8.0, deg=None, direct_T=True):
    out = np.zeros_like(r)
    a = asym_coeffs(nu, K)
    chi = (2 * nu + 1) * np.pi / 4
    pb = w.max()
    L = 0
    while z0 * beta ** L / (2 * np.pi * r_hi) < pb:  # levels until the threshold passes the panel end
        L += 1
    edges = [r_hi / beta ** l for l in range(L + 1)] + [0.0]
    for l in range(L + 1):
        hi_r, lo_r = edges[l], edges[l + 1]
        m = (r <= hi_r) & (r > lo_r) if l < L else (r <= hi_r)
        if l == 0:
            m = (r > lo_r)
        if not m.any():
            continue
        Om = z0 * beta ** (l + 1) / (2 * np.pi * r_hi) if l < L else np.inf
        far = w >= Om
        near = ~far
        rr = r[m]
        val = np.zeros(rr.size)
        if far.any():
            wf, cf = w[far], c[far]
            acc = np.zeros(rr.size, dtype=complex)
            for k in range(K):
                Tk = np.array([np.sum(cf * (wf / Om) ** (-(k + 0.5)) * np.exp(2j * np.pi * ((wf * ri) % 1.0))) for ri in rr])
                acc += (1j ** k) * a[k] * (2 * np.pi * Om * rr) ** (-(k + 0.5)) * Tk
            val += np.real(np.exp(-1j * chi) * np.sqrt(2 / np.pi) * acc)
        if near.any():
            wn, cn = w[near], c[near]
            # Chebyshev interpolation on [lo_r, hi_r]; degree ~ (type * half length) + margin
            half = 0.5 * (hi_r - lo_r)
            typ = 2 * np.pi * wn.max()
            d = deg or int(typ * half + 30)
            f = lambda s: np.array([np.sum(cn * jv(nu, 2 * np.pi * wn * (lo_r + half * (si + 1)))) for si in s])
            cc = Ch.chebinterpolate(f, d)
            val += Ch.chebval((rr - lo_r) / half - 1, cc)
        out[m] = val
    return out, L

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    m, k = 256, 16
    M = m * k
    t, wt = roots_legendre(2 * m)
    for (pa, r_hi) in ((0.0, 1.0), (2048.0, 1.0), (0.0, 0.3)):
        pb = pa + M / (2 * r_hi)
        edges = np.linspace(pa, pb, k + 1)
        w = np.concatenate([(edges[i + 1] - edges[i]) / 2 * t + (edges[i + 1] + edges[i]) / 2 for i in range(k)])
        wq = np.concatenate([wt * (edges[i + 1] - edges[i]) / 2 for i in range(k)])
        for nu in (0, 1):
            c = wq * w ** (1 + nu * 0) * 0.7 * (1.3 ** 2 + w ** 2) ** -2.8
            r = np.concatenate([r_hi * 10.0 ** rng.uniform(-5, 0, 150), [r_hi]])
            ref = np.array([np.sum(c * jv(nu, 2 * np.pi * w * ri)) for ri in r])
            for z0, K in ((24.0, 16), (30.0, 14), (20.0, 18)):
                got, L = hankel_levels(w, c, r, r_hi, nu=nu, z0=z0, K=K)
                print("pa %.0f r_hi %.1f nu %d z0 %.0f K %d L %d: max err %.2e (scale %.2e)" % (pa, r_hi, nu, z0, K, L, np.abs(got - ref).max(), np.abs(c).sum()))
