"""Numpy prototype of the inner region of the round-3 Hankel path:
   B(r) = sum_j c_j J_nu(2 pi w_j r)   for the lags r <= r_c  (r_c = r_hi / 8, or r_hi when the panel has no far field)
 by compression on both sides:
   1. the panel's sources, in NB bands of whole sub-panels, are replaced per band by their Chebyshev moments
      C_k = sum_j c_j T_k(x_j)  (x = band coordinate), by the Reinsch form of the three-term recurrence (2 FMAs a step,
      error growth O(k eps) up to the band ends);
   2. the lag range (0, r_c] is cut in Q equal intervals of r-bandwidth <= ZR; per interval and band the first
      N(Z_w) moments (Z_w = pi x band width x top of the interval) become N equivalent sources at Chebyshev nodes
      (a DCT with exact cosines), the equivalent sources are summed at NR Chebyshev nodes of the interval with J_nu itself, and the
      node values become Chebyshev coefficients in r -- evaluated at the lags by Clenshaw.
 Prints the error against direct sums and the number of J_nu evaluations."""
import numpy as np
from scipy.special import jv, roots_legendre

NR = 160
ZR = 96.0


def n_of_z(z):
    return int(np.ceil(z + 9.0 * z ** (1.0 / 3.0) + 14.0))


def moments_reinsch(x, c, ng):
    """C_k = sum_j c_j T_k(x_j), k < ng; per source the recurrence d <- delta T + s d, T <- s T + d (s = sign x)."""
    s = np.where(x >= 0.0, 1.0, -1.0)
    delta = 2.0 * (x - s)
    out = np.empty(ng)
    T = c.copy()                     # c T_0
    out[0] = T.sum()
    d = c * x - s * T                # c (T_1 - s T_0)
    for k in range(1, ng):
        T = s * T + d                # c T_k
        out[k] = T.sum()
        d = delta * T + s * d
    return out


def exact_cos_table(n):
    """cos(pi j / (2 n)), j < 4 n: the DCT matrices below index it with integer arithmetic, so T_k at the Chebyshev
    nodes is exact to rounding (cos of a rounded k theta is not: the error grows with k and lifts the coefficient
    tail to 1e-14)."""
    return np.cos(np.pi * np.arange(4 * n) / (2.0 * n))


def eq_weights(C, n):
    """W_i, i < n: sum_j c_j f(x_j) ~= sum_i W_i f(x_i) at the n first-kind Chebyshev nodes, from the first n moments:
    W_i = 2/n sum'_k C_k cos(pi k (2 i + 1) / (2 n)).  (Clenshaw at the nodes loses an order of magnitude here: the
    moments of a band whose mass sits at one end are all of size sum|c| and cancel.)"""
    i = np.arange(n)
    tab = exact_cos_table(n)
    Tm = tab[np.outer(2 * i + 1, np.arange(n)) % (4 * n)]
    val = Tm[:, 1:] @ C[1:n] + 0.5 * C[0]
    return tab[2 * i + 1], val * 2.0 / n


def inner_region(w_bands, c_bands, band_edges, r, r_c, pb, nu):
    q_cnt = max(1, int(np.ceil(np.pi * pb * r_c / ZR)))
    ell = r_c / q_cnt
    nb = len(w_bands)
    # stage 1: moments up to the largest degree any interval needs
    cm = []
    for b in range(nb):
        half = 0.5 * (band_edges[b + 1] - band_edges[b])
        mid = 0.5 * (band_edges[b + 1] + band_edges[b])
        ngmax = n_of_z(np.pi * 2 * half * r_c)
        cm.append(moments_reinsch((w_bands[b] - mid) / half, c_bands[b], ngmax))
    out = np.zeros_like(r)
    evals = 0
    tabr = exact_cos_table(NR)
    dct = tabr[np.outer(np.arange(NR), 2 * np.arange(NR) + 1) % (4 * NR)]  # [m][i]
    xr = tabr[2 * np.arange(NR) + 1]
    qi = np.minimum(q_cnt - 1, np.floor(r / ell).astype(int))
    for q in range(q_cnt):
        sel = qi == q
        if not sel.any():
            continue
        r1, r2 = q * ell, (q + 1) * ell
        rn = 0.5 * (r1 + r2) + 0.5 * ell * xr
        V = np.zeros(NR)
        for b in range(nb):
            half = 0.5 * (band_edges[b + 1] - band_edges[b])
            mid = 0.5 * (band_edges[b + 1] + band_edges[b])
            n = min(n_of_z(np.pi * 2 * half * r2), len(cm[b]))
            xi, W = eq_weights(cm[b], n)
            wn = mid + half * xi
            V += jv(nu, 2 * np.pi * np.outer(rn, wn)) @ W
            evals += NR * n
        a = dct @ V * 2.0 / NR
        a[0] *= 0.5
        xx = (r[sel] - 0.5 * (r1 + r2)) / (0.5 * ell)
        out[sel] = np.polynomial.chebyshev.chebval(xx, a)
    return out, q_cnt, evals


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for (m, k, nbands) in ((256, 16, 4), (64, 6, 3)):
        M = m * k
        for rule in (0, 1):
            t, wt = roots_legendre(m * (rule + 1))
            for (panel, r_hi, shrink) in ((0, 1.4, 1.0), (1, 1.4, 1.0), (7, 1.0, 1.0), (39, 0.7, 1.0), (0, 1.4, 0.25)):
                width = M / (2 * r_hi) * shrink
                pa = panel * width
                pb = pa + width
                sub = np.linspace(pa, pb, k + 1)
                w = np.concatenate([(sub[i + 1] - sub[i]) / 2 * t + (sub[i + 1] + sub[i]) / 2 for i in range(k)])
                wq = np.concatenate([wt * (sub[i + 1] - sub[i]) / 2 for i in range(k)])
                gs = k // nbands
                band_edges = sub[::gs]
                per = gs * len(t)
                for nu in (0, 1):
                    c = wq * w * 0.7 * (1.3 ** 2 + w ** 2) ** -2.8 if panel == 0 else wq * np.cos(w) / w
                    has_far = 24.0 * 8 / (2 * np.pi * r_hi) < pb
                    r_c = r_hi / 8 if has_far else r_hi
                    r = np.concatenate([r_c * 10.0 ** rng.uniform(-6, 0, 300), r_c * rng.uniform(0, 1, 300), [r_c]])
                    ref = np.array([np.sum(c * jv(nu, 2 * np.pi * w * ri)) for ri in r])
                    wb = [w[b * per:(b + 1) * per] for b in range(nbands)]
                    cb = [c[b * per:(b + 1) * per] for b in range(nbands)]
                    got, Q, ev = inner_region(wb, cb, band_edges, r, r_c, pb, nu)
                    print("m %d k %d B %d rule %d panel %2d r_hi %.1f shrink %.2f nu %d: Q %3d evals %.2fM  max err / sum|c| %.2e"
                          % (m, k, nbands, rule, panel, r_hi, shrink, nu, Q, ev / 1e6, np.abs(got - ref).max() / np.abs(c).sum()))
