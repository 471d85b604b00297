"""CPU tier: the arithmetic of the Hankel inner box (csrc/sk_hankel_inner.inc) in its numpy prototype
(tools/proto_hankel_inner.py) -- Reinsch-form Chebyshev moments, exact-cosine DCT to equivalent sources, Bessel values at
the interval nodes, Chebyshev interpolation in r -- against direct Bessel sums."""
import os
import sys

import numpy as np
import pytest
from scipy.special import jv, roots_legendre

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import proto_hankel_inner as P  # noqa: E402


@pytest.mark.parametrize("panel,nu", [(0, 0), (1, 1), (7, 0)])
def test_inner_box_matches_direct_sums(panel, nu):
    m, k, nbands, r_hi = 64, 6, 3, 1.4
    t, wt = roots_legendre(2 * m)
    width = m * k / (2 * r_hi)
    pa, pb = panel * width, (panel + 1) * width
    sub = np.linspace(pa, pb, k + 1)
    w = np.concatenate([(sub[i + 1] - sub[i]) / 2 * t + (sub[i + 1] + sub[i]) / 2 for i in range(k)])
    c = np.concatenate([wt * (sub[i + 1] - sub[i]) / 2 for i in range(k)]) * (w * 0.7 * (1.3 ** 2 + w ** 2) ** -2.8 if panel == 0 else np.cos(w) / w)
    r_c = r_hi / 8
    rng = np.random.default_rng(panel)
    r = np.concatenate([r_c * 10.0 ** rng.uniform(-6, 0, 60), r_c * rng.uniform(0, 1, 60), [r_c]])
    ref = np.array([np.sum(c * jv(nu, 2 * np.pi * w * ri)) for ri in r])
    per = (k // nbands) * len(t)
    wb = [w[b * per:(b + 1) * per] for b in range(nbands)]
    cb = [c[b * per:(b + 1) * per] for b in range(nbands)]
    got, q, evals = P.inner_region(wb, cb, sub[::k // nbands], r, r_c, pb, nu)
    assert np.abs(got - ref).max() <= 5e-14 * np.abs(c).sum()
    assert q == max(1, int(np.ceil(np.pi * pb * r_c / 96.0)))


def test_reinsch_recurrence_is_stable_at_the_band_ends():
    # moments up to degree 900 of sources crowded at x = -1: the plain three-term recurrence loses k^2 eps there
    x = -1.0 + 10.0 ** np.linspace(-9, -1, 400)
    c = np.ones_like(x)
    got = P.moments_reinsch(x, c, 900)
    th = np.arccos(x.astype(np.longdouble))
    ref = np.array([np.sum(np.cos(kk * th)) for kk in range(900)], dtype=np.longdouble)
    assert float(np.abs(got - ref).max()) <= 2e-13 * c.sum()
