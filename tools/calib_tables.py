"""Calibrate polynomial degrees for the ES-kernel tables (dev tool)."""
import numpy as np, mpmath as mp
from numpy.polynomial import chebyshev as Ch, polynomial as P
import sys
sys.path.insert(0, "/root/repo/tools")
from proto_nufft3 import es_kernel, es_hat

def piece_err(w, d, bpw=2.30):
    beta = bpw * w
    worst = 0
    for q in range(w // 2):
        g = lambda y: es_kernel(np.asarray(q - (w - 1) / 2 + y, dtype=np.longdouble).astype(float), w, beta)
        # chebyshev interpolation on [-1/2, 1/2]
        c = Ch.chebinterpolate(lambda s: g(s / 2), d)
        mono = Ch.cheb2poly(c)  # in s
        mono_y = mono * (2.0 ** np.arange(len(mono)))
        y = np.linspace(-0.5, 0.5, 2001)
        approx = P.polyval(y, mono_y)
        worst = max(worst, np.abs(approx - g(y)).max())
    return worst

for w in (8, 10, 12, 14, 16):
    print("w", w, [(d, "%.1e" % piece_err(w, d)) for d in range(w - 2, w + 7, 2)])

# 1/phi_hat fit as even polynomial in theta on [0, thmax]
for w in (8, 12, 16):
    beta = 2.30 * w
    thmax = np.pi / 2 * 1.01
    f = lambda s: 1.0 / es_hat(np.sqrt((s + 1) / 2) * thmax, w, beta, nq=128)  # s in [-1,1] -> theta^2 scaled
    for deg in (10, 14, 18, 22, 26, 30):
        c = Ch.chebinterpolate(f, deg)
        s = np.linspace(-1, 1, 3001)
        err = np.abs(Ch.chebval(s, c) - f(s)) / np.abs(f(s))
        print("w", w, "deg", deg, "max rel err %.1e" % err.max(), "ratio hat(0)/hat(max) %.1e" % (f(1.0) / f(-1.0)))
