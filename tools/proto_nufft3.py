"""Throwaway numpy prototype of the fixed-geometry 1-D type-3 NUFFT used by the HIP path.
Validates kernel width / oversampling / table degrees against direct sums."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import roots_legendre

def es_kernel(z, w, beta):
    t = 1.0 - (2.0 * z / w) ** 2
    out = np.zeros_like(z)
    m = t > 0
    out[m] = np.exp(beta * (np.sqrt(t[m]) - 1.0))
    return out

def es_hat(xi, w, beta, nq=64):
    # phi_hat(xi) = int_{-w/2}^{w/2} phi(z) cos(xi z) dz
    t, wt = roots_legendre(nq)
    z = (t + 1) * w / 4
    wz = wt * w / 4
    ph = es_kernel(z, w, beta)
    return 2.0 * (np.cos(np.outer(np.atleast_1d(xi), z)) * (ph * wz)).sum(axis=1)

def nufft3(u, c, v, Lam, n1, w=16, sigma2=2, beta_per_w=2.30):
    """G(v_i) = sum_j c_j exp(i Lam u_j v_i), |u|<=1, |v|<=1."""
    beta = beta_per_w * w
    delta = 2.0 / (n1 - w)
    n2 = sigma2 * n1
    sig_eff = np.pi / (Lam * delta)
    # spread (input-driven here, numpy)
    p = u / delta  # grid units in [-(n1-w)/2, (n1-w)/2]
    b = np.zeros(n1, dtype=complex)
    l0 = np.ceil(p - w / 2).astype(int)
    for q in range(w):
        l = l0 + q
        np.add.at(b, l + n1 // 2, c * es_kernel(l - p, w, beta))
    # deconvolve for the inner type 2 and zero-pad in FFT order
    l = np.arange(-n1 // 2, n1 // 2)
    d = b / es_hat(2 * np.pi * l / n2, w, beta)
    fin = np.zeros(n2, dtype=complex)
    fin[l % n2] = d
    # U_k = sum_l d_l exp(+2 pi i l k / n2)
    U = np.fft.ifft(fin) * n2
    theta = Lam * delta * v
    t = theta * n2 / (2 * np.pi)
    k0 = np.ceil(t - w / 2).astype(int)
    H = np.zeros(len(v), dtype=complex)
    for q in range(w):
        k = k0 + q
        H += U[k % n2] * es_kernel(k - t, w, beta)
    return H / es_hat(theta, w, beta), sig_eff

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    m, k = 4096, 16
    M = m * k
    t, wt = roots_legendre(m)
    edges = np.linspace(-1, 1, k + 1)
    u = np.concatenate([(edges[i + 1] - edges[i]) / 2 * t + (edges[i + 1] + edges[i]) / 2 for i in range(k)])
    c = np.concatenate([wt / k for _ in range(k)]) * (1 + rng.random(M))
    v = np.concatenate([np.linspace(-1, 1, 300), rng.uniform(-1, 1, 200)])
    Lam = np.pi * M / 4
    for w in (8, 10, 12, 14, 16):
        for bpw in (2.30,):
            G, se = nufft3(u, c.astype(complex), v, Lam, M, w=w, beta_per_w=bpw)
            # direct, high-precision-ish phase
            ref = np.array([np.sum(c * np.exp(1j * Lam * u * vi)) for vi in v])
            print(w, bpw, "sig_eff %.4f" % se, "max rel err %.2e" % (np.abs(G - ref).max() / np.abs(c).sum()))
