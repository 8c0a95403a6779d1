"""Fit of the polynomial behind gelu_poly4 (stdiff_amd/csrc/common.h): Phi(x) = 0.5 + x*Q(u), u = 2x^2/25 - 1 on
|x| <= 5, degree 12, Lawson-reweighted least squares (near-minimax in |x * dQ| = |dPhi|); prints the power-basis
coefficients in u and the fp32-evaluated error against scipy's erf."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf


def phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


c, d = 5.0, 12
u = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000)
x = np.sqrt(c * c * (u + 1) / 2)
q = (phi(x) - 0.5) / x
w, lw = x + 1e-3, np.ones_like(u)
for _ in range(200):
    coef = C.chebfit(u, q, d, w=w * lw)
    err = np.abs((C.chebval(u, coef) - q) * x)
    lw = lw * (err / err.max() + 1e-3) ** 0.5
    lw /= lw.max()
pc = C.cheb2poly(coef).astype(np.float32)
print(", ".join(f"{v:.9e}f" for v in pc))
xs = np.linspace(-c, c, 200001).astype(np.float32)
uu = (xs * xs * np.float32(0.08) - np.float32(1)).astype(np.float32)
acc = np.full_like(uu, pc[-1])
for k in range(d - 1, -1, -1):
    acc = (acc * uu + pc[k]).astype(np.float32)
ph = (np.float32(0.5) + xs * acc).astype(np.float32)
x64 = xs.astype(np.float64)
print("max |dPhi|", np.abs(ph - phi(x64)).max(), "max |dGelu|", np.abs(xs * ph - x64 * phi(x64)).max())
