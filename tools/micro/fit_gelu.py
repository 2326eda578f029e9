"""Coefficients of the sigmoid-form GELU used by the fused GEMM epilogues (grit_amd/csrc/gemm.hip):
gelu(x) ~= x * sigmoid(x * (c0 + c1 x^2 + c2 x^4)), x^2 clamped at 50 -- minimax fit against 0.5 x (1 + erf(x / sqrt 2)) on [-9, 9].
Prints the coefficients and the maximum absolute error of the value and of the derivative."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x = np.linspace(-9, 9, 200001)
ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
dref = 0.5 * (1 + erf(x / np.sqrt(2))) + x * np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def sig(c, x):
    xx = np.minimum(x * x, 50.0)
    with np.errstate(over='ignore'):
        return 1 / (1 + np.exp(-x * (c[0] + c[1] * xx + c[2] * xx * xx))), xx


def loss(c):
    return np.abs(x * sig(c, x)[0] - ref).max()


best = None
rng = np.random.default_rng(0)
for _ in range(8):
    r = minimize(loss, np.array([1.5957691216, 0.0713548163, 0.0]) * (1 + 0.01 * rng.standard_normal(3)), method='Nelder-Mead',
                 options=dict(xatol=1e-11, fatol=1e-13, maxiter=40000, maxfev=80000))
    if best is None or r.fun < best.fun:
        best = r
c = best.x
s, xx = sig(c, x)
d = s + x * s * (1 - s) * (c[0] + 3 * c[1] * xx + 5 * c[2] * xx * xx)
print("c =", list(c), "max |gelu err|", best.fun, "max |dgelu err|", np.abs(d - dref).max())
shipped = [1.5950157685537665, 0.07401129205936302, -0.0007030335796160927]
s, xx = sig(shipped, x)
d = s + x * s * (1 - s) * (shipped[0] + 3 * shipped[1] * xx + 5 * shipped[2] * xx * xx)
print("shipped: max |gelu err|", np.abs(x * s - ref).max(), "max |dgelu err|", np.abs(d - dref).max())
