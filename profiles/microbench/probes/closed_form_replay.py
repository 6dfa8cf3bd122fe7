"""Truncation error of the closed-form zero-gradient Adam replay (double precision study).
dp = m * sum_i w_i / (s a_i + eps),  w_i = ns_{l+i} b1^i,  a_i = r^i d_{l+i}
closed form: q = 1/(s abar + eps), y = s q, dp ~ m q sum_n (-y)^n N_n,  N_n = sum_i w_i (a_i - abar)^n, N_1 = 0."""
import numpy as np
b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3
r = np.sqrt(b2)

def tables(l, k):
    i = np.arange(1, k + 1, dtype=np.float64)
    t = l + i
    ns = -lr / (1 - b1 ** t)
    d = 1 / np.sqrt(1 - b2 ** t)
    w = ns * b1 ** i
    a = r ** i * d
    return w, a

def err(l, k, nterms):
    w, a = tables(l, k)
    abar = (w * a).sum() / w.sum()
    N = [(w * (a - abar) ** n).sum() for n in range(nterms + 1)]
    s = np.concatenate([[0.0], np.logspace(-12, 2, 400)])
    exact = (w[None, :] / (s[:, None] * a[None, :] + eps)).sum(1)
    q = 1 / (s * abar + eps)
    y = s * q
    approx = q * sum(((-y) ** n) * N[n] for n in range(nterms + 1))
    return np.abs(approx / exact - 1).max()

for l in [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1000, 4000]:
    for k in [2, 8, 32, 150, 1000]:
        print(l, k, " ".join(f"{err(l, k, n):.1e}" for n in [0, 2, 3, 4, 5, 6]))
