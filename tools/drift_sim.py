"""Numerical check (host, numpy) of the table-driven closed form the kernels use for the steps TF's dense
Adam applies to a row that is absent from G consecutive batches:

    p_G = p - m * sum_{i=1..G} alpha_{s+i} b1^i / (sqrt(v) g^i + eps),   g = sqrt(b2)
        = p - m * sum_i c_i / (sqrt(v) + eps g^-i),                      c_i = alpha_{s+i} (b1/g)^i

eps g^-i moves by 1 % over the steps that carry weight, so 1/(sqrt(v) + e_bar + d_i) is expanded to second
order in d_i = eps (g^-i - g^-c) around e_bar = eps g^-c:  y = 1/(sqrt(v) + e_bar),
    sum = y C0 - y^2 C1 + y^3 C2,   C_k = sum_i c_i d_i^k
and (C0, C1, C2) depend on (s, G) only: a 256-entry table per optimizer step for G < 256, and for G >= 256
(where the tail beyond 255 steps weighs 0.9^255) a table over s built once, constant beyond s = 16384.

Compares, in fp32 arithmetic for the closed form, against the sequential update in float64.
"""
import numpy as np

b1, b2, eps = 0.9, 0.999, 1e-8
C = 8
GT = 256
SMAX = 16384


def alpha(lr, j):
    return lr * np.sqrt(1 - b2 ** j) / (1 - b1 ** j)


def exact(p, m, v, s, G, lr):
    for k in range(1, G + 1):
        m = m * b1
        v = v * b2
        p = p - alpha(lr, s + k) * m / (np.sqrt(v) + eps)
    return p


def coeffs(s, G, lr):
    """float32 (C0, C1, C2) as the table kernel forms them (sums in fp32, terms from fp32 powers)."""
    g = np.float32(np.sqrt(np.float32(b2)))
    n = min(G, GT - 1)
    i = np.arange(1, n + 1, dtype=np.float64)
    a = np.array([alpha(lr, s + int(k)) for k in i]).astype(np.float32)
    q0 = (np.float32(b1) / g) ** i.astype(np.float32)
    gi = g ** (-i.astype(np.float32))
    d = (gi - g ** np.float32(-C)).astype(np.float32) * np.float32(eps)
    c = (a * q0).astype(np.float32)
    return (np.float32(c.sum(dtype=np.float32)), np.float32((c * d).sum(dtype=np.float32)),
            np.float32((c * d * d).sum(dtype=np.float32)))


def closed(p, m, v, s, G, lr):
    if s >= SMAX and G >= GT:
        s = SMAX  # the constant-alpha entry
    c0, c1, c2 = coeffs(s, G, lr)
    g = np.float32(np.sqrt(np.float32(b2)))
    ebar = np.float32(eps) * g ** np.float32(-C)
    y = np.float32(1.0) / (np.sqrt(np.float32(v)) + ebar)
    drift = np.float32(m) * y * (c0 - y * (c1 - y * c2))
    return np.float32(p) - drift


if __name__ == "__main__":
    for lr in (0.002, 0.05):
        worst_abs, worst_rel, arg = 0.0, 0.0, None
        for s in (1, 2, 5, 10, 30, 63, 100, 1000, 5000, 16383, 16384, 20000, 200000):
            for G in (1, 2, 5, 10, 24, 25, 45, 90, 200, 255, 256, 300, 5000):
                for gmag in (1e-2, 1e-5, 1e-7, 3e-8, 1e-8, 1e-9, 1e-12):
                    for ratio, vf in ((0.1, 0.001), (1.0, 1.0), (-1.0, 1.0)):
                        m0, v0 = ratio * gmag, vf * gmag * gmag
                        pe = exact(0.25, m0, v0, s, G, lr)
                        pa = float(closed(0.25, m0, v0, s, G, lr))
                        err = abs(pe - pa)
                        rel = err / max(abs(pe - 0.25), 1e-30)
                        if err > worst_abs:
                            worst_abs, arg = err, (s, G, gmag, ratio)
                        if abs(pe - 0.25) > 1e-6:
                            worst_rel = max(worst_rel, rel)
        print("lr %g: worst |error| %.3e at (s, G, |g|, m/|g|) = %s; worst error / drift %.3e "
              "(fp32 ulp of p = 0.25: 3e-8)" % (lr, worst_abs, arg, worst_rel))
    # split invariance: 30 steps at once against 20 then 10 (m, v decayed in between)
    lr = 0.05
    m0, v0, s = 3e-6, 2e-11, 500
    one = float(closed(0.25, m0, v0, s, 30, lr))
    mid = float(closed(0.25, m0, v0, s, 20, lr))
    two = float(closed(mid, m0 * b1 ** 20, v0 * b2 ** 20, s + 20, 10, lr))
    print("split 30 = 20 + 10 at lr 0.05: %.3e apart (drift %.3e)" % (abs(one - two), abs(one - 0.25)))
