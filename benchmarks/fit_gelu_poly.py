#!/usr/bin/env python3
"""Coefficients and accuracy of the one-polynomial GELU of csrc/gemm_common.h (gelu_half_erfc): h(t) = -log2(erfc(t)) / t on [0, 4] as a degree-8 polynomial,
weighted least squares on Chebyshev nodes with Remez-style re-weighting (weight = sensitivity erfc(t) t of erf to an error in h), then the fp32 Horner / FMA
evaluation emulated in numpy against the float64 function -- next to torch's own fp32 GELU.  CPU only (numpy, scipy, torch).  Dev tool."""
import numpy as np
import torch
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf, erfc

T, DEG = 4.0, 8


def h(t):
    t = np.asarray(t, dtype=np.float64)
    out = np.full_like(t, 2 / np.sqrt(np.pi) / np.log(2))
    big = t >= 1e-8
    out[big] = -np.log2(erfc(t[big])) / t[big]
    return out


def fit():
    n = 4000
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n); t = (x + 1) * T / 2
    w0 = np.maximum(erfc(t) * t, 1e-3); w = w0.copy()
    V = C.chebvander(x, DEG)
    coef = np.linalg.lstsq(V * w[:, None], h(t) * w, rcond=None)[0]
    for _ in range(30):
        r = (V @ coef - h(t)) * w
        w2 = w * (1 + 3 * np.abs(r) / np.abs(r).max())
        coef = np.linalg.lstsq(V * w2[:, None], h(t) * w2, rcond=None)[0]
        w = w2 / w2.max() * w0.max()
    return P.Polynomial(C.cheb2poly(coef))(P.Polynomial([-1, 2 / T])).coef.astype(np.float32)


def f32(v):
    return np.asarray(v, dtype=np.float64).astype(np.float32)


def fma(a, b, c):
    return f32(np.asarray(a, np.float32).astype(np.float64) * np.asarray(b, np.float32).astype(np.float64) + np.asarray(c, np.float32).astype(np.float64))


def half_erfc(x, c):
    t = f32(np.abs(x).astype(np.float64) * np.float64(np.float32(0.70710678118654752440)))
    tc = np.minimum(t, np.float32(T))
    acc = np.full_like(tc, c[-1])
    for ci in c[-2::-1]:
        acc = fma(acc, tc, ci)
    return f32(np.exp2(fma(-acc, t, np.float32(-1.0)).astype(np.float64)))


def main():
    c = fit()
    print("coefficients (t^0 .. t^8):", [float(v) for v in c])
    rng = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-12, 12, 4000001), rng.standard_normal(1000000) * 1.5]).astype(np.float32); xd = x.astype(np.float64)
    ref = 0.5 * xd * (1 + erf(xd / np.sqrt(2)))
    refg = 0.5 * (1 + erf(xd / np.sqrt(2))) + xd * np.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi)
    e2 = half_erfc(x, c)
    g = fma(-np.abs(x), e2, np.maximum(x, np.float32(0)))
    Phi = np.where(x >= 0, f32(1.0 - e2.astype(np.float64)), e2)
    ph = f32(np.exp2(f32(f32(xd * xd).astype(np.float64) * np.float64(np.float32(-0.72134752044448170368))).astype(np.float64)))
    gg = fma(f32(xd * np.float64(np.float32(0.39894228040143267794))), ph, Phi)
    tg = torch.nn.functional.gelu(torch.from_numpy(x)).numpy()
    xt = torch.from_numpy(x).requires_grad_(True); torch.nn.functional.gelu(xt).sum().backward(); tgg = xt.grad.numpy()
    for name, a, r in (("one-polynomial gelu", g, ref), ("torch fp32 gelu", tg, ref), ("one-polynomial gelu'", gg, refg), ("torch fp32 gelu'", tgg, refg)):
        err = np.abs(a.astype(np.float64) - r)
        print(f"{name:22s} max abs err {err.max():.3e} at x = {x[err.argmax()]:8.4f}   max err / max(|ref|, 1e-3) {(err / np.maximum(np.abs(r), 1e-3)).max():.3e}   rms {np.sqrt((err ** 2).mean()):.2e}")


if __name__ == "__main__":
    main()
