"""The one-polynomial GELU of csrc/gemm_common.h (round 6): its coefficients are the ones benchmarks/fit_gelu_poly.py derives, and their fp32 Horner / FMA evaluation
emulated in numpy stays within the documented distance of the float64 function (CPU only; the on-device check is tests/test_gpu_dense.py)."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))


def _coefficients_in_header():
    src = open(os.path.join(ROOT, "act_amd", "csrc", "gemm_common.h")).read()
    body = src[src.index("float gelu_half_erfc(float x)"):src.index("return __builtin_amdgcn_exp2f(__builtin_fmaf(-h, t, -1.0f));")]
    first = re.search(r"float h = (-?[0-9.e+-]+)f;", body).group(1)
    rest = re.findall(r"__builtin_fmaf\(h, tc, (-?[0-9.e+-]+)f\)", body)
    return np.array([float(v) for v in [first] + rest][::-1], dtype=np.float32)            # t^0 .. t^8


def test_header_coefficients_are_the_fitted_ones_and_accurate():
    import fit_gelu_poly as F
    from scipy.special import erf
    c = _coefficients_in_header()
    assert len(c) == F.DEG + 1
    fitted = F.fit()
    assert np.allclose(c, fitted, rtol=2e-4, atol=2e-8), (c, fitted)        # (the fit is a float64 least-squares problem: tiny platform differences in the last digits)
    x = np.concatenate([np.linspace(-12, 12, 400001), np.random.default_rng(0).standard_normal(100000) * 1.5]).astype(np.float32); xd = x.astype(np.float64)
    e2 = F.half_erfc(x, c)
    g = F.fma(-np.abs(x), e2, np.maximum(x, np.float32(0))).astype(np.float64)
    assert np.abs(g - 0.5 * xd * (1 + erf(xd / np.sqrt(2)))).max() <= 3e-7
    assert (e2 >= 0).all() and (e2 <= 0.5).all()
