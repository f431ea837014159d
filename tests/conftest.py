import os
import sys
import numpy as np
import pytest

# The GPU suite runs the PRODUCT default (ACT_GEMM_AUTOTUNE=1): shipped table (act_amd/gemm_tune_gfx950.json) first, and for a shape that is not
# listed a first-use timing over candidates that are bit-identical to each other (kernels.stable_candidates: one tile family, one shape-determined
# split-K; tests/test_gpu_dense.py::test_first_use_tuning_cannot_change_a_result_bit).  No result of the suite depends on a stopwatch, listed or not;
# the end of the session only REPORTS the shapes that were tuned (add them to the table to skip their first-use timing).

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """report (never fail on) the GEMM shapes that were tuned on first use: their results are bit-stable by construction (see the note above)"""
    K = sys.modules.get("act_amd.kernels")
    if K is None or os.environ.get("ACT_GEMM_TUNE_SAVE"):
        return
    new = getattr(K, "_NEW_TUNED", {})
    if new and K.AUTOTUNE:
        sys.stderr.write("\n[conftest] %d GEMM shape(s) not in act_amd/gemm_tune_gfx950.json were tuned on first use (bit-stable candidates only): %s\n"
                         % (len(new), sorted(new)[:20]))


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle_c():
    """ctypes handle on the plain-C oracle (built on demand; test infrastructure only)."""
    import ctypes
    import subprocess
    d = os.path.join(ROOT, "oracle")
    so = os.path.join(d, "liboracle_point_ops.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(d, "point_ops_c.c")):
        subprocess.check_call(["make", "-C", d, "-s"])
    return ctypes.CDLL(so)
