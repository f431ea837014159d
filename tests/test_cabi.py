"""CPU: the C-ABI library loads and exports every symbol include/act_hip.h declares (no compute calls)."""
import os
import re
import ctypes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "act_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(act_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "act_amd", "lib", "libact_hip.so"))
    names = _declared()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.act_arch.restype = ctypes.c_char_p
    assert lib.act_arch() == b"gfx950" and lib.act_version() >= 100


def test_binding_declares_every_symbol():
    import act_amd._C as C
    import act_amd.kernels  # noqa: F401  (declares the dense-kernel signatures)
    import act_amd.composite  # noqa: F401  (declares the composite entry points)
    assert set(_declared()) <= set(C.SIGNATURES), sorted(set(_declared()) - set(C.SIGNATURES))


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    import act_amd._C as C
    from act_amd.pointnet2_ops import pointnet2_utils as pu
    from act_amd.knn_cuda import KNN
    with pytest.raises(RuntimeError):
        pu.furthest_point_sample(torch.zeros(1, 16, 3), 4)
    with pytest.raises(RuntimeError):
        KNN(4, True)(torch.zeros(1, 16, 3), torch.zeros(1, 2, 3))
    with pytest.raises(C.ActHipError):
        C.ptr(torch.zeros(3))
