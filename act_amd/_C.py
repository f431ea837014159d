"""ctypes binding of libact_hip.so (C ABI declared in include/act_hip.h).

The HIP library is the product: there is NO CPU fallback.  Importing this module
fails loudly if the library has not been built (``python -m act_amd.build`` or
``__graft_entry__.build()``), and every op raises on non-CUDA tensors.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACT_LIB_PATH") or os.path.join(_HERE, "lib", "libact_hip.so")      # (ACT_LIB_PATH: A/B builds of the same ABI)

_vp, _i, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
_f = ctypes.c_float


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP kernels are the product path and have no fallback. "
            "Build them with `python -m act_amd.build` (hipcc --offload-arch=gfx950).")
    return ctypes.CDLL(LIB_PATH)


lib = _load()

# name -> argtypes   (every function returns int unless listed in _RESTYPE)
SIGNATURES = {
    "act_version": [],
    "act_arch": [],
    "act_prof_enable": [_i],
    "act_prof_reset": [],
    "act_prof_num_kernels": [],
    "act_prof_kernel_name": [_i],
    "act_prof_read": [_i, _vp, _vp, _vp, _vp],
    "act_fps_scratch_floats": [_i, _i],
    "act_fps_f32": [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp],
    "act_fps_chain_probe": [_i, _i, _vp, _vp, _vp, _vp],
    "act_knn_group_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "act_gather_points_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "act_gather_points_bwd_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "act_scale_translate_f32": [_vp, _vp, _vp, _i, _i, _vp],
    "act_rotate_points_f32": [_vp, _vp, _i, _i, _vp],
    "act_chamfer_fwd_f32": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "act_chamfer_fwd_ex_f32": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "act_chamfer_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
}
_RESTYPE = {"act_arch": ctypes.c_char_p, "act_prof_kernel_name": ctypes.c_char_p, "act_fps_scratch_floats": ctypes.c_size_t}


def _declare(extra=None):
    sigs = dict(SIGNATURES)
    if extra:
        sigs.update(extra)
    for name, args in sigs.items():
        fn = getattr(lib, name)          # AttributeError here == symbol missing from the build
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, _i)


_declare()


class ActHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise ActHipError(f"{what} failed with code {rc}")


def ptr(t):
    """device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ActHipError("act_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise ActHipError("act_amd kernels need contiguous tensors")
    _same_device(t)
    return ctypes.c_void_p(t.data_ptr())


def _same_device(t):
    """launches go to the CURRENT device's current stream: a tensor living on another GPU would be dereferenced there"""
    if t.device.index != torch.cuda.current_device():
        raise ActHipError(f"tensor on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()} "
                          "(call torch.cuda.set_device(local_rank) first; kernels are launched on the current device's stream)")


def ptr_rows(t):
    """device pointer of a 2-D CUDA tensor with unit inner stride (row-strided views allowed: the callee takes a leading dimension)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ActHipError("act_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")
    if not (t.is_contiguous() or (t.dim() == 2 and t.stride(1) == 1)):
        raise ActHipError("act_amd GEMM operands need unit inner stride")
    _same_device(t)
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle(device_index=None):
    """raw hipStream_t (int) of torch's current stream on the device: the cheap C accessor when this torch build has it
    (torch.cuda.current_stream() costs ~7 us of Python per call, and a step makes ~4,000 of them)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def stream():
    return ctypes.c_void_p(stream_handle())


# ---- profiler helpers -------------------------------------------------------------------------
def prof_enable(on=True):
    return lib.act_prof_enable(1 if on else 0)


def prof_reset():
    lib.act_prof_reset()


def prof_table():
    """{kernel: dict(ms, launches, flops, bytes)} for kernels launched since the last reset."""
    out = {}
    for k in range(lib.act_prof_num_kernels()):
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = ctypes.c_longlong()
        lib.act_prof_read(k, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        if n.value:
            out[lib.act_prof_kernel_name(k).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out
