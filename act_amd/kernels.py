"""Host-side plumbing over the C ABI: raw launch wrappers + the autograd Functions the model mirror uses.

Nothing here computes on the host: every function allocates outputs with torch (device memory, caching
allocator) and launches kernels of libact_hip.so on the current HIP stream.
"""
import ctypes
import os

import torch

from . import _C

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

EPI_NONE, EPI_GELU, EPI_RELU, EPI_MUL_GELU_GRAD, EPI_MUL_RELU_MASK = 0, 1, 2, 3, 4


class GemmEpilogue(ctypes.Structure):
    _fields_ = [("alpha", _f), ("act", _i), ("accumulate", _i), ("rows_per_scale", _i), ("ldr", _i), ("ldaux", _i),
                ("res_row_div", _i), ("bias", _vp), ("rowscale", _vp), ("res", _vp), ("aux", _vp)]


_C._declare({
    "act_sgemm_f32": [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, ctypes.POINTER(GemmEpilogue), _vp, _sz, _vp],
    "act_sgemm_ex_f32": [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, ctypes.POINTER(GemmEpilogue), _vp, _sz, _i, _i, _vp],
    "act_layernorm_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "act_layernorm_bwd_workspace": [_i, _i],
    "act_layernorm_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _vp],
    "act_colsum_workspace": [_i, _i],
    "act_colsum_f32": [_vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp],
    "act_attention_fwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "act_attention_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "act_cosine_loss_fwd_f32": [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp],
    "act_cosine_loss_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp],
    "act_regression_loss_fwd_f32": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "act_regression_loss_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "act_softmax_xent_fwd_f32": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "act_softmax_xent_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp],
})
_C.lib.act_layernorm_bwd_workspace.restype = _sz
_C.lib.act_colsum_workspace.restype = _sz
for _n in ("act_sgemm_f32", "act_sgemm_ex_f32", "act_layernorm_fwd_f32", "act_layernorm_bwd_workspace", "act_layernorm_bwd_f32",
           "act_colsum_workspace", "act_colsum_f32", "act_attention_fwd_f32", "act_attention_bwd_f32",
           "act_cosine_loss_fwd_f32", "act_cosine_loss_bwd_f32", "act_softmax_xent_fwd_f32", "act_softmax_xent_bwd_f32",
           "act_regression_loss_fwd_f32", "act_regression_loss_bwd_f32"):
    _C.SIGNATURES.setdefault(_n, getattr(_C.lib, _n).argtypes)

lib, ptr, stream, check = _C.lib, _C.ptr, _C.stream, _C.check


class GemmTnProblem(ctypes.Structure):                 # act_gemm_tn_problem_t
    _fields_ = [("A", _vp), ("lda", _i), ("B", _vp), ("ldb", _i), ("C", _vp), ("ldc", _i), ("M", _i), ("N", _i), ("bias_out", _vp)]


_C._declare({"act_sgemm_tn_grouped_workspace": [ctypes.POINTER(GemmTnProblem), _i, _i, _i],
             "act_sgemm_tn_grouped_splits": [ctypes.POINTER(GemmTnProblem), _i, _i],
             "act_sgemm_tn_grouped_f32": [ctypes.POINTER(GemmTnProblem), _i, _i, _i, _vp, _sz, _vp]})
_C.lib.act_sgemm_tn_grouped_workspace.restype = _sz
for _n in ("act_sgemm_tn_grouped_workspace", "act_sgemm_tn_grouped_splits", "act_sgemm_tn_grouped_f32"):
    _C.SIGNATURES.setdefault(_n, getattr(_C.lib, _n).argtypes)

# ---- persistent scratch (split-K partials, LN / colsum partial rows): one buffer per device ----------
_WS = {}
_WS_BYTES = 160 << 20       # split-K / grouped-GEMM partials: 7 K ranges of the two d=768 MLP weight gradients need 132 MB (round 3; 64 MB before)


def workspace(device, nbytes=_WS_BYTES):
    """scratch buffer of the CURRENT stream on ``device`` (kernels of different streams may run concurrently)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _C.stream_handle(idx))
    w = _WS.get(key)
    if w is None or w.numel() * 4 < nbytes:
        w = torch.empty(max(nbytes, _WS_BYTES) // 4, dtype=torch.float32, device=device)
        _WS[key] = w
    return w


_SIDE = {}


def side_stream(device, which=0):
    """auxiliary HIP streams per device for work that is independent of the main chain: 0 = frozen teacher forward (may run a
    whole step ahead), 1 = weight-gradient GEMMs of the student's backward (forked and joined inside one block's backward)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), which)
    st = _SIDE.get(key)
    if st is None:
        prio = int(os.environ.get("ACT_SIDE_PRIO%d" % which, "0"))      # experiment knob: hipStream priority of the auxiliary streams
        st = _SIDE[key] = torch.cuda.Stream(device=device, priority=prio)
    return st


OVERLAP_DW = os.environ.get("ACT_OVERLAP_DW", "1") != "0"
LINEAR_OVERLAP_DW = os.environ.get("ACT_LINEAR_OVERLAP_DW", "0") == "1"     # LinearFn.backward: dW / db on the auxiliary stream (measured per workload; see DESIGN)
GROUPED_DW = os.environ.get("ACT_GROUPED_DW", "1") != "0"      # weight + bias gradients of two Linears per launch (0: one GEMM + column sum each)


class fork_side:
    """``with fork_side(dev):`` enqueues the body on the auxiliary stream, ordered after everything already enqueued on the
    current stream.  Used for weight-gradient GEMMs / bias column sums, which nothing in the rest of the backward chain reads:
    they then share the chip with the (small, latency-bound) dX GEMMs of the student instead of running back to back."""

    def __init__(self, device):
        self.main, self.side = torch.cuda.current_stream(device), side_stream(device, 1)

    def __enter__(self):
        self.side.wait_stream(self.main)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def join_side(device, *tensors):
    """the current stream waits for the auxiliary stream; ``tensors`` (allocated there) are handed over to the current stream."""
    main = torch.cuda.current_stream(device)
    main.wait_stream(side_stream(device, 1))
    for t in tensors:
        if t is not None:
            t.record_stream(main)


def _f32c(t, name="tensor"):
    if t.dtype != torch.float32:
        raise _C.ActHipError(f"{name}: expected float32")
    return t if t.is_contiguous() else t.contiguous()


def _f32rows(t, name="tensor"):
    """2-D operand with unit inner stride (row-strided views such as column slices of a weight are passed as is: the GEMM
    takes a leading dimension)."""
    if t.dtype != torch.float32:
        raise _C.ActHipError(f"{name}: expected float32")
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and (
            t.is_contiguous() or (t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0)):
        return t                              # strided views must keep the float4 path of the tuned kernels available
    return t.contiguous()


# ---- raw wrappers ---------------------------------------------------------------------------------------
def gemm(a, b, a_kmajor=True, b_kmajor=True, bias=None, act=EPI_NONE, aux=None, res=None, rowscale=None,
         rows_per_scale=0, out=None, accumulate=False, alpha=1.0, res_row_div=0, cfg=None):
    """C[M,N] = epilogue(op(a) @ op(b)); a: [M,K] if a_kmajor else [K,M]; b: [N,K] if b_kmajor else [K,N]."""
    a = _f32rows(a, "a"); b = _f32rows(b, "b")
    if a_kmajor:
        M, K = a.shape
    else:
        K, M = a.shape
    if b_kmajor:
        N, Kb = b.shape
    else:
        Kb, N = b.shape
    if K != Kb:
        raise _C.ActHipError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    e = GemmEpilogue(alpha=alpha, act=act, accumulate=int(accumulate), rows_per_scale=int(rows_per_scale),
                     ldr=(res.stride(0) if res is not None else 0), ldaux=(aux.stride(0) if aux is not None else 0),
                     res_row_div=int(res_row_div), bias=ptr(bias), rowscale=ptr(rowscale), res=ptr(res), aux=ptr(aux))
    ws = workspace(a.device)
    tile, splits = cfg if cfg is not None else _gemm_config(a, b, a_kmajor, b_kmajor, M, N, K, ws)
    check(lib.act_sgemm_ex_f32(int(a_kmajor), int(b_kmajor), M, N, K, _C.ptr_rows(a), a.stride(0), _C.ptr_rows(b), b.stride(0), ptr(out),
                               out.stride(0), ctypes.byref(e), ptr(ws), ws.numel() * 4, tile, splits, stream()), "act_sgemm_f32")
    return out


def gemm_tn_grouped(pairs, want_bias=True, splits=0):
    """[(dy [T,M], x [T,N]), ...] -> ([dW_p = dy_p^T . x_p  [M,N]], [db_p = column sums of dy_p  [M]]) in ONE launch (+ one reduction launch
    for the K ranges): the weight / bias gradients of several Linears that share the token dimension (csrc/gemm_grouped.hip).
    splits <= 0: the library's count, capped by the shared workspace (_WS_BYTES) exactly as the composite block backward caps it, so the
    per-kernel and the composite host paths fold the same K ranges in the same order (bit-identical) for any model width."""
    if not pairs or len(pairs) > 8:
        raise _C.ActHipError(f"gemm_tn_grouped: 1..8 problems per launch, got {len(pairs)}")
    pairs = [(_f32rows(dy, "dy"), _f32rows(x, "x")) for dy, x in pairs]      # fp32, unit inner stride, float4-able rows (else a contiguous copy)
    T = pairs[0][0].shape[0]
    dev = pairs[0][0].device
    probs = (GemmTnProblem * len(pairs))()
    dws, dbs = [], []
    for i, (dy, x) in enumerate(pairs):
        if dy.dim() != 2 or x.dim() != 2 or dy.shape[0] != T or x.shape[0] != T:
            raise _C.ActHipError("gemm_tn_grouped: every operand needs the same number of rows")
        M, N = dy.shape[1], x.shape[1]
        dw = torch.empty(M, N, dtype=torch.float32, device=dev)
        db = torch.empty(M, dtype=torch.float32, device=dev) if want_bias else None
        dws.append(dw); dbs.append(db)
        probs[i] = GemmTnProblem(_C.ptr_rows(dy).value, dy.stride(0), _C.ptr_rows(x).value, x.stride(0), dw.data_ptr(), N, M, N,
                                 db.data_ptr() if db is not None else None)
    ws = workspace(dev)
    if splits <= 0:
        splits = lib.act_sgemm_tn_grouped_splits(probs, len(pairs), T)
        while splits > 1 and lib.act_sgemm_tn_grouped_workspace(probs, len(pairs), T, splits) > ws.numel() * 4:
            splits -= 1
    else:
        ws = workspace(dev, lib.act_sgemm_tn_grouped_workspace(probs, len(pairs), T, splits))      # an explicit count gets the space it needs
    check(lib.act_sgemm_tn_grouped_f32(probs, len(pairs), T, splits, ptr(ws), ws.numel() * 4, stream()), "act_sgemm_tn_grouped_f32")
    return dws, dbs


# ---- OPT-IN split-bf16 products (csrc/gemm_bf16x3.hip): frozen teacher only, never the default -------------------------------------------------
_C._declare({"act_split_bf16x2_f32": [_vp, _i, _i, _i, _vp, _vp, _vp],
             "act_sgemm_nt_bf16x3_supported": [_i, _i, _i],
             "act_sgemm_nt_bf16x3_f32": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, ctypes.POINTER(GemmEpilogue), _vp],
             "act_sgemm_nt_bf16x3_planes_f32": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, ctypes.POINTER(GemmEpilogue), _vp],
             # producers that emit planes (used from C by the teacher composite; declared here so the binding covers the whole header)
             "act_layernorm_fwd_planes_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
             "act_prompt_layernorm_fwd_planes_f32": [_vp, _vp, _i, _i, _i, _f, ctypes.c_uint64, _vp, _vp, _vp, _f, _vp, _vp, _vp],
             "act_attention_fwd_prefix_planes_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]})
for _n in ("act_split_bf16x2_f32", "act_sgemm_nt_bf16x3_supported", "act_sgemm_nt_bf16x3_f32", "act_sgemm_nt_bf16x3_planes_f32",
           "act_layernorm_fwd_planes_f32", "act_prompt_layernorm_fwd_planes_f32", "act_attention_fwd_prefix_planes_f32"):
    _C.SIGNATURES.setdefault(_n, getattr(_C.lib, _n).argtypes)


def split_bf16x2(x, out=None):
    """fp32 [R, K] -> bf16 planes [2, R, K]: hi = bf16(x), lo = bf16(x - hi) (round to nearest even) -- the operand form of gemm_nt_bf16x3."""
    x = _f32rows(x, "x")
    R, Kd = x.shape
    if out is None:
        out = torch.empty(2, R, Kd, dtype=torch.bfloat16, device=x.device)
    check(lib.act_split_bf16x2_f32(_C.ptr_rows(x), R, Kd, x.stride(0), ptr(out[0]), ptr(out[1]), stream()), "act_split_bf16x2_f32")
    return out


def layernorm_planes(x, gamma, beta, eps, pos=None):
    """LN(x + pos) * gamma + beta written as (hi, lo) bf16 planes [2, T, D] (bit-identical to split_bf16x2(layernorm_fwd(...)))"""
    x = _f32c(x)
    T, D = x.shape
    planes = torch.empty(2, T, D, dtype=torch.bfloat16, device=x.device)
    check(lib.act_layernorm_fwd_planes_f32(ptr(x), ptr(_f32c(pos)) if pos is not None else None, ptr(gamma), ptr(beta), None, None, ptr(planes[0]), ptr(planes[1]),
                                           None, None, T, D, float(eps), stream()), "act_layernorm_fwd_planes_f32")
    return planes


def gemm_nt_bf16x3(a_planes, b_planes, bias=None, act=EPI_NONE, res=None, out=None, planes_out=None):
    """C[M,N] = epilogue(a . b^T) with both operands given as (hi, lo) bf16 planes [2, rows, K] (split_bf16x2) and the three products hi.hi + hi.lo + lo.hi
    on the bf16 matrix cores (fp32 accumulation): 4e-6 relative per product.  M, N % 128 == 0, K % 64 == 0.  Opt-in path of the frozen teacher."""
    _, M, Kd = a_planes.shape
    _, N, Kb = b_planes.shape
    if Kd != Kb or a_planes.dtype != torch.bfloat16 or b_planes.dtype != torch.bfloat16 or not (a_planes.is_contiguous() and b_planes.is_contiguous()):
        raise _C.ActHipError("gemm_nt_bf16x3: contiguous bf16 planes [2, rows, K] with equal K expected")
    if not lib.act_sgemm_nt_bf16x3_supported(M, N, Kd):
        raise _C.ActHipError(f"gemm_nt_bf16x3: unsupported shape {M} x {N} x {Kd} (M, N % 128, K % 64)")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a_planes.device)
    e = GemmEpilogue(alpha=1.0, act=act, accumulate=0, rows_per_scale=0, ldr=(res.stride(0) if res is not None else 0), ldaux=0, res_row_div=0,
                     bias=ptr(bias), rowscale=None, res=ptr(res), aux=None)
    if planes_out is not None:                               # the result also as (hi, lo) planes [2, M, N] -- the A operand of a following product
        if planes_out.dtype != torch.bfloat16 or tuple(planes_out.shape) != (2, M, N) or not planes_out.is_contiguous():
            raise _C.ActHipError("gemm_nt_bf16x3: planes_out must be a contiguous bf16 tensor [2, M, N]")
        check(lib.act_sgemm_nt_bf16x3_planes_f32(M, N, Kd, ptr(a_planes[0]), ptr(a_planes[1]), ptr(b_planes[0]), ptr(b_planes[1]), ptr(out), out.stride(0),
                                                 ptr(planes_out[0]), ptr(planes_out[1]), ctypes.byref(e), stream()), "act_sgemm_nt_bf16x3_planes_f32")
        return out
    check(lib.act_sgemm_nt_bf16x3_f32(M, N, Kd, ptr(a_planes[0]), ptr(a_planes[1]), ptr(b_planes[0]), ptr(b_planes[1]), ptr(out), out.stride(0),
                                      ctypes.byref(e), stream()), "act_sgemm_nt_bf16x3_f32")
    return out


# ---- GEMM autotuner: the step has ~40 distinct (layout, M, N, K) shapes; each is timed once (tile shape x split-K) on first
# use -- i.e. during the warm-up steps -- and the winner is cached, so steady-state steps never synchronise.
_GEMM_CACHE = {}
AUTOTUNE = os.environ.get("ACT_GEMM_AUTOTUNE", "1") != "0"
# ACT_GEMM_AUTOTUNE=1 (default): an unlisted shape is timed on first use over candidates that are BIT-IDENTICAL to each other (stable_candidates:
# one tile family, one deterministic split-K), so which of them the stopwatch prefers never changes a result bit.  "full": every tile family x split-K
# (a different split-K is a different fp32 summation order: results may differ in the last bits between runs) -- what benchmarks/tune_table.py
# uses to BUILD the shipped table, whose entries are then fixed.  "0": shipped table + built-in cost model only.
AUTOTUNE_FULL = os.environ.get("ACT_GEMM_AUTOTUNE", "1") == "full"
# Shipped winners for the shapes of the benchmarked workloads (measured on one MI355X by this same autotuner and dumped with
# ACT_GEMM_TUNE_SAVE=<file>): first use of a listed shape costs nothing; unlisted shapes are still tuned on first use.
_TUNE_FILE = os.environ.get("ACT_GEMM_TUNE_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tune_gfx950.json")
_GEMM_TABLE = {}
_MAX_SPLIT = int(os.environ.get("ACT_GEMM_MAX_SPLIT", "0"))              # experiment knob: cap split-K (table entries above the cap are re-tuned on first use)
if os.environ.get("ACT_GEMM_TUNE_TABLE", "1") != "0" and os.path.exists(_TUNE_FILE):
    import json as _json
    with open(_TUNE_FILE) as _fh:
        _GEMM_TABLE = {tuple(int(v) for v in k.split(",")): tuple(c) for k, c in _json.load(_fh)["configs"].items()}
    if _MAX_SPLIT > 0:
        _GEMM_TABLE = {k: c for k, c in _GEMM_TABLE.items() if c[1] <= _MAX_SPLIT or k[4] > 8192}
_NEW_TUNED = {}
if os.environ.get("ACT_GEMM_TUNE_SAVE"):
    import atexit as _atexit

    def _dump_tuned(path=os.environ["ACT_GEMM_TUNE_SAVE"].replace("%p", str(os.getpid()))):      # "%p" -> pid: one file per process of a test session
        import json
        merged = dict(_GEMM_TABLE); merged.update(_NEW_TUNED)
        with open(path, "w") as f:
            json.dump({"arch": "gfx950", "key": "a_kmajor,b_kmajor,M,N,K", "value": "[tile id, split-K]",
                       "configs": {",".join(str(v) for v in k): list(c) for k, c in sorted(merged.items())}}, f, indent=0)
    _atexit.register(_dump_tuned)


def stable_split(M, N, K, ws_bytes):
    """split-K of the first-use tuner: a function of the shape only.  No split while the 64 x 64 tile grid alone gives every CU work or K is short;
    else about two workgroups per CU, K ranges of >= 256, at most 8 ranges, partial sums within the workspace."""
    nb = -(-M // 64) * -(-N // 64)
    if nb >= 384 or K < 1024:
        return 1
    s = max(1, min(8, K // 256, int(round(512.0 / nb))))
    while s > 1 and s * M * N * 4 > ws_bytes:
        s -= 1
    return s


def stable_candidates(a, b, ak, bk, M, N, K, ws):
    """(tile id, split-K) configurations of ONE tile family at ONE deterministic split-K: every candidate adds the same fp32 products in the same order
    per output element (tests/test_gpu_dense.py: tiles 30 / 31 / 32 / 10 / 11 / 12 / 20 / 21 for NT, the quad-fragment tiles 13..16 for NN), so the
    timing-based choice between them cannot change a bit of the result.  [] = no fast family applies: built-in cost model, no timing at all."""
    aligned = ((a.data_ptr() | b.data_ptr()) & 15) == 0 and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0
    if not aligned or K % 32 != 0:
        return []
    sp = stable_split(M, N, K, ws.numel() * 4)             # (the library rounds the K range up to a multiple of 32 and recounts the ranges: same for every tile)
    if ak and bk:                                             # NT: hand-scheduled loop first, compiler loop as the alternative; M tails allowed
        fam = [(30, 128), (31, 64), (32, 64), (10, 128), (11, 64), (12, 64)]
        return [(t, sp) for t, bn in fam if N % bn == 0]
    if ak and not bk:                                         # NN: quad-fragment tiles (13: 128x128, 14: 64x128, 16: 128x64, 15: 64x64)
        return [(t, sp) for t, bn in ((33, 128), (34, 128), (36, 64), (35, 64), (13, 128), (14, 128), (16, 64), (15, 64)) if N % bn == 0]
    if not ak and not bk and M % 128 == 0 and N % 128 == 0:   # TN: the quad-fragment tile, hand-scheduled or compiler-scheduled
        return [(33, sp), (13, sp)]
    return []


def first_use_config(a, b, ak, bk, M, N, K, ws):
    """the launch configuration of a shape that is in no table: (tile id, split-K)"""
    if AUTOTUNE_FULL:
        return gemm_tune(a, b, ak, bk, M, N, K, ws)[0]
    cands = stable_candidates(a, b, ak, bk, M, N, K, ws)
    if not cands:
        return 0, 0
    if len(cands) == 1:
        return cands[0]
    best, t = gemm_tune(a, b, ak, bk, M, N, K, ws, cands=cands)
    return best if t < float("inf") else (0, 0)


def gemm_tune(a, b, ak, bk, M, N, K, ws, reps=3, rounds=1, trace=None, cands=None):
    """time every (tile id, split-K) candidate for this product (``cands``: only these) -> (best config, best ms per launch); ``trace`` (a list)
    receives every (tile, splits, ms) measured."""
    given = cands
    cands = []
    for tile, (bm, bn) in ((1, (128, 128)), (2, (128, 64)), (3, (64, 64))):
        nb = -(-M // bm) * -(-N // bn)
        if nb > 16384 and tile > 1:
            continue
        sp_list = [1]
        if K >= 1024 and nb < 2048:
            sp_list += [s for s in (2, 3, 4, 6, 8, 12, 16, 24, 32) if K // s >= 256 and s * M * N * 4 <= ws.numel() * 4 and nb * s <= 8192]
        # prune hopeless configurations (a long serial K loop on a handful of workgroups takes tens of ms per trial)
        sp_list = [s for s in sp_list if not (K // s > 8192 and nb * s < 256) or s == sp_list[-1]]
        if _MAX_SPLIT > 0 and K <= 8192:
            sp_list = [s for s in sp_list if s <= _MAX_SPLIT]
        cands += [(tile, s) for s in sp_list]
        if M % bm == 0 and N % bn == 0 and K % 32 == 0:
            cands += [(tile + 3, s) for s in sp_list if s <= 4 and K % 32 == 0]       # software-pipelined main loop
        if (M % bm == 0 or ak) and N % bn == 0 and K % 32 == 0:                              # the 16x16x4 kernels take an M tail (K-major A)
            cands += [(tile + 6, s) for s in sp_list if K % 32 == 0]                  # v_mfma_f32_16x16x4_f32 main loop
            if ak and bk:
                cands += [(tile + 9, s) for s in sp_list if K % 32 == 0]              # NT: K-contiguous LDS image, b128 fragments
                cands += [(29 + tile, s) for s in sp_list if K % 32 == 0]             # NT: the hand-scheduled main loop (30: 128x128, 31: 128x64, 32: 64x64)
                if tile in (1, 2) and M % bm == 0:                                            # ... with 32-deep K tiles (20: 128x128, 21: 128x64): full 128-byte rows per load
                    cands += [(19 + tile, s) for s in sp_list if K % 32 == 0]
                if tile == 1 and M % bm == 0 and K >= 1536:                                   # ... with the software-pipelined main loop (17: 128x128; 18 = 128x64
                    cands += [(17, s) for s in sp_list if K % 32 == 0]                 # exists but never won a shape): pays on long K only
        if not bk and N % 128 == 0 and K % 32 == 0 and ((tile == 1 and (ak or M % 128 == 0)) or (tile == 2 and ak)):
            qsp = [1]
            nbq = -(-M // (128 if tile == 1 else 64)) * (N // 128)
            if K >= 1024 and nbq < 2048:
                qsp += [s for s in (2, 3, 4, 6, 8, 12, 16, 24, 32) if K // s >= 256 and s * M * N * 4 <= ws.numel() * 4 and nbq * s <= 8192]
            if K >= 65536 and nbq <= 64:                              # weight gradients over a few hundred thousand rows on a handful of tiles
                qsp += [s for s in (48, 64, 96, 128, 192, 256) if s * M * N * 4 <= ws.numel() * 4 and nbq * s <= 1024]
            if K >= 512 and nbq < 128:                                # a handful of tiles: K ranges down to 128 rows, up to one round of 512 workgroups
                qsp += [s for s in (5, 7, 9, 10, 12, 14, 16) if s not in qsp and K // s >= 128 and s * M * N * 4 <= ws.numel() * 4 and nbq * s <= 512]
            if _MAX_SPLIT > 0 and K <= 8192:
                qsp = [s for s in qsp if s <= _MAX_SPLIT]
            cands += [(12 + tile, s) for s in qsp if K % 32 == 0]                     # NN / TN: quad fragments (13: 128x128, 14: 64x128)
            cands += [(32 + tile, s) for s in qsp if K % 32 == 0]                     # ... on the hand-scheduled main loop (33: 128x128, 34: 64x128)
        if ak and not bk and N % 64 == 0 and K % 32 == 0 and tile in (2, 3):                  # NN, 64-column quad tiles (4 x 1 waves): 16 = 128x64, 15 = 64x64
            qsp = [1]
            nbq = -(-M // (128 if tile == 2 else 64)) * (N // 64)
            if K >= 1024 and nbq < 2048:
                qsp += [s for s in (2, 3, 4, 6, 8, 12, 16, 24, 32) if K // s >= 256 and s * M * N * 4 <= ws.numel() * 4 and nbq * s <= 8192]
            if K >= 65536 and nbq <= 64:                              # weight gradients over a few hundred thousand rows on a handful of tiles
                qsp += [s for s in (48, 64, 96, 128, 192, 256) if s * M * N * 4 <= ws.numel() * 4 and nbq * s <= 1024]
            if K >= 512 and nbq < 128:                                # a handful of tiles: K ranges down to 128 rows, up to one round of 512 workgroups
                qsp += [s for s in (5, 7, 9, 10, 12, 14, 16) if s not in qsp and K // s >= 128 and s * M * N * 4 <= ws.numel() * 4 and nbq * s <= 512]
            if _MAX_SPLIT > 0 and K <= 8192:
                qsp = [s for s in qsp if s <= _MAX_SPLIT]
            cands += [(16 if tile == 2 else 15, s) for s in qsp if K % 32 == 0]
            cands += [(36 if tile == 2 else 35, s) for s in qsp if K % 32 == 0]       # ... hand-scheduled (36: 128x64, 35: 64x64)
    if given is not None:
        cands = list(given)
    scratch = torch.empty(M, N, dtype=torch.float32, device=a.device)
    e = GemmEpilogue(alpha=1.0)
    best, best_t = (0, 0), float("inf")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for tile, sp in cands:
        def run():
            return lib.act_sgemm_ex_f32(int(ak), int(bk), M, N, K, _C.ptr_rows(a), a.stride(0), _C.ptr_rows(b), b.stride(0), ptr(scratch), N,
                                        ctypes.byref(e), ptr(ws), ws.numel() * 4, tile, sp, stream())
        if run() != 0:
            continue
        t = float("inf")
        for _ in range(rounds):
            ev[0].record()
            for _ in range(reps):
                run()
            ev[1].record()
            ev[1].synchronize()
            t = min(t, ev[0].elapsed_time(ev[1]) / reps)
        if trace is not None:
            trace.append((tile, sp, t))
        if t < best_t:
            best, best_t = (tile, sp), t
    return best, best_t


def _gemm_config(a, b, ak, bk, M, N, K, ws):
    if not AUTOTUNE or M * N * K < (1 << 24):
        return 0, 0                                            # tiny products: built-in cost model
    if not ak and not bk and min(M, N) <= 8:
        return 0, 0                                            # skinny weight gradients: streaming-reduction kernel (gemm.hip)
    key = (int(ak), int(bk), M, N, K, a.device.index)
    cfg = _GEMM_CACHE.get(key)
    if cfg is not None:
        return cfg
    cfg = _GEMM_TABLE.get(key[:5])
    if cfg is not None:
        _GEMM_CACHE[key] = cfg
        return cfg
    if torch.cuda.is_current_stream_capturing():
        return 0, 0
    best = first_use_config(a, b, ak, bk, M, N, K, ws)
    _GEMM_CACHE[key] = best
    _NEW_TUNED[key[:5]] = best
    lib.act_gemm_tune_set(int(ak), int(bk), M, N, K, int(best[0]), int(best[1]))      # the composite entry points launch the same configuration
    return best


def layernorm_fwd(x, pos, gamma, beta, eps, want_xin=True, want_stats=True):
    x = _f32c(x)
    T, D = x.shape
    pos = _f32c(pos) if pos is not None else None
    y = torch.empty_like(x)
    xin = torch.empty_like(x) if (pos is not None and want_xin) else None
    mean = torch.empty(T, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(T, dtype=torch.float32, device=x.device) if want_stats else None
    check(lib.act_layernorm_fwd_f32(ptr(x), ptr(pos), ptr(gamma), ptr(beta), ptr(xin), ptr(y), ptr(mean), ptr(rstd), T, D,
                                    float(eps), stream()), "act_layernorm_fwd_f32")
    return y, (xin if xin is not None else x), mean, rstd


def layernorm_bwd(dy, xin, gamma, mean, rstd, dres=None, want_params=True):
    dy = _f32c(dy)
    T, D = dy.shape
    dx = torch.empty_like(dy)
    dg = torch.empty(D, dtype=torch.float32, device=dy.device) if want_params else None
    db = torch.empty(D, dtype=torch.float32, device=dy.device) if want_params else None
    ws = workspace(dy.device, lib.act_layernorm_bwd_workspace(T, D)) if want_params else None
    check(lib.act_layernorm_bwd_f32(ptr(dy), ptr(xin), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg), ptr(db), 0,
                                    ptr(ws), (ws.numel() * 4 if ws is not None else 0), T, D, stream()), "act_layernorm_bwd_f32")
    return dx, dg, db


def colsum(x):
    x = _f32c(x)
    R, C = x.shape
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(x.device, lib.act_colsum_workspace(R, C))
    check(lib.act_colsum_f32(ptr(x), R, C, x.stride(0), ptr(out), 0, ptr(ws), ws.numel() * 4, stream()), "act_colsum_f32")
    return out


def attention_fwd(qkv, B, S, H, hd, want_lse=True):
    out = torch.empty(B * S, H * hd, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=qkv.device) if want_lse else None
    check(lib.act_attention_fwd_f32(ptr(qkv), ptr(out), ptr(lse), B, S, H, hd, float(hd) ** -0.5, stream()), "act_attention_fwd_f32")
    return out, lse


def attention_bwd(qkv, out, dout, lse, B, S, H, hd):
    dqkv = torch.empty_like(qkv)
    check(lib.act_attention_bwd_f32(ptr(qkv), ptr(out), ptr(_f32c(dout)), ptr(lse), ptr(dqkv), B, S, H, hd, float(hd) ** -0.5,
                                    stream()), "act_attention_bwd_f32")
    return dqkv


# ---- autograd Functions -----------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x @ w^T + b on [T,in] rows (nn.Linear / Conv1d k=1)."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = _f32c(x).reshape(-1, shp[-1])
        y = gemm(x2, w, True, True, bias=b)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        ctx.shp = shp
        return y.reshape(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = _f32c(dy).reshape(-1, w.shape[0])
        want_dw, want_db = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        # LINEAR_OVERLAP_DW (Stage-I autoencoder step, runner_autoencoder.train_step): dW / db on auxiliary stream 1 while the input gradient
        # runs on the main stream -- two chip-filling GEMMs side by side fill each other's partial last round of workgroups; joined before
        # returning (autograd accumulates the gradients on the main stream right away)
        par = LINEAR_OVERLAP_DW and OVERLAP_DW and dy2.is_cuda and ctx.needs_input_grad[0] and (want_dw or want_db) and dy2.shape[0] >= 4096
        if par:
            with fork_side(dy2.device):
                dw = gemm(dy2, x2, False, False) if want_dw else None
                db = colsum(dy2) if want_db else None
            dx = gemm(dy2, w, True, False).reshape(ctx.shp)
            join_side(dy2.device, dw, db)
            return dx, dw, db
        dx = gemm(dy2, w, True, False).reshape(ctx.shp) if ctx.needs_input_grad[0] else None
        dw = gemm(dy2, x2, False, False) if want_dw else None
        db = colsum(dy2) if want_db else None
        return dx, dw, db


def linear(x, w, b=None):
    return LinearFn.apply(x, w, b)


class MlpFn(torch.autograd.Function):
    """fc2(gelu(fc1(x))) with the GELU fused in fc1's epilogue and gelu' fused in fc2's input-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        shp = x.shape
        x2 = _f32c(x).reshape(-1, shp[-1])
        hpre = torch.empty(x2.shape[0], w1.shape[0], dtype=torch.float32, device=x.device)
        a = gemm(x2, w1, True, True, bias=b1, act=EPI_GELU, aux=hpre)
        y = gemm(a, w2, True, True, bias=b2)
        ctx.save_for_backward(x2, w1, w2, hpre, a)
        ctx.shp = shp
        return y.reshape(*shp[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, hpre, a = ctx.saved_tensors
        dy2 = _f32c(dy).reshape(-1, w2.shape[0])
        dh = gemm(dy2, w2, True, False, act=EPI_MUL_GELU_GRAD, aux=hpre)
        dw2 = gemm(dy2, a, False, False) if ctx.needs_input_grad[3] else None
        db2 = colsum(dy2) if ctx.needs_input_grad[4] else None
        dx = gemm(dh, w1, True, False).reshape(ctx.shp) if ctx.needs_input_grad[0] else None
        dw1 = gemm(dh, x2, False, False) if ctx.needs_input_grad[1] else None
        db1 = colsum(dh) if ctx.needs_input_grad[2] else None
        return dx, dw1, db1, dw2, db2


def mlp(x, w1, b1, w2, b2):
    return MlpFn.apply(x, w1, b1, w2, b2)


class MlpRelu3Fn(torch.autograd.Function):
    """Linear-ReLU-Linear-ReLU-Linear (the FoldingNet coarse MLP, models/dvae.py:226-232): ReLU fused in the producing GEMM's epilogue,
    the ReLU mask of the backward fused in the input-gradient GEMM of the following layer (ACT_EPI_MUL_RELU_MASK)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2):
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        h1 = gemm(x2, w0, True, True, bias=b0, act=EPI_RELU)
        h2 = gemm(h1, w1, True, True, bias=b1, act=EPI_RELU)
        y = gemm(h2, w2, True, True, bias=b2)
        ctx.save_for_backward(x2, h1, h2, w0, w1, w2)
        ctx.shp = x.shape
        return y.reshape(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, h1, h2, w0, w1, w2 = ctx.saved_tensors
        dy = _f32c(dy).reshape(-1, w2.shape[0])
        dw2, db2 = gemm(dy, h2, False, False), colsum(dy)
        d2 = gemm(dy, w2, True, False, act=EPI_MUL_RELU_MASK, aux=h2)        # gradient before the second ReLU
        dw1, db1 = gemm(d2, h1, False, False), colsum(d2)
        d1 = gemm(d2, w1, True, False, act=EPI_MUL_RELU_MASK, aux=h1)
        dw0, db0 = gemm(d1, x2, False, False), colsum(d1)
        dx = gemm(d1, w0, True, False).reshape(ctx.shp) if ctx.needs_input_grad[0] else None
        return dx, dw0, db0, dw1, db1, dw2, db2


def mlp_relu3(x, w0, b0, w1, b1, w2, b2):
    return MlpRelu3Fn.apply(x, w0, b0, w1, b1, w2, b2)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        shp = x.shape
        x2 = _f32c(x).reshape(-1, shp[-1])
        y, _, mean, rstd = layernorm_fwd(x2, None, gamma, beta, eps)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shp = shp
        return y.reshape(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        dy2 = _f32c(dy).reshape(x2.shape)
        want = ctx.needs_input_grad[1]
        dx, dg, db = layernorm_bwd(dy2, x2, gamma, mean, rstd, None, want_params=want)
        return dx.reshape(ctx.shp), dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps)


class BlockFnPerKernel(torch.autograd.Function):
    """One pre-LN Transformer block applied to (x + pos)  -- models/act.py:72-90 called as blk(x + pos) (:109-112), issued from the
    host one kernel at a time.  The product path is composite.BlockFn (one host call per direction, same kernels, bit-identical
    results); this form stays for ACT_COMPOSITE=0 A/B measurements and the bit-identity test.

    forward : xin = x+pos ; x1 = xin + g1*(proj(attn(LN1(xin)))+b) ; x2 = x1 + g2*(fc2(gelu(fc1(LN2(x1))))+b)
    g1/g2 are the per-sample DropPath gates (floor(keep+U)/keep) or None.  7 launches forward.
    """

    @staticmethod
    def forward(ctx, x, pos, gate1, gate2, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps, train_w):
        B, S, D = x.shape
        hd = D // heads
        x2d = _f32c(x).reshape(B * S, D)
        pos2d = _f32c(pos).reshape(B * S, D) if pos is not None else None
        need_grad = any(ctx.needs_input_grad)
        n1, xin, mean1, rstd1 = layernorm_fwd(x2d, pos2d, n1w, n1b, eps, want_stats=need_grad)
        qkv = gemm(n1, wqkv, True, True, bias=bqkv)
        att, lse = attention_fwd(qkv, B, S, heads, hd, want_lse=need_grad)
        x1 = gemm(att, wproj, True, True, bias=bproj, rowscale=gate1, rows_per_scale=S, res=xin)
        n2, _, mean2, rstd2 = layernorm_fwd(x1, None, n2w, n2b, eps, want_stats=need_grad)
        hpre = torch.empty(B * S, w1.shape[0], dtype=torch.float32, device=x.device) if need_grad else None
        a = gemm(n2, w1, True, True, bias=b1, act=EPI_GELU, aux=hpre)
        x2 = gemm(a, w2, True, True, bias=b2, rowscale=gate2, rows_per_scale=S, res=x1)
        if need_grad:
            ctx.save_for_backward(xin, mean1, rstd1, n1, qkv, att, lse, x1, mean2, rstd2, n2, hpre, a, gate1, gate2,
                                  n1w, wqkv, wproj, n2w, w1, w2)
            ctx.dims = (B, S, D, heads, hd)
            ctx.has_bqkv = bqkv is not None
            ctx.has_pos = pos is not None
            ctx.train_w = train_w
        return x2.reshape(B, S, D)

    @staticmethod
    def backward(ctx, dx2):
        (xin, mean1, rstd1, n1, qkv, att, lse, x1, mean2, rstd2, n2, hpre, a, gate1, gate2,
         n1w, wqkv, wproj, n2w, w1, w2) = ctx.saved_tensors
        B, S, D, heads, hd = ctx.dims
        tw = bool(ctx.train_w)
        dx2 = _f32c(dx2).reshape(B * S, D)
        dy2 = dx2 if gate2 is None else scale_rows(dx2, gate2, S)
        dev = dx2.device
        # train_w == 2: weight gradients on the auxiliary stream, concurrent with the dX chain (measured: +1 % on the Stage-II step,
        # -7 % on the finetune step, so only ACT_PointDistillation asks for it)
        par = ctx.train_w == 2 and OVERLAP_DW and dx2.is_cuda
        dw2 = db2 = dw1 = db1 = dwproj = dbproj = dwqkv = dbqkv = None

        def wgrad(dy, x, want_bias=True):
            """dW = dy^T x, db = column sums of dy"""
            if not tw:
                return None, None
            if par:
                with fork_side(dev):
                    return gemm(dy, x, False, False), (colsum(dy) if want_bias else None)
            return gemm(dy, x, False, False), (colsum(dy) if want_bias else None)

        # two Linears at a time through the grouped launch (csrc/gemm_grouped.hip), as act_block_bwd_f32 does -- same kernels, same K-range
        # counts, so the per-kernel path stays bit-identical to the composite one
        T = B * S
        grouped = tw and GROUPED_DW and D % 128 == 0 and w1.shape[0] % 128 == 0 and T % 16 == 0

        def wgrad2(p0, p1, bias1=True):
            def run():
                dws, dbs = gemm_tn_grouped([p0, p1], want_bias=True)
                return dws[0], dbs[0], dws[1], (dbs[1] if bias1 else None)
            if par:
                with fork_side(dev):
                    return run()
            return run()

        if not grouped:
            dw2, db2 = wgrad(dy2, a)
        dh = gemm(dy2, w2, True, False, act=EPI_MUL_GELU_GRAD, aux=hpre)
        if grouped:
            dw2, db2, dw1, db1 = wgrad2((dy2, a), (dh, n2))
        else:
            dw1, db1 = wgrad(dh, n2)
        dn2 = gemm(dh, w1, True, False)
        dx1, dg2, dbt2 = layernorm_bwd(dn2, x1, n2w, mean2, rstd2, dres=dx2, want_params=tw)
        dy1 = dx1 if gate1 is None else scale_rows(dx1, gate1, S)
        if not grouped:
            dwproj, dbproj = wgrad(dy1, att)
        datt = gemm(dy1, wproj, True, False)
        dqkv = attention_bwd(qkv, att, datt, lse, B, S, heads, hd)
        if grouped:
            dwproj, dbproj, dwqkv, dbqkv = wgrad2((dy1, att), (dqkv, n1), ctx.has_bqkv)
        else:
            dwqkv, dbqkv = wgrad(dqkv, n1, ctx.has_bqkv)
        dn1 = gemm(dqkv, wqkv, True, False)
        dxin, dg1, dbt1 = layernorm_bwd(dn1, xin, n1w, mean1, rstd1, dres=dx1, want_params=tw)
        if par:
            join_side(dev, dw2, db2, dw1, db1, dwproj, dbproj, dwqkv, dbqkv)
        dxin = dxin.reshape(B, S, D)
        return (dxin, dxin if ctx.has_pos else None, None, None, dg1, dbt1, dwqkv, dbqkv, dwproj, dbproj, dg2, dbt2,
                dw1, db1, dw2, db2, None, None, None)


def scale_rows(x, gate, rows_per_scale):
    """x[r,:] * gate[r // rows_per_scale]  (DropPath gate on a gradient)."""
    T, D = x.shape
    return (x.view(-1, rows_per_scale, D) * gate.view(-1, 1, 1)).view(T, D)


class CosineLossFn(torch.autograd.Function):
    """mean over rows of 1 - cos(student, teacher)   (models/act.py:1243-1254 with loss='cosine')."""

    @staticmethod
    def forward(ctx, student, teacher):
        D = student.shape[-1]
        s2 = _f32c(student).reshape(-1, D)
        t2 = _f32c(teacher).reshape(-1, D)
        R = s2.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=s2.device)
        row = torch.empty(R, dtype=torch.float32, device=s2.device)
        stats = torch.empty(R, 3, dtype=torch.float32, device=s2.device)
        check(lib.act_cosine_loss_fwd_f32(ptr(s2), ptr(t2), R, D, 1e-8, ptr(loss), ptr(row), ptr(stats), stream()),
              "act_cosine_loss_fwd_f32")
        ctx.save_for_backward(s2, t2, stats)
        ctx.shp = student.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        s2, t2, stats = ctx.saved_tensors
        R, D = s2.shape
        ds = torch.empty_like(s2)
        check(lib.act_cosine_loss_bwd_f32(ptr(s2), ptr(t2), ptr(stats), ptr(_f32c(g).reshape(-1)), R, D, 1e-8, ptr(ds), stream()),
              "act_cosine_loss_bwd_f32")
        return ds.reshape(ctx.shp), None


def cosine_distill_loss(student, teacher):
    return CosineLossFn.apply(student, teacher).reshape(())


def pairwise_distill_loss(student, teacher, kind, num_mask, temperature=0.07, lambda_param=5e-3):
    """loss: 'ntxent' | 'barlow' of ACT_PointDistillation (models/act.py:1192-1195,1250-1254): per cloud lightly's NTXentLoss(temperature=0.07) /
    BarlowTwinsLoss(lambda_param=5e-3) on (student[b], teacher[b]) -- the masked tokens of one cloud are the "batch" of the contrastive loss -- divided by
    num_mask, summed over the clouds, / batch size.  lightly 1.2.28 is not importable here: the algorithms are the published ones (oracle/layers.py
    restates them the same way; parity against the package itself is unpinned).  Not in any shipped recipe, so not a tuned path: the matrix products run
    on the library's GEMMs through K.linear (forward and both gradients), the row-wise pieces are a handful of elementwise / reduction ops.
      ntxent: the 2n x 2n similarity blocks of 16 clouds at a time are the diagonal blocks of ONE [16 * 2n, C] x [16 * 2n, C]^T product.
      barlow: one [D, n] x [D, n]^T product per cloud -- O(B) launches plus a transposed copy per cloud (there is no batched GEMM entry point in the
      library; acceptable for a loss that no shipped recipe selects, a cost to know about before selecting it at B = 128).
    Returns a 0-dim tensor like every other loss of this module (the reference's ``loss.mean() / batch_size``)."""
    B, n, C = student.shape
    s = _f32c(student); t = _f32c(teacher)
    if kind == "ntxent":
        z = torch.cat([torch.nn.functional.normalize(s, dim=2), torch.nn.functional.normalize(t, dim=2)], dim=1)       # [B, 2n, C]
        m = 2 * n
        eye = torch.eye(m, dtype=torch.bool, device=s.device)
        partner = torch.cat([torch.arange(n, m, device=s.device), torch.arange(0, n, device=s.device)])
        total = s.new_zeros(())
        for c0 in range(0, B, 16):
            zc = z[c0:c0 + 16].reshape(-1, C)
            g = zc.shape[0] // m
            sim = linear(zc, zc).view(g, m, g, m)
            blk = torch.diagonal(sim, dim1=0, dim2=2).permute(2, 0, 1) / temperature                                   # [g, 2n, 2n]
            pos = blk.gather(2, partner.view(1, m, 1).expand(g, m, 1)).squeeze(2)
            lse = torch.logsumexp(blk.masked_fill(eye, float("-inf")), dim=2)
            total = total + (lse - pos).mean(dim=1).sum()
        return (total / num_mask / B).reshape(())
    if kind == "barlow":
        za = (s - s.mean(1, keepdim=True)) / s.std(1, keepdim=True)                                                    # unbiased std along the tokens
        zb = (t - t.mean(1, keepdim=True)) / t.std(1, keepdim=True)
        eye = torch.eye(C, device=s.device)
        w = torch.full((C, C), lambda_param, device=s.device); w.fill_diagonal_(1.0)
        total = s.new_zeros(())
        for b in range(B):
            c = linear(za[b].t().contiguous(), zb[b].t().contiguous()) / n                                             # z_a^T z_b / n  [C, C]
            total = total + ((c - eye).pow(2) * w).sum()
        return (total / num_mask / B).reshape(())
    raise _C.ActHipError(f"pairwise_distill_loss: unknown kind {kind!r}")


class RegressionLossFn(torch.autograd.Function):
    """loss: 'l2' (nn.MSELoss) / 'smoothl1' (nn.SmoothL1Loss), mean over all elements (models/act.py:1186-1191,1255)."""

    @staticmethod
    def forward(ctx, student, teacher, kind):
        D = student.shape[-1]
        s2 = _f32c(student).reshape(-1, D); t2 = _f32c(teacher).reshape(-1, D)
        R = s2.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=s2.device)
        row = torch.empty(R, dtype=torch.float32, device=s2.device)
        check(lib.act_regression_loss_fwd_f32(ptr(s2), ptr(t2), R, D, int(kind), ptr(loss), ptr(row), stream()), "act_regression_loss_fwd_f32")
        ctx.save_for_backward(s2, t2)
        ctx.kind, ctx.shp = int(kind), student.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        s2, t2 = ctx.saved_tensors
        R, D = s2.shape
        ds = torch.empty_like(s2)
        check(lib.act_regression_loss_bwd_f32(ptr(s2), ptr(t2), ptr(_f32c(g).reshape(-1)), R, D, ctx.kind, ptr(ds), stream()),
              "act_regression_loss_bwd_f32")
        return ds.reshape(ctx.shp), None, None


def regression_distill_loss(student, teacher, kind):
    return RegressionLossFn.apply(student, teacher, {"l2": 0, "smoothl1": 1}[kind]).reshape(())


class SoftmaxXentFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean) + top-1 accuracy fraction of the same logits (models/act.py:823-830)."""

    @staticmethod
    def forward(ctx, logits, labels):
        z = _f32c(logits)
        R, C = z.shape
        lab = labels.to(torch.int64).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        acc = torch.empty(1, dtype=torch.float32, device=z.device)                 # fraction of correct rows
        buf = torch.empty(3, R, dtype=torch.float32, device=z.device)
        check(lib.act_softmax_xent_fwd_f32(ptr(z), ptr(lab), R, C, ptr(loss), ptr(buf), ptr(acc), stream()),
              "act_softmax_xent_fwd_f32")
        ctx.save_for_backward(z, lab, buf)
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g, _gacc):
        z, lab, buf = ctx.saved_tensors
        R, C = z.shape
        dz = torch.empty_like(z)
        check(lib.act_softmax_xent_bwd_f32(ptr(z), ptr(lab), ptr(buf), ptr(_f32c(g).reshape(-1)), R, C, ptr(dz), stream()),
              "act_softmax_xent_bwd_f32")
        return dz, None


def softmax_xent(logits, labels):
    """-> (mean cross-entropy loss, fraction of rows with arg-max == label), both 0-d tensors on the device."""
    loss, acc = SoftmaxXentFn.apply(logits, labels)
    return loss.reshape(()), acc.reshape(())


# ---- mini-PointNet / FoldingNet row kernels (csrc/pointnet.hip) --------------------------------------------------
_C._declare({
    "act_colstats_workspace": [_i, _i],
    "act_bn_stats_f32": [_vp, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "act_affine_act_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "act_bn_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp],
    "act_group_max_f32": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "act_group_max_bwd_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "act_group_sum_f32": [_vp, _i, _i, _i, _vp, _vp],
    "act_group_max_bwd_matmul_f32": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp],
    "act_group_max_bwd_wgrad_workspace": [_i, _i, _i, _i],
    "act_group_live_i32": [_vp, _i, _i, _vp, _vp],
    "act_group_max_bwd_matmul_live_f32": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp],
    "act_bn_bwd_groups_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp],
    "act_group_max_bwd_wgrad_f32": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp],
    "act_col_mean_var_f32": [_vp, _i, _i, _vp, _vp, _vp, _sz, _vp],
    "act_bn_bwd_sums_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp],
    "act_bn_bwd_apply_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp, _vp],
})
_C.lib.act_layernorm_bwd_workspace.restype = _sz
_C.lib.act_colsum_workspace.restype = _sz
_C.lib.act_colstats_workspace.restype = _sz
_C.lib.act_group_max_bwd_wgrad_workspace.restype = _sz
for _n in ("act_colstats_workspace", "act_bn_stats_f32", "act_affine_act_f32", "act_bn_bwd_f32", "act_group_max_f32",
           "act_group_max_bwd_f32", "act_group_max_bwd_matmul_f32", "act_group_max_bwd_wgrad_workspace", "act_group_max_bwd_wgrad_f32", "act_group_live_i32",
           "act_group_max_bwd_matmul_live_f32", "act_bn_bwd_groups_f32", "act_group_sum_f32", "act_col_mean_var_f32", "act_bn_bwd_sums_f32", "act_bn_bwd_apply_f32"):
    _C.SIGNATURES.setdefault(_n, getattr(_C.lib, _n).argtypes)


class BNActFn(torch.autograd.Function):
    """BatchNorm1d over the rows of x [R,C] (+ optional ReLU).  train: batch statistics (biased variance), running stats
    updated in place (momentum, unbiased variance) like nn.BatchNorm1d; eval: running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        x = _f32c(x)
        R, C = x.shape
        dev = x.device
        y = torch.empty_like(x)
        if training:
            mean, rstd, scale, shift = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4))
            ws = workspace(dev, lib.act_colstats_workspace(R, C))
            check(lib.act_bn_stats_f32(ptr(x), R, C, ptr(gamma), ptr(beta), float(eps), float(momentum), ptr(running_mean),
                                       ptr(running_var), ptr(mean), ptr(rstd), ptr(scale), ptr(shift), ptr(ws), ws.numel() * 4,
                                       stream()), "act_bn_stats_f32")
        else:
            rstd = torch.rsqrt(running_var + eps)
            mean = running_mean
            scale = gamma * rstd
            shift = beta - running_mean * scale
        check(lib.act_affine_act_f32(ptr(x), ptr(scale), ptr(shift), int(relu), R, C, ptr(y), stream()), "act_affine_act_f32")
        ctx.save_for_backward(x, scale, shift, mean, rstd)
        ctx.training, ctx.relu = training, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean, rstd = ctx.saved_tensors
        dy = _f32c(dy)
        R, C = x.shape
        if not ctx.training:
            raise NotImplementedError("BatchNorm backward in eval mode is off the training path")
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace(x.device, lib.act_colstats_workspace(R, C))
        check(lib.act_bn_bwd_f32(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), int(ctx.relu), R, C, ptr(dx), ptr(dg),
                                 ptr(db), ptr(ws), ws.numel() * 4, stream()), "act_bn_bwd_f32")
        return dx, dg, db, None, None, None, None, None, None


class SyncBNActFn(torch.autograd.Function):
    """nn.SyncBatchNorm (+ReLU) on rows [R,C] in train mode: batch statistics over the rows of ALL ranks of ``group`` (the reference's
    --sync_bn path, tools/runner_pretrain.py:86-88).  Per rank: one statistics pass (mean, biased variance, row count), an all-gather of
    3 x C floats, Chan's combination of the per-rank moments, the affine apply; backward: the two local column sums are all-reduced
    before dx is formed (dgamma / dbeta stay local sums, DDP averages them like every other gradient -- torch's SyncBatchNorm does the same)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, group):
        import torch.distributed as dist
        x = _f32c(x)
        R, C = x.shape
        dev = x.device
        local = torch.empty(3, C, dtype=torch.float32, device=dev)         # mean | var | count
        ws = workspace(dev, lib.act_colstats_workspace(R, C))
        check(lib.act_col_mean_var_f32(ptr(x), R, C, ptr(local[0]), ptr(local[1]), ptr(ws), ws.numel() * 4, stream()), "act_col_mean_var_f32")
        local[2].fill_(float(R))
        world = dist.get_world_size(group)
        gathered = torch.empty(world, 3, C, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(gathered, local, group=group) if dist.get_backend(group) == "nccl" else \
            dist.all_gather(list(gathered.unbind(0)), local, group=group)
        n = gathered[:, 2]                                                  # [world, C]
        total = n.sum(0)
        mean = (gathered[:, 0] * n).sum(0) / total
        var = ((gathered[:, 1] + (gathered[:, 0] - mean) ** 2) * n).sum(0) / total          # biased variance over all rows
        rstd = torch.rsqrt(var + eps)
        scale = (gamma * rstd).contiguous()
        shift = (beta - mean * scale).contiguous()
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                running_var.mul_(1 - momentum).add_(var * (total / (total - 1).clamp_min(1)), alpha=momentum)
        y = torch.empty_like(x)
        check(lib.act_affine_act_f32(ptr(x), ptr(scale), ptr(shift), int(relu), R, C, ptr(y), stream()), "act_affine_act_f32")
        ctx.save_for_backward(x, scale, shift, mean.contiguous(), rstd.contiguous(), total[:1].contiguous())
        ctx.relu, ctx.group = relu, group
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, scale, shift, mean, rstd, total = ctx.saved_tensors
        dy = _f32c(dy)
        R, C = x.shape
        dev = x.device
        sums = torch.empty(2, C, dtype=torch.float32, device=dev)           # sum dy | sum dy * xhat  (this rank)
        ws = workspace(dev, lib.act_colstats_workspace(R, C))
        check(lib.act_bn_bwd_sums_f32(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), int(ctx.relu), R, C, ptr(sums[0]), ptr(sums[1]),
                                      ptr(ws), ws.numel() * 4, stream()), "act_bn_bwd_sums_f32")
        db, dg = sums[0].clone(), sums[1].clone()
        dist.all_reduce(sums, group=ctx.group)
        dx = torch.empty_like(x)
        # the kernel divides by a HOST scalar; the true row count over all ranks lives on the device (``total``, gathered in forward).  Rescale
        # the sums by (R * world) / total there: exactly 1.0 for equal shards (bit-identical to dividing by R * world), and the right
        # normalisation for ragged last batches / non-drop_last loaders -- without a host synchronisation.
        count = float(R * dist.get_world_size(ctx.group))
        sums.mul_(count / total)
        check(lib.act_bn_bwd_apply_f32(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), ptr(sums[0]), ptr(sums[1]), count, int(ctx.relu),
                                       R, C, ptr(dx), stream()), "act_bn_bwd_apply_f32")
        return dx, dg, db, None, None, None, None, None, None


def _sync_group(bn):
    """process group of a SyncBatchNorm module when its statistics must be synchronised right now, else None"""
    if not isinstance(bn, torch.nn.SyncBatchNorm):
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return group if dist.get_world_size(group) > 1 else None


def batch_norm_act(x, bn, training, relu=True):
    """functional nn.BatchNorm1d / nn.SyncBatchNorm (+ReLU) on rows [R,C] with the module's parameters and buffers."""
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    group = _sync_group(bn) if training else None
    if group is not None:
        return SyncBNActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu, group)
    return BNActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps, relu)


class GroupMaxFn(torch.autograd.Function):
    """x [G*n, C] -> max over the n rows of every group, [G, C]   (torch.max(feature, dim=2) of models/dvae.py:211,214)."""

    @staticmethod
    def forward(ctx, x, n):
        x = _f32c(x)
        R, C = x.shape
        G = R // n
        out = torch.empty(G, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(G, C, dtype=torch.int32, device=x.device)
        check(lib.act_group_max_f32(ptr(x), G, n, C, ptr(out), ptr(arg), stream()), "act_group_max_f32")
        ctx.save_for_backward(arg)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        dout = _f32c(dout)
        G, C = dout.shape
        din = torch.empty(G * ctx.n, C, dtype=torch.float32, device=dout.device)
        check(lib.act_group_max_bwd_f32(ptr(dout), ptr(arg), G, ctx.n, C, 0, ptr(din), stream()), "act_group_max_bwd_f32")
        return din, None


def group_max(x, n):
    return GroupMaxFn.apply(x, n)


class LinearGroupAddFn(torch.autograd.Function):
    """y[r,:] = x[r,:] @ w^T + g[r // n, :]  -- the per-point half of a conv over cat(per-group feature, per-point feature)
    with the per-group half g added in the GEMM epilogue (models/dvae.py:212-213, :266-271)."""

    @staticmethod
    def forward(ctx, x, w, g, n):
        x = _f32c(x); g = _f32c(g)
        y = gemm(x, w, True, True, res=g, res_row_div=n)
        ctx.save_for_backward(x, w)
        ctx.n = n
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy)
        dx = gemm(dy, w, True, False) if ctx.needs_input_grad[0] else None
        dw = gemm(dy, x, False, False) if ctx.needs_input_grad[1] else None
        dg = None
        if ctx.needs_input_grad[2]:
            R, C = dy.shape
            dg = torch.empty(R // ctx.n, C, dtype=torch.float32, device=dy.device)
            check(lib.act_group_sum_f32(ptr(dy), R // ctx.n, ctx.n, C, ptr(dg), stream()), "act_group_sum_f32")
        return dx, dw, dg, None


def linear_group_add(x, w, g, n):
    return LinearGroupAddFn.apply(x, w, g, n)


# ---- DGCNN / tokenizer glue (csrc/dgcnn.hip) -------------------------------------------------------------------------
_u64 = ctypes.c_uint64
_C._declare({
    "act_edge_gn_lrelu_max_f32": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _vp],
    "act_gn_gumbel_argmax_gather_f32": [_vp, _i, _i, _i, _i, _vp, _vp, _f, _f, _vp, _u64, _vp, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "act_edge_gn_lrelu_max_bwd_f32": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _vp],
    "act_edge_bwd_lds": [_i],
    "act_gumbel_softmax_fwd_f32": [_vp, _i, _i, _vp, _u64, _f, _vp, _vp],
    "act_gumbel_softmax_bwd_f32": [_vp, _vp, _i, _i, _f, _vp, _vp],
    "act_kl_uniform_fwd_f32": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "act_kl_uniform_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
})
_C.lib.act_layernorm_bwd_workspace.restype = _sz
_C.lib.act_colsum_workspace.restype = _sz
_C.lib.act_colstats_workspace.restype = _sz
for _n in ("act_edge_gn_lrelu_max_f32", "act_gn_gumbel_argmax_gather_f32", "act_edge_gn_lrelu_max_bwd_f32", "act_edge_bwd_lds",
           "act_gumbel_softmax_fwd_f32", "act_gumbel_softmax_bwd_f32", "act_kl_uniform_fwd_f32", "act_kl_uniform_bwd_f32"):
    _C.SIGNATURES.setdefault(_n, getattr(_C.lib, _n).argtypes)


def edge_gn_lrelu_max(yz, zoff, idx, B, G, k, C, gn, out=None, ooff=0, slope=0.2):
    """inference-only tail of a DGCNN layer (see csrc/dgcnn.hip); yz [B*G, ld], -> out[:, ooff:ooff+C]."""
    yz = _f32c(yz)
    if out is None:
        out = torch.empty(B * G, C, dtype=torch.float32, device=yz.device)
    stats = torch.empty(18 * B * gn.num_groups, dtype=torch.float32, device=yz.device)
    check(lib.act_edge_gn_lrelu_max_f32(ptr(yz), yz.stride(0), int(zoff), ptr(idx), B, G, k, C, gn.num_groups, ptr(gn.weight),
                                        ptr(gn.bias), float(gn.eps), float(slope), ptr(stats), ptr(out), out.stride(0), int(ooff),
                                        stream()), "act_edge_gn_lrelu_max_f32")
    return out


class EdgeGnLreluMaxFn(torch.autograd.Function):
    """differentiable tail of a DGCNN layer: out[b*G+g, c] = max_j LeakyReLU(GroupNorm(Y[b, idx[b,j,g], c] + Z[b,g,c])) with
    yz = [Y | Z] (zoff = column of Z, -1: none); idx None: the k = 1 GroupNorm + LeakyReLU head.  Backward = three HIP launches
    (csrc/dgcnn.hip) + two column sums for dgamma / dbeta."""

    @staticmethod
    def forward(ctx, yz, gamma, beta, idx, zoff, B, G, k, C, groups, eps, slope):
        yz = _f32c(yz)
        out = torch.empty(B * G, C, dtype=torch.float32, device=yz.device)
        stats = torch.empty(18 * B * groups, dtype=torch.float32, device=yz.device)
        check(lib.act_edge_gn_lrelu_max_f32(ptr(yz), yz.stride(0), int(zoff), ptr(idx), B, G, k, C, groups, ptr(gamma), ptr(beta),
                                            float(eps), float(slope), ptr(stats), ptr(out), C, 0, stream()), "act_edge_gn_lrelu_max_f32")
        ctx.save_for_backward(yz, gamma, beta, idx, stats)
        ctx.cfg = (int(zoff), B, G, k, C, groups, float(slope))
        return out

    @staticmethod
    def backward(ctx, dout):
        yz, gamma, beta, idx, stats = ctx.saved_tensors
        zoff, B, G, k, C, groups, slope = ctx.cfg
        dout = _f32c(dout)
        dyz = torch.empty_like(yz)
        part = torch.empty(2, B, C, dtype=torch.float32, device=yz.device)
        mstat = torch.empty(2 * B * groups, dtype=torch.float32, device=yz.device)
        check(lib.act_edge_gn_lrelu_max_bwd_f32(ptr(yz), yz.stride(0), zoff, ptr(idx), B, G, k, C, groups, ptr(gamma), ptr(beta),
                                                ptr(stats), slope, ptr(dout), dout.stride(0), ptr(dyz), ptr(part), ptr(mstat),
                                                stream()), "act_edge_gn_lrelu_max_bwd_f32")
        return (dyz, colsum(part[0]), colsum(part[1])) + (None,) * 9


def edge_gn_lrelu_max_train(yz, zoff, idx, B, G, k, C, gn, slope=0.2):
    return EdgeGnLreluMaxFn.apply(yz, gn.weight, gn.bias, idx, zoff, B, G, k, C, gn.num_groups, gn.eps, slope)


class GumbelSoftmaxFn(torch.autograd.Function):
    """F.gumbel_softmax(logits, tau, hard=False, dim=-1) on rows [R,C]; noise: given gumbel draws or None -> in-kernel Philox."""

    @staticmethod
    def forward(ctx, logits, noise, seed, tau):
        z = _f32c(logits)
        C = z.shape[-1]
        z2 = z.reshape(-1, C)
        y = torch.empty_like(z2)
        nz = _f32c(noise).reshape(-1, C) if noise is not None else None
        check(lib.act_gumbel_softmax_fwd_f32(ptr(z2), z2.shape[0], C, ptr(nz), int(seed), float(tau), ptr(y), stream()),
              "act_gumbel_softmax_fwd_f32")
        ctx.save_for_backward(y)
        ctx.tau, ctx.shp = float(tau), logits.shape
        return y.reshape(logits.shape)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        R, C = y.shape
        dy = _f32c(dy).reshape(R, C)
        dl = torch.empty_like(y)
        check(lib.act_gumbel_softmax_bwd_f32(ptr(y), ptr(dy), R, C, ctx.tau, ptr(dl), stream()), "act_gumbel_softmax_bwd_f32")
        return dl.reshape(ctx.shp), None, None, None


def gumbel_softmax(logits, tau, noise=None, seed=0):
    return GumbelSoftmaxFn.apply(logits, noise, seed, tau)


class KLUniformFn(torch.autograd.Function):
    """KL(mean_g softmax(logits[b,g,:]) || uniform), 'batchmean' (models/dvae.py:470-476); logits [B,G,C] -> scalar."""

    @staticmethod
    def forward(ctx, logits):
        z = _f32c(logits)
        B, G, C = z.shape
        lse = torch.empty(B * G, dtype=torch.float32, device=z.device)
        qbar = torch.empty(B, C, dtype=torch.float32, device=z.device)
        out = torch.empty(1, dtype=torch.float32, device=z.device)
        check(lib.act_kl_uniform_fwd_f32(ptr(z), B, G, C, ptr(lse), ptr(qbar), ptr(out), stream()), "act_kl_uniform_fwd_f32")
        ctx.save_for_backward(z, lse, qbar)
        return out

    @staticmethod
    def backward(ctx, g):
        z, lse, qbar = ctx.saved_tensors
        B, G, C = z.shape
        dl = torch.empty_like(z)
        check(lib.act_kl_uniform_bwd_f32(ptr(z), ptr(lse), ptr(qbar), ptr(_f32c(g).reshape(-1)), B, G, C, ptr(dl), stream()),
              "act_kl_uniform_bwd_f32")
        return dl


def kl_to_uniform(logits):
    return KLUniformFn.apply(logits).reshape(())


def gn_gumbel_argmax_gather(h, B, G, gn, codebook, noise=None, seed=0, tau=1.0, want_logits=False, slope=0.2, seed_dev=None):
    """fused layer5 GroupNorm + LeakyReLU + hard gumbel-softmax + codebook lookup -> (codes [B,G,D], index [B,G], logits|None)."""
    h = _f32c(h)
    C = h.shape[1]
    D = codebook.shape[1]
    dev = h.device
    stats = torch.empty(18 * B * gn.num_groups, dtype=torch.float32, device=dev)
    index = torch.empty(B, G, dtype=torch.int64, device=dev)
    out = torch.empty(B, G, D, dtype=torch.float32, device=dev)
    logits = torch.empty(B, G, C, dtype=torch.float32, device=dev) if want_logits else None
    noise = _f32c(noise) if noise is not None else None
    check(lib.act_gn_gumbel_argmax_gather_f32(ptr(h), B, G, C, gn.num_groups, ptr(gn.weight), ptr(gn.bias), float(gn.eps), float(slope),
                                              ptr(noise), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(seed_dev), float(tau), ptr(_f32c(codebook)), D, ptr(stats),
                                              ptr(index), ptr(out), ptr(logits), stream()), "act_gn_gumbel_argmax_gather_f32")
    return out, index, logits


_C._declare({"act_attention_fwd_prefix_f32": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp],
             "act_attention_bwd_prefix_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]})
_C.lib.act_layernorm_bwd_workspace.restype = _sz
_C.lib.act_colsum_workspace.restype = _sz
_C.lib.act_colstats_workspace.restype = _sz
_C.SIGNATURES.setdefault("act_attention_fwd_prefix_f32", _C.lib.act_attention_fwd_prefix_f32.argtypes)
_C.SIGNATURES.setdefault("act_attention_bwd_prefix_f32", _C.lib.act_attention_bwd_prefix_f32.argtypes)


def attention_fwd_prefix(kv0, S0, qkv1, Sq, B, H, hd, want_lse=False):
    """queries = the Sq rows of qkv1 [B*Sq, 3*H*hd]; keys/values = S0 rows of kv0 [B*S0, 2*H*hd] then the rows of qkv1."""
    out = torch.empty(B * Sq, H * hd, dtype=torch.float32, device=qkv1.device)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=qkv1.device) if want_lse else None
    check(lib.act_attention_fwd_prefix_f32(ptr(_f32c(kv0)), S0, ptr(_f32c(qkv1)), Sq, ptr(out), ptr(lse), B, H, hd, float(hd) ** -0.5,
                                           stream()), "act_attention_fwd_prefix_f32")
    return (out, lse) if want_lse else out


def attention_bwd_prefix(kv0, S0, qkv1, Sq, out, dout, lse, B, H, hd):
    """-> (dkv0 [B*S0, 2*H*hd], dqkv1 [B*Sq, 3*H*hd])"""
    dkv0 = torch.empty_like(kv0)
    dqkv1 = torch.empty_like(qkv1)
    check(lib.act_attention_bwd_prefix_f32(ptr(kv0), S0, ptr(qkv1), Sq, ptr(out), ptr(_f32c(dout)), ptr(lse), ptr(dkv0), ptr(dqkv1),
                                           B, H, hd, float(hd) ** -0.5, stream()), "act_attention_bwd_prefix_f32")
    return dkv0, dqkv1


class PrefixBlockFnPerKernel(torch.autograd.Function):
    """Pre-LN block on G patch tokens per cloud with P prompt tokens acting as keys/values only, WITH backward to the patch
    tokens, their positions and the prompts (Stage-I prompt tuning of the frozen Transformer, models/dvae.py:536-576: every
    layer replaces the prompt rows of its input and the output drops them, so prompt rows never need queries / proj / MLP).
    The block weights are frozen (freeze_visual_embed: True); inputs x2d [B*G,D], pos2d [B*G,D], prm2d [B*P,D] = prompt+pos."""

    @staticmethod
    def forward(ctx, x2d, pos2d, prm2d, B, P, G, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps):
        D = x2d.shape[1]
        hd = D // heads
        n1p, _, meanp, rstdp = layernorm_fwd(_f32c(prm2d), None, n1w, n1b, eps, want_stats=True)
        kvp = gemm(n1p, wqkv[D:], True, True, bias=(bqkv[D:] if bqkv is not None else None))
        n1x, xin, mean1, rstd1 = layernorm_fwd(_f32c(x2d), _f32c(pos2d), n1w, n1b, eps, want_stats=True)
        qkvx = gemm(n1x, wqkv, True, True, bias=bqkv)
        att, lse = attention_fwd_prefix(kvp, P, qkvx, G, B, heads, hd, want_lse=True)
        x1 = gemm(att, wproj, True, True, bias=bproj, res=xin)
        n2, _, mean2, rstd2 = layernorm_fwd(x1, None, n2w, n2b, eps, want_stats=True)
        hpre = torch.empty(B * G, w1.shape[0], dtype=torch.float32, device=x2d.device)
        a = gemm(n2, w1, True, True, bias=b1, act=EPI_GELU, aux=hpre)
        x2 = gemm(a, w2, True, True, bias=b2, res=x1)
        ctx.save_for_backward(prm2d, meanp, rstdp, kvp, xin, mean1, rstd1, qkvx, att, lse, x1, mean2, rstd2, hpre,
                              n1w, wqkv, wproj, n2w, w1, w2)
        ctx.dims = (B, P, G, D, heads, hd)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        (prm2d, meanp, rstdp, kvp, xin, mean1, rstd1, qkvx, att, lse, x1, mean2, rstd2, hpre,
         n1w, wqkv, wproj, n2w, w1, w2) = ctx.saved_tensors
        B, P, G, D, heads, hd = ctx.dims
        dx2 = _f32c(dx2)
        dh = gemm(dx2, w2, True, False, act=EPI_MUL_GELU_GRAD, aux=hpre)
        dn2 = gemm(dh, w1, True, False)
        dx1, _, _ = layernorm_bwd(dn2, x1, n2w, mean2, rstd2, dres=dx2, want_params=False)
        datt = gemm(dx1, wproj, True, False)
        dkvp, dqkvx = attention_bwd_prefix(kvp, P, qkvx, G, att, datt, lse, B, heads, hd)
        dn1x = gemm(dqkvx, wqkv, True, False)
        dxin, _, _ = layernorm_bwd(dn1x, xin, n1w, mean1, rstd1, dres=dx1, want_params=False)
        dn1p = gemm(dkvp, wqkv[D:], True, False)
        dprm, _, _ = layernorm_bwd(dn1p, prm2d, n1w, meanp, rstdp, dres=None, want_params=False)
        return (dxin, dxin, dprm) + (None,) * 17


_C._declare({"act_prompt_layernorm_fwd_f32": [_vp, _vp, _i, _i, _i, _f, _u64, _vp, _vp, _vp, _f, _vp, _vp]})
_C.SIGNATURES.setdefault("act_prompt_layernorm_fwd_f32", _C.lib.act_prompt_layernorm_fwd_f32.argtypes)


_C._declare({"act_prompt_rows_fwd_f32": [_vp, _vp, _vp, _i, _i, _i, _f, _u64, _vp, _vp],
             "act_prompt_rows_bwd_f32": [_vp, _vp, _i, _i, _i, _f, _u64, _vp, _vp, _vp]})
for _n in ("act_prompt_rows_fwd_f32", "act_prompt_rows_bwd_f32"):
    _C.SIGNATURES.setdefault(_n, getattr(_C.lib, _n).argtypes)


class PromptRowsFn(torch.autograd.Function):
    """prompt rows of a trained prompt layer: y[b*P+p, :] = dropout(tok[p, :]) + ppos[p, :] (models/dvae.py:485-498, 556-566), one launch per
    direction.  mask: 0/1 keep mask [B, P, D] (injected / recorded draws) or None -> in-kernel Philox(seed), regenerated in the backward."""

    @staticmethod
    def forward(ctx, tok, ppos, B, drop_p, seed, mask):
        P, D = tok.shape
        tok, ppos = _f32c(tok), _f32c(ppos)
        mask = _f32c(mask).reshape(B * P, D) if mask is not None else None
        y = torch.empty(B * P, D, dtype=torch.float32, device=tok.device)
        check(lib.act_prompt_rows_fwd_f32(ptr(tok), ptr(ppos), ptr(mask), B, P, D, float(drop_p), int(seed), ptr(y), stream()), "act_prompt_rows_fwd_f32")
        ctx.save_for_backward(mask)
        ctx.cfg = (B, P, D, float(drop_p), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        B, P, D, drop_p, seed = ctx.cfg
        dy = _f32c(dy)
        dtok = torch.empty(P, D, dtype=torch.float32, device=dy.device); dppos = torch.empty_like(dtok)
        check(lib.act_prompt_rows_bwd_f32(ptr(dy), ptr(mask), B, P, D, drop_p, seed, ptr(dtok), ptr(dppos), stream()), "act_prompt_rows_bwd_f32")
        return dtok, dppos, None, None, None, None


def prompt_rows(tok, ppos, B, drop_p=0.0, seed=0, mask=None):
    return PromptRowsFn.apply(tok, ppos, B, drop_p, seed, mask)


def prompt_layernorm(tok, ppos, B, drop_p, seed, gamma, beta, eps, seed_dev=None):
    """LN(dropout(tok) + ppos) for the B x P prompt rows of one layer of the frozen teacher, dropout mask from in-kernel Philox."""
    P, D = tok.shape
    y = torch.empty(B * P, D, dtype=torch.float32, device=tok.device)
    check(lib.act_prompt_layernorm_fwd_f32(ptr(_f32c(tok)), ptr(_f32c(ppos)), B, P, D, float(drop_p), int(seed), ptr(seed_dev), ptr(gamma), ptr(beta),
                                           float(eps), ptr(y), stream()), "act_prompt_layernorm_fwd_f32")
    return y


def block_forward_prefix_perkernel(x2d, pos2d, prm2d, B, P, G, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps, n1p=None):
    """Inference-only pre-LN block on G 'patch' tokens per cloud with P extra 'prompt' tokens that act as keys/values only
    (their outputs are discarded by the caller): x2d [B*G, D] (+ pos2d), prm2d [B*P, D] = prompt + prompt_pos (or n1p = its
    LayerNorm, already computed by prompt_layernorm).
    Exactly the patch-token rows of  blk(cat(prompt, x) + cat(prompt_pos, pos))  of models/dvae.py:549-571."""
    D = x2d.shape[1]
    hd = D // heads
    if n1p is None:
        n1p, _, _, _ = layernorm_fwd(prm2d, None, n1w, n1b, eps, want_stats=False)
    kvp = gemm(n1p, wqkv[D:], True, True, bias=(bqkv[D:] if bqkv is not None else None))          # K,V of the prompts
    n1x, xin, _, _ = layernorm_fwd(x2d, pos2d, n1w, n1b, eps, want_stats=False)
    qkvx = gemm(n1x, wqkv, True, True, bias=bqkv)
    att = attention_fwd_prefix(kvp, P, qkvx, G, B, heads, hd)
    x1 = gemm(att, wproj, True, True, bias=bproj, res=xin)
    n2, _, _, _ = layernorm_fwd(x1, None, n2w, n2b, eps, want_stats=False)
    a = gemm(n2, w1, True, True, bias=b1, act=EPI_GELU)
    return gemm(a, w2, True, True, bias=b2, res=x1)


# ---- the product forms of the block-level Functions live in act_amd.composite (one host call per module); resolved lazily so that
# either module may be imported first.  ACT_COMPOSITE=0 selects the per-kernel host path above.
def __getattr__(name):
    if name == "block_stack":
        from . import composite
        return composite.block_stack
    if name in ("BlockFn", "PrefixBlockFn", "block_forward_prefix"):
        from . import composite
        if composite.ENABLED:
            return getattr(composite, name)
        return {"BlockFn": BlockFnPerKernel, "PrefixBlockFn": PrefixBlockFnPerKernel, "block_forward_prefix": block_forward_prefix_perkernel}[name]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
