"""Host side of the composite entry points of libact_hip.so (include/act_hip.h, csrc/composite.hip): one ctypes call enqueues
every kernel of a module (Transformer block forward / backward, the whole frozen prompt-tuned Transformer of the teacher, the
mini-PointNet patch embedding, DGCNN) instead of one call per kernel.

Nothing is computed here.  Each autograd Function allocates its output, ONE slab for the activations the backward needs and ONE
scratch slab per backward, fills a small parameter struct with device pointers and crosses the FFI once.

GEMM launch configurations: the C side looks every GEMM up in a table keyed by (layout, M, N, K).  Before the first execution of a
composite with given dimensions the shapes it will launch are collected (act_composite_collect_begin/_end: the call runs without
launching anything), tuned exactly like the single-GEMM path (shipped table, else first-use timing) and registered, so results never
depend on the call history.
"""
import ctypes
import os
import weakref

import torch

from . import _C
from . import kernels as K

_vp, _i, _f, _sz, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_uint64
lib, check = _C.lib, _C.check

ENABLED = os.environ.get("ACT_COMPOSITE", "1") != "0"       # 0: the one-call-per-kernel host path (A/B measurements, bit-identity test)


class BlockParams(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm2_w", "norm2_b",
                                   "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class BlockDims(ctypes.Structure):
    _fields_ = [("B", _i), ("S", _i), ("D", _i), ("heads", _i), ("hidden", _i), ("eps", _f)]


class BlockStack(ctypes.Structure):
    _fields_ = [("depth", _i), ("blocks", _vp), ("gate1", _vp), ("gate2", _vp)]


class PrefixVit(ctypes.Structure):
    _fields_ = ([(n, _i) for n in ("B", "P", "G", "D", "heads", "hidden", "depth", "tokens_dims", "pos_hidden")] +
                [("eps", _f), ("drop_p", _f), ("seed_base", _u64), ("seed_dev", _vp)] +
                [(n, _vp) for n in ("pos_w0", "pos_b0", "pos_w1", "pos_b1", "pre_w", "pre_b", "post_w", "post_b", "norm_w", "norm_b")] +
                [("prompt_tok", ctypes.POINTER(_vp)), ("prompt_pos", ctypes.POINTER(_vp)), ("blocks", ctypes.POINTER(BlockParams))])


class VitBf16x3(ctypes.Structure):
    _fields_ = [("w_planes", _vp), ("a_planes", _vp), ("a_planes_elems", _sz)]


class PointnetParams(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("c1_w", "c1_b", "bn1_w", "bn1_b", "c2_w", "c2_b", "c3_w", "c3_b", "bn2_w", "bn2_b", "c4_w", "c4_b",
                                   "bn1_mean", "bn1_var", "bn2_mean", "bn2_var")]


class PointnetGrads(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("c1_w", "c1_b", "bn1_w", "bn1_b", "c2_w", "c2_b", "c3_w", "c3_b", "bn2_w", "bn2_b", "c4_w", "c4_b")]


class PointnetDims(ctypes.Structure):
    _fields_ = [("BG", _i), ("n", _i), ("C", _i), ("eps1", _f), ("eps2", _f), ("momentum1", _f), ("momentum2", _f)]


class Dgcnn(ctypes.Structure):
    _fields_ = ([(n, _i) for n in ("B", "G", "k", "Cin", "Cout", "groups")] + [("eps", _f), ("slope", _f)] +
                [(n, _vp) for n in ("w_in", "b_in", "w5")] + [("stacked", _vp * 4), ("gn_w", _vp * 4), ("gn_b", _vp * 4)])


class GemmFx(ctypes.Structure):
    _fields_ = ([(n, _vp) for n in ("a_scale", "a_shift", "b_scale", "b_shift", "tile_stats", "gmax", "garg")] + [("group", _i), ("store_c", _i)]
                + [(n, _vp) for n in ("sa_src", "sa_arg", "ep_src", "ep_arg", "row_groups")])


_P = ctypes.POINTER
_SIGS = {
    "act_sgemm_fx_tile_stats_floats": [_i, _i],
    "act_sgemm_fx_f32": [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _P(K.GemmEpilogue), _P(GemmFx), _vp, _sz, _vp],
    "act_bn_tiles_finalize_f32": [_vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "act_gemm_tune_set": [_i] * 7,
    "act_gemm_tune_get": [_i] * 5 + [_P(_i), _P(_i)],
    "act_gemm_tune_clear": [],
    "act_gemm_fx_asm": [_i],
    "act_composite_collect_begin": [],
    "act_composite_collect_end": [_P(_i), _i],
    "act_composite_shutdown": [],
    "act_scale_rows_f32": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "act_bn_eval_affine_f32": [_vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp],
    "act_block_saved_floats": [_P(BlockDims)],
    "act_block_bwd_scratch_floats": [_P(BlockDims)],
    "act_block_fwd_f32": [_P(BlockDims), _P(BlockParams), _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp],
    "act_block_bwd_f32": [_P(BlockDims), _P(BlockParams), _vp, _vp, _vp, _vp, _vp, _P(BlockParams), _vp, _vp, _sz, _vp, _sz, _vp, _vp],
    "act_block_stack_saved_floats": [_P(BlockDims), _i, _i],
    "act_block_stack_bwd_scratch_floats": [_P(BlockDims), _i],
    "act_block_stack_fwd_f32": [_P(BlockDims), _P(BlockStack), _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp],
    "act_block_stack_bwd_f32": [_P(BlockDims), _P(BlockStack), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp],
    "act_add_f32": [_vp, _vp, _vp, ctypes.c_longlong, _vp],
    "act_prefix_block_saved_floats": [_P(BlockDims), _i],
    "act_prefix_block_bwd_scratch_floats": [_P(BlockDims), _i],
    "act_prefix_block_fwd_f32": [_P(BlockDims), _i, _P(BlockParams), _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp],
    "act_prefix_block_fwd_bf16x3_f32": [_P(BlockDims), _i, _P(BlockParams), _P(VitBf16x3), _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp],
    "act_prefix_block_bwd_bf16x3_f32": [_P(BlockDims), _i, _P(BlockParams), _P(VitBf16x3), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "act_prefix_block_bwd_f32": [_P(BlockDims), _i, _P(BlockParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "act_prefix_vit_scratch_floats": [_P(PrefixVit)],
    "act_prefix_vit_fwd_f32": [_P(PrefixVit), _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "act_prefix_vit_fwd_bf16x3_f32": [_P(PrefixVit), _P(VitBf16x3), _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "act_pointnet_saved_floats": [_P(PointnetDims)],
    "act_pointnet_bwd_scratch_floats": [_P(PointnetDims)],
    "act_pointnet_fwd_f32": [_P(PointnetDims), _P(PointnetParams), _vp, _i, _i, _vp, _vp, _vp, _sz, _vp],
    "act_pointnet_fwd_groups_f32": [_P(PointnetDims), _P(PointnetParams), _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp],
    "act_pointnet_bwd_f32": [_P(PointnetDims), _P(PointnetParams), _vp, _vp, _vp, _P(PointnetGrads), _vp, _vp, _sz, _vp],
    "act_dgcnn_scratch_floats": [_P(Dgcnn)],
    "act_dgcnn_features_f32": [_P(Dgcnn), _vp, _vp, _vp, _vp, _vp, _sz, _vp],
}
_C._declare(_SIGS)
for _n in _SIGS:
    _C.SIGNATURES.setdefault(_n, getattr(lib, _n).argtypes)
    if _n.endswith("_floats"):
        getattr(lib, _n).restype = _sz


def _p(t):
    """device address (int) of a contiguous CUDA tensor, 0 for None -- the cheap form of _C.ptr for struct fields"""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise _C.ActHipError("act_amd kernels need contiguous CUDA tensors (there is no CPU fallback)")
    _C._same_device(t)
    return t.data_ptr()


def _ws_of(device, stream_handle):
    """split-K / reduction scratch of one stream (kernels of different streams run concurrently)"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, stream_handle)
    w = K._WS.get(key)
    if w is None:
        w = K._WS[key] = torch.empty(K._WS_BYTES // 4, dtype=torch.float32, device=device)
    return w


# ---- tuning bridge ------------------------------------------------------------------------------------------------------
for (_ak, _bk, _M, _N, _K), (_tile, _sp) in K._GEMM_TABLE.items():
    lib.act_gemm_tune_set(_ak, _bk, _M, _N, _K, _tile, _sp)
_TUNED = set()
_SHAPE_BUF = (_i * (5 * 256))()


def _tune_shape(ak, bk, M, N, Kd, device):
    """the decision K._gemm_config would take for this product, registered with the C-side table; False when the shape could not be tuned
    right now (stream capture in progress) and has to be looked at again"""
    if lib.act_gemm_tune_get(ak, bk, M, N, Kd, None, None) == 0:
        return True
    if not K.AUTOTUNE or M * N * Kd < (1 << 24) or (not ak and not bk and min(M, N) <= 8):
        return True                                             # built-in cost model / skinny streaming kernel (tile 0)
    key = (int(ak), int(bk), M, N, Kd, device.index)
    cfg = K._GEMM_CACHE.get(key)                                # (0, 0) -- the cost model's pick -- is a valid cached decision: no `or`
    if cfg is None:
        cfg = K._GEMM_TABLE.get(key[:5])
    if cfg is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        a = torch.randn((M, Kd) if ak else (Kd, M), dtype=torch.float32, device=device)
        b = torch.randn((N, Kd) if bk else (Kd, N), dtype=torch.float32, device=device)
        cfg = K.first_use_config(a, b, ak, bk, M, N, Kd, K.workspace(device))
        K._NEW_TUNED[key[:5]] = cfg
    K._GEMM_CACHE[key] = cfg
    lib.act_gemm_tune_set(ak, bk, M, N, Kd, int(cfg[0]), int(cfg[1]))
    return True


def ensure_tuned(key, call, device):
    """first use of a composite with these dimensions: collect the GEMM shapes it launches (dry call), tune + register them"""
    if key in _TUNED:
        return
    check(lib.act_composite_collect_begin(), "act_composite_collect_begin")
    try:
        rc = call()
    finally:
        n = lib.act_composite_collect_end(_SHAPE_BUF, 256)
    check(rc, "composite dry call")
    done = True
    for i in range(min(n, 256)):
        done = _tune_shape(*[int(_SHAPE_BUF[5 * i + j]) for j in range(5)], device) and done
    if done:                                                    # a shape skipped during stream capture is tuned on the next eager call
        _TUNED.add(key)


def register_tuned(ak, bk, M, N, Kd, cfg):
    """called by the single-GEMM autotuner (kernels._gemm_config) so both host paths launch the same configuration"""
    lib.act_gemm_tune_set(int(ak), int(bk), M, N, Kd, int(cfg[0]), int(cfg[1]))


def reset_tuning():
    """forget every first-use decision (tests that need identical configurations across processes)"""
    _TUNED.clear()
    lib.act_gemm_tune_clear()
    for (ak, bk, M, N, Kd), (tile, sp) in K._GEMM_TABLE.items():
        lib.act_gemm_tune_set(ak, bk, M, N, Kd, tile, sp)


def shutdown():
    """destroy the fork / join events the library created for the streams seen so far (the only state it owns); synchronises the device first.
    Later composite calls create new ones.  -> number of events destroyed"""
    torch.cuda.synchronize()
    return int(lib.act_composite_shutdown())


# ---- Transformer block ------------------------------------------------------------------------------------------------
_DIMS = {}


def _block_dims(B, S, D, heads, hidden, eps):
    key = (B, S, D, heads, hidden, float(eps))
    d = _DIMS.get(key)
    if d is None:
        d = BlockDims(B, S, D, heads, hidden, float(eps))
        _DIMS[key] = d = (d, int(lib.act_block_saved_floats(ctypes.byref(d))), int(lib.act_block_bwd_scratch_floats(ctypes.byref(d))))
    return d


def _block_params(n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2):
    return BlockParams(_p(n1w), _p(n1b), _p(wqkv), _p(bqkv), _p(wproj), _p(bproj), _p(n2w), _p(n2b), _p(w1), _p(b1), _p(w2), _p(b2))


class BlockFn(torch.autograd.Function):
    """One pre-LN Transformer block applied to (x + pos) -- models/act.py:72-90 called as blk(x + pos) (:109-112) -- as ONE host call
    per direction (act_block_fwd_f32 / act_block_bwd_f32: 7 launches forward, 16-20 backward).

    forward : xin = x+pos ; x1 = xin + g1*(proj(attn(LN1(xin)))+b) ; x2 = x1 + g2*(fc2(gelu(fc1(LN2(x1))))+b)
    g1/g2 are the per-sample DropPath gates (floor(keep+U)/keep) or None.
    train_w: 0 frozen weights (dX only), 1 weight gradients in line, 2 weight gradients on the auxiliary stream."""

    @staticmethod
    def forward(ctx, x, pos, gate1, gate2, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps, train_w):
        B, S, D = x.shape
        dev = x.device
        dims, n_saved, _ = _block_dims(B, S, D, heads, w1.shape[0], eps)
        x2d = K._f32c(x).reshape(B * S, D)
        pos2d = K._f32c(pos).reshape(B * S, D) if pos is not None else None
        need_grad = any(ctx.needs_input_grad)
        prm = _block_params(n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2)
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        out = torch.empty(B, S, D, dtype=torch.float32, device=dev)
        ws = K.workspace(dev)
        args = (ctypes.byref(dims), ctypes.byref(prm), _p(x2d), _p(pos2d), _p(gate1), _p(gate2), int(need_grad), _p(saved), _p(out),
                _p(ws), ws.numel() * 4)
        ensure_tuned(("blk_fwd", B, S, D, heads, w1.shape[0]), lambda: lib.act_block_fwd_f32(*args, _C.stream()), dev)
        check(lib.act_block_fwd_f32(*args, _C.stream()), "act_block_fwd_f32")
        if need_grad:
            ctx.save_for_backward(saved, gate1, gate2, n1w, wqkv, bqkv, wproj, n2w, w1, w2)
            ctx.dims = (B, S, D, heads, w1.shape[0], eps)
            ctx.has_pos = pos is not None
            ctx.train_w = int(train_w)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved, gate1, gate2, n1w, wqkv, bqkv, wproj, n2w, w1, w2 = ctx.saved_tensors
        B, S, D, heads, hidden, eps = ctx.dims
        dev = dout.device
        dims, _, n_scratch = _block_dims(B, S, D, heads, hidden, eps)
        dout = K._f32c(dout).reshape(B * S, D)
        tw = ctx.train_w
        prm = _block_params(n1w, None, wqkv, bqkv, wproj, None, n2w, None, w1, None, w2, None)
        dx = torch.empty(B, S, D, dtype=torch.float32, device=dev)
        scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
        grads, gp = (None,) * 12, None
        if tw:
            e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
            grads = (e(D), e(D), e(3 * D, D), e(3 * D) if bqkv is not None else None, e(D, D), e(D), e(D), e(D),
                     e(hidden, D), e(hidden), e(D, hidden), e(D))
            gp = ctypes.byref(BlockParams(*[_p(g) for g in grads]))
        ws = K.workspace(dev)
        side, sws = None, None
        if tw == 2 and K.OVERLAP_DW:
            side = K.side_stream(dev, 1).cuda_stream
            sws = _ws_of(dev, side)
        args = (ctypes.byref(dims), ctypes.byref(prm), _p(gate1), _p(gate2), _p(saved), _p(dout), _p(dx), gp, _p(scratch),
                _p(ws), ws.numel() * 4, _p(sws), (sws.numel() * 4 if sws is not None else 0))
        ensure_tuned(("blk_bwd", B, S, D, heads, hidden, bool(tw)), lambda: lib.act_block_bwd_f32(*args, _C.stream(), side), dev)
        check(lib.act_block_bwd_f32(*args, _C.stream(), side), "act_block_bwd_f32")
        dg1, dbt1, dwqkv, dbqkv, dwproj, dbproj, dg2, dbt2, dw1, db1, dw2, db2 = grads
        return (dx, dx if ctx.has_pos else None, None, None, dg1, dbt1, dwqkv, dbqkv, dwproj, dbproj, dg2, dbt2,
                dw1, db1, dw2, db2, None, None, None)


# ---- a stack of blocks: the loop of TransformerEncoder / TransformerDecoder in ONE host call per direction -------------------------------
STACK = os.environ.get("ACT_BLOCK_STACK", "1") != "0"       # 0: one BlockFn per block (A/B measurements, bit-identity test)
# blocks per host call (0 = the whole stack).  Under DDP a stack's parameter gradients become ready together, when its backward call returns:
# chunks of 4 let the bucket all-reduces of the deeper blocks start while the shallower ones are still being differentiated.
STACK_CHUNK = int(os.environ.get("ACT_BLOCK_STACK_CHUNK", "0"))
_SDIMS = {}
_NPB = 12                                                    # tensors per block, in BlockParams order


def _stack_dims(B, S, D, heads, hidden, eps, depth):
    key = (B, S, D, heads, hidden, float(eps), depth)
    d = _SDIMS.get(key)
    if d is None:
        dims = BlockDims(B, S, D, heads, hidden, float(eps))
        r = ctypes.byref(dims)
        _SDIMS[key] = d = (dims, int(lib.act_block_stack_saved_floats(r, depth, 1)), int(lib.act_block_stack_saved_floats(r, depth, 0)),
                           int(lib.act_block_stack_bwd_scratch_floats(r, depth)))
    return d


def _ptr_array(tensors, dev_index):
    """(c_void_p * n) of the device addresses of contiguous CUDA tensors on the current device (None -> NULL): the lean form of _p for long lists"""
    ptrs = []
    for t in tensors:
        if t is None:
            ptrs.append(None)
            continue
        if not t.is_cuda or not t.is_contiguous() or t.device.index != dev_index:
            raise _C.ActHipError("act_amd kernels need contiguous CUDA tensors on the current device (there is no CPU fallback)")
        ptrs.append(t.data_ptr())
    return (_vp * len(ptrs))(*ptrs)


class BlockStackFn(torch.autograd.Function):
    """x = blk_l(x + pos) for the blocks of a TransformerEncoder / TransformerDecoder (models/act.py:109-112,140-143) as ONE host call per
    direction (act_block_stack_fwd_f32 / act_block_stack_bwd_f32): the same launches in the same order as ``depth`` BlockFn calls -- bit-identical,
    gradient of ``pos`` included (accumulated in the order an autograd engine folds the per-block gradients) -- for 1/depth of the host work.
    ``gates``: list of (gate_attn, gate_mlp) / None per block; ``params``: 12 tensors per block in BlockParams order (qkv bias may be None).

    ``emit_pos`` (a stack differentiated in chunks, round 6): the call also returns ``pos`` itself; the NEXT chunk takes that alias as its pos input, so
    its accumulated pos gradient arrives here as the gradient of the second output and the fold continues through it,
    ((dpos_deeper + dx_{L-1}) + dx_{L-2}) + ... -- the association of the unchunked stack, not (s_2 + s_1) + s_0 of independent chunk sums."""

    @staticmethod
    def forward(ctx, x, pos, gates, heads, eps, train_w, emit_pos, *params):
        B, S, D = x.shape
        dev = x.device
        depth = len(params) // _NPB
        hidden = params[8].shape[0]
        dims, n_keep, n_nokeep, _ = _stack_dims(B, S, D, heads, hidden, eps, depth)
        _C._same_device(x)
        x2d = K._f32c(x).reshape(B * S, D)
        pos2d = K._f32c(pos).reshape(B * S, D) if pos is not None else None
        need_grad = any(ctx.needs_input_grad)
        parr = _ptr_array(params, dev.index)
        g1 = g2 = None
        gts = ()
        if gates is not None and any(g is not None for g in gates):
            gts = tuple(t for g in gates for t in (g if g is not None else (None, None)))
            g1, g2 = _ptr_array(gts[0::2], dev.index), _ptr_array(gts[1::2], dev.index)
        st = BlockStack(depth, ctypes.cast(parr, _vp), ctypes.cast(g1, _vp) if g1 is not None else None,
                        ctypes.cast(g2, _vp) if g2 is not None else None)
        saved = torch.empty(n_keep if need_grad else n_nokeep, dtype=torch.float32, device=dev)
        out = torch.empty(B, S, D, dtype=torch.float32, device=dev)
        ws = K.workspace(dev)
        args = (ctypes.byref(dims), ctypes.byref(st), x2d.data_ptr(), pos2d.data_ptr() if pos2d is not None else None, int(need_grad),
                saved.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel() * 4)
        ensure_tuned(("stack_fwd", B, S, D, heads, hidden, depth), lambda: lib.act_block_stack_fwd_f32(*args, _C.stream()), dev)
        check(lib.act_block_stack_fwd_f32(*args, _C.stream()), "act_block_stack_fwd_f32")
        ctx.set_materialize_grads(False)                     # an unused pos alias has no gradient: None, not a zero tensor
        if need_grad:
            ctx.save_for_backward(saved, *[t for t in params if t is not None], *[t for t in gts if t is not None])
            ctx.layout = ([t is not None for t in params], [t is not None for t in gts])
            ctx.dims = (B, S, D, heads, hidden, eps, depth)
            ctx.has_pos = pos is not None
            ctx.train_w = int(train_w)
        return (out, pos) if emit_pos else out

    @staticmethod
    def backward(ctx, dout, dpos_in=None):
        sv = ctx.saved_tensors                                # (also runs the in-place-modification check on every saved weight)
        saved = sv[0]
        B, S, D, heads, hidden, eps, depth = ctx.dims
        dev = saved.device
        # the pointer arrays are rebuilt from the SAVED tensors (round 6): a ``param.data = ...`` re-assignment between forward and backward changes an
        # address without bumping a version counter; a pointer array captured in forward would then be stale (BlockFn has always re-derived them)
        it = iter(sv[1:])
        params = tuple(next(it) if has else None for has in ctx.layout[0])
        gts = tuple(next(it) if has else None for has in ctx.layout[1])
        parr = _ptr_array(params, dev.index)
        g1 = g2 = None
        if gts:
            g1, g2 = _ptr_array(gts[0::2], dev.index), _ptr_array(gts[1::2], dev.index)
        st = BlockStack(depth, ctypes.cast(parr, _vp), ctypes.cast(g1, _vp) if g1 is not None else None,
                        ctypes.cast(g2, _vp) if g2 is not None else None)
        dims, _, _, n_scratch = _stack_dims(B, S, D, heads, hidden, eps, depth)
        tw = ctx.train_w
        dx = torch.empty(B, S, D, dtype=torch.float32, device=dev)
        if dout is None:                                     # only the pos alias of this chunk was used downstream (never the case in the models)
            dout = torch.zeros(B * S, D, dtype=torch.float32, device=dev)
        dout = K._f32c(dout).reshape(B * S, D)
        if dpos_in is not None:
            dpos_in = K._f32c(dpos_in).reshape(B * S, D)
        dpos = torch.empty(B, S, D, dtype=torch.float32, device=dev) if (ctx.has_pos and (depth > 1 or dpos_in is not None)) else None
        scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
        grads, garr = (None,) * len(params), None
        if tw:
            grads = tuple(torch.empty_like(t) if t is not None else None for t in params)
            garr = _ptr_array(grads, dev.index)
        ws = K.workspace(dev)
        side, sws = None, None
        if tw == 2 and K.OVERLAP_DW:
            side = K.side_stream(dev, 1).cuda_stream
            sws = _ws_of(dev, side)
        args = (ctypes.byref(dims), ctypes.byref(st), saved.data_ptr(), dout.data_ptr(), dx.data_ptr(), dpos.data_ptr() if dpos is not None else None,
                dpos_in.data_ptr() if (dpos_in is not None and dpos is not None) else None,
                ctypes.cast(garr, _vp) if garr is not None else None, scratch.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                sws.data_ptr() if sws is not None else None, (sws.numel() * 4 if sws is not None else 0))
        ensure_tuned(("stack_bwd", B, S, D, heads, hidden, depth, bool(tw)), lambda: lib.act_block_stack_bwd_f32(*args, _C.stream(), side), dev)
        check(lib.act_block_stack_bwd_f32(*args, _C.stream(), side), "act_block_stack_bwd_f32")
        return (dx, (dpos if dpos is not None else dx) if ctx.has_pos else None, None, None, None, None, None) + grads


# host-side caches keyed by module, kept OUTSIDE the modules (weak keys): nothing un-picklable (ctypes structs) or stale is ever attached to a model that
# the user may deepcopy / torch.save / hand to mp.spawn
_LEAVES = weakref.WeakKeyDictionary()          # ModuleList of blocks -> [(norm1, qkv, proj, norm2, fc1, fc2, block)]
_VIT_LEAVES = weakref.WeakKeyDictionary()      # tokenizer -> (blocks, pos0, pos2, proj_pre, proj_post, norm)
_VIT_STRUCT = weakref.WeakKeyDictionary()      # tokenizer -> (signature, PrefixVit, scratch floats, keep-alive ctypes arrays)
_VIT_PLANES = weakref.WeakKeyDictionary()      # tokenizer -> (signature incl. weight versions, [bf16 plane tensors], pointer array)   (opt-in split-bf16 teacher)
# OPT-IN, default OFF: the Linear products of the FROZEN teacher's ViT blocks on the split-bf16 kernel (csrc/gemm_bf16x3.hip: hi + lo bf16 planes of both
# operands, three products, fp32 accumulation; teacher features move by ~7e-6 of their range, parity bar 1e-4).  Never part of a headline number:
# bench.py reports this configuration on a separate line.  Covers the inference form of the teacher (Stage II) and the same frozen ViT inside the Stage-I
# prompt-tuning graph (PrefixBlockFn: forward products, and -- unless ACT_TEACHER_BF16X3_BWD=0 -- the five input-gradient products of its backward).
TEACHER_BF16X3 = os.environ.get("ACT_TEACHER_BF16X3", "0") == "1"
TEACHER_BF16X3_BWD = os.environ.get("ACT_TEACHER_BF16X3_BWD", "1") == "1"


def _stack_leaves(blocks):
    """(norm1, qkv, proj, norm2, fc1, fc2) modules of every block, cached per ModuleList (attribute walks through nn.Module.__getattr__ are the
    bulk of the host cost of collecting 12 x depth tensors per call); rebuilt when a block of the list was replaced"""
    cache = _LEAVES.get(blocks)
    if cache is None or len(cache) != len(blocks) or any(c[6] is not b for c, b in zip(cache, blocks)):
        cache = _LEAVES[blocks] = [(b.norm1, b.attn.qkv, b.attn.proj, b.norm2, b.mlp.fc1, b.mlp.fc2, b) for b in blocks]
    return cache


def _stock_blocks(blocks):
    """every block runs the stock Block.forward of act_amd.models.act (a subclass that overrides forward, or a block carrying forward hooks, must be
    CALLED -- the stack path bypasses Module.__call__) -- cached per ModuleList by _stack_leaves' key"""
    from act_amd.models.act import Block
    for b in blocks:
        if type(b).forward is not Block.forward or b._forward_hooks or b._forward_pre_hooks or b._backward_hooks or b._backward_pre_hooks:
            return False
    return True


def block_stack(blocks, x, pos, gates, draws=None, tag="enc", chunk=None):
    """the loop ``for blk in blocks: x = blk(x + pos)`` -- one BlockStackFn per chunk of blocks when the composite path is on, else one
    Block.forward per block (ACT_COMPOSITE=0 / ACT_BLOCK_STACK=0).  ``chunk``: blocks per host call (None: the ACT_BLOCK_STACK_CHUNK default,
    0: the whole stack); the owning TransformerEncoder / Decoder passes its ``stack_chunk`` (set by runner_pretrain.wrap_ddp under multi-rank DDP)."""
    n = len(blocks)
    if not (ENABLED and STACK) or n == 0 or not _stock_blocks(blocks):
        for i, blk in enumerate(blocks):
            x = blk(x, pos, draws, f"{tag}.{i}", gates[i] if gates is not None else None)
        return x
    if draws is not None:                                    # injected DropPath draws (parity tests): the gates every Block.forward would compute
        gates = [blk.gates(x.shape[0], x.device, draws, f"{tag}.{i}") for i, blk in enumerate(blocks)]
        gates = [tuple(g) if g[0] is not None else None for g in gates]
    leaves = _stack_leaves(blocks)
    b0 = blocks[0]
    heads, eps, tw = b0.attn.num_heads, b0.norm1.eps, (2 if b0.overlap_wgrad else 1)
    if any(l[6].attn.num_heads != heads or l[0].eps != eps or l[3].eps != eps or bool(l[6].overlap_wgrad) != bool(b0.overlap_wgrad) for l in leaves[1:]):
        for i, blk in enumerate(blocks):                     # blocks that differ in more than their weights: one call each
            x = blk(x, pos, None, f"{tag}.{i}", gates[i] if gates is not None else None)
        return x
    if chunk is None:
        chunk = STACK_CHUNK
    chunk = chunk if chunk > 0 else n
    carry = pos is not None and chunk < n and torch.is_grad_enabled() and pos.requires_grad
    for c0 in range(0, n, chunk):
        params = []
        for n1, qkv, proj, n2, fc1, fc2, _ in leaves[c0:c0 + chunk]:
            p1, pq, pp, p2, pf1, pf2 = n1._parameters, qkv._parameters, proj._parameters, n2._parameters, fc1._parameters, fc2._parameters
            params += [p1["weight"], p1["bias"], pq["weight"], pq["bias"], pp["weight"], pp["bias"], p2["weight"], p2["bias"],
                       pf1["weight"], pf1["bias"], pf2["weight"], pf2["bias"]]
        emit = carry and c0 + chunk < n                      # every chunk but the deepest hands pos on (see BlockStackFn)
        r = BlockStackFn.apply(x, pos, gates[c0:c0 + chunk] if gates is not None else None, heads, eps, tw, emit, *params)
        x, pos = r if emit else (r, pos)
    return x


# ---- prefix block (prompts = keys / values only) ---------------------------------------------------------------------------
_PDIMS = {}


def _prefix_dims(B, G, D, heads, hidden, eps, P):
    key = (B, G, D, heads, hidden, float(eps), P)
    d = _PDIMS.get(key)
    if d is None:
        d = BlockDims(B, G, D, heads, hidden, float(eps))
        _PDIMS[key] = d = (d, int(lib.act_prefix_block_saved_floats(ctypes.byref(d), P)),
                           int(lib.act_prefix_block_bwd_scratch_floats(ctypes.byref(d), P)))
    return d


class PrefixBlockFn(torch.autograd.Function):
    """Pre-LN block on G patch tokens per cloud with P prompt tokens acting as keys/values only, WITH backward to the patch
    tokens, their positions and the prompts (Stage-I prompt tuning of the frozen Transformer, models/dvae.py:536-576: every
    layer replaces the prompt rows of its input and the output drops them, so prompt rows never need queries / proj / MLP).
    The block weights are frozen (freeze_visual_embed: True); inputs x2d [B*G,D], pos2d [B*G,D], prm2d [B*P,D] = prompt+pos."""

    @staticmethod
    def forward(ctx, x2d, pos2d, prm2d, B, P, G, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps):
        D = x2d.shape[1]
        dev = x2d.device
        dims, n_saved, _ = _prefix_dims(B, G, D, heads, w1.shape[0], eps, P)
        x2d, pos2d, prm2d = K._f32c(x2d), K._f32c(pos2d), K._f32c(prm2d)
        prm = _block_params(n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2)
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        out = torch.empty(B * G, D, dtype=torch.float32, device=dev)
        ws = K.workspace(dev)
        args = (ctypes.byref(dims), P, ctypes.byref(prm), _p(x2d), _p(pos2d), _p(prm2d), None, 1, _p(saved), _p(out), _p(ws), ws.numel() * 4)
        ensure_tuned(("pfx_fwd", B, G, D, heads, w1.shape[0], P), lambda: lib.act_prefix_block_fwd_f32(*args, _C.stream()), dev)
        if TEACHER_BF16X3 and not any(w.requires_grad for w in (wqkv, wproj, w1, w2)):
            # OPT-IN: the frozen block's products in split-bf16 (the backward stays f32: it reads only what this forward still writes in fp32)
            x3, keep = _block_planes((wqkv, wproj, w1, w2), B * G * w1.shape[0] + B * max(G, P) * D, dev)
            check(lib.act_prefix_block_fwd_bf16x3_f32(*args[:3], ctypes.byref(x3), *args[3:6], 1, *args[8:], _C.stream()), "act_prefix_block_fwd_bf16x3_f32")
            del keep
            ctx.x3 = True
        else:
            check(lib.act_prefix_block_fwd_f32(*args, _C.stream()), "act_prefix_block_fwd_f32")
            ctx.x3 = False
        ctx.save_for_backward(saved, prm2d, n1w, wqkv, wproj, n2w, w1, w2)
        ctx.dims = (B, P, G, D, heads, w1.shape[0], eps)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved, prm2d, n1w, wqkv, wproj, n2w, w1, w2 = ctx.saved_tensors
        B, P, G, D, heads, hidden, eps = ctx.dims
        dev = dout.device
        dims, _, n_scratch = _prefix_dims(B, G, D, heads, hidden, eps, P)
        dout = K._f32c(dout)
        prm = _block_params(n1w, None, wqkv, None, wproj, None, n2w, None, w1, None, w2, None)
        dx = torch.empty(B * G, D, dtype=torch.float32, device=dev)
        dprm = torch.empty(B * P, D, dtype=torch.float32, device=dev)
        scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
        ws = K.workspace(dev)
        args = (ctypes.byref(dims), P, ctypes.byref(prm), _p(prm2d), _p(saved), _p(dout), _p(dx), _p(dprm), _p(scratch), _p(ws), ws.numel() * 4)
        ensure_tuned(("pfx_bwd", B, G, D, heads, hidden, P), lambda: lib.act_prefix_block_bwd_f32(*args, _C.stream()), dev)
        if ctx.x3 and TEACHER_BF16X3_BWD:
            x3, keep = _block_planes((wqkv, wproj, w1, w2), B * G * hidden + B * max(G, P) * D, dev, transposed=True)
            check(lib.act_prefix_block_bwd_bf16x3_f32(*args[:3], ctypes.byref(x3), *args[3:], _C.stream()), "act_prefix_block_bwd_bf16x3_f32")
            del keep
            return (dx, dx, dprm) + (None,) * 17
        check(lib.act_prefix_block_bwd_f32(*args, _C.stream()), "act_prefix_block_bwd_f32")
        return (dx, dx, dprm) + (None,) * 17


def block_forward_prefix(x2d, pos2d, prm2d, B, P, G, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps, n1p=None):
    """Inference-only prefix block (one host call): exactly the patch-token rows of
    blk(cat(prompt, x) + cat(prompt_pos, pos)) of models/dvae.py:549-571.  prm2d [B*P, D] = prompt + prompt_pos, or n1p = its LayerNorm."""
    D = x2d.shape[1]
    dev = x2d.device
    dims, n_saved, _ = _prefix_dims(B, G, D, heads, w1.shape[0], eps, P)
    x2d, pos2d = K._f32c(x2d), K._f32c(pos2d)
    prm2d = K._f32c(prm2d) if prm2d is not None else None
    n1p = K._f32c(n1p) if n1p is not None else None
    prm = _block_params(n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2)
    tmp = torch.empty(n_saved, dtype=torch.float32, device=dev)
    out = torch.empty(B * G, D, dtype=torch.float32, device=dev)
    ws = K.workspace(dev)
    args = (ctypes.byref(dims), P, ctypes.byref(prm), _p(x2d), _p(pos2d), _p(prm2d), _p(n1p), 0, _p(tmp), _p(out), _p(ws), ws.numel() * 4)
    ensure_tuned(("pfx_fwd", B, G, D, heads, w1.shape[0], P), lambda: lib.act_prefix_block_fwd_f32(*args, _C.stream()), dev)
    check(lib.act_prefix_block_fwd_f32(*args, _C.stream()), "act_prefix_block_fwd_f32")
    return out


# ---- the frozen prompt-tuned Transformer of the teacher, whole stack --------------------------------------------------------
def _vit_tensors(tok):
    """every tensor the frozen prompt-tuned Transformer reads, in a fixed order: 10 stem tensors, 12 per block, 4 prompt tables.  The leaf MODULES are
    cached per ``tok`` (attribute walks through nn.Module.__getattr__ cost more than the launches they feed); the tensors are read fresh each call."""
    lv = _VIT_LEAVES.get(tok)
    blocks = tok.visual_embed[0]
    if lv is None or lv[0] is not blocks:
        vp = tok.visual_pos_embed
        lv = _VIT_LEAVES[tok] = (blocks, vp[0], vp[2], tok.proj_pre, tok.proj_post, tok.visual_embed[1])
    _, vp0, vp2, pre, post, nrm = lv
    ts = []
    for m in (vp0, vp2, pre, post, nrm):
        pr = m._parameters
        ts += [pr["weight"], pr["bias"]]
    for n1, qkv, proj, n2, fc1, fc2, _ in _stack_leaves(blocks):
        p1, pq, pp, p2, pf1, pf2 = n1._parameters, qkv._parameters, proj._parameters, n2._parameters, fc1._parameters, fc2._parameters
        ts += [p1["weight"], p1["bias"], pq["weight"], pq["bias"], pp["weight"], pp["bias"], p2["weight"], p2["bias"],
               pf1["weight"], pf1["bias"], pf2["weight"], pf2["bias"]]
    tp = tok._parameters
    ts += [tp["visual_prompt_token"], tp["visual_prompt_pos"], tp.get("deep_prompt_tokens"), tp.get("deep_prompt_pos")]
    return ts, blocks


def prefix_vit_forward(tok, tokens, center, drop_p, seed_base, seed_dev):
    """``tok`` = ACTPromptedDiscreteVAEwithVIT (frozen); tokens [B,G,tokens_dims], center [B,G,3] -> [B,G,tokens_dims].
    visual_embedding_deep_prompt (models/dvae.py:536-576), inference form, in one host call (~125 launches).  The parameter struct (~190 device
    pointers) is cached per ``tok`` (module-level weak dictionary) and re-used as long as every tensor still sits at the same address (a ``.to()``, a re-assigned Parameter or a
    changed batch geometry rebuilds it)."""
    B, G, td = tokens.shape
    dev = tokens.device
    ts, blocks = _vit_tensors(tok)
    depth = tok.visual_embed_depth
    sig = (B, G, td, dev.index, depth, tuple([t.data_ptr() if t is not None else 0 for t in ts]))
    cache = _VIT_STRUCT.get(tok)
    if cache is None or cache[0] != sig:
        Pn, D = tok.num_prompt_token, tok.visual_embed_dim
        b0 = blocks[0]
        if depth > 1 and (ts[-2] is None or ts[-1] is None):
            raise _C.ActHipError("prefix_vit_forward: a depth > 1 prompt-tuned Transformer needs deep_prompt_tokens / deep_prompt_pos")
        ptr = _ptr_array(ts, dev.index)                        # validates device / contiguity of every tensor
        m = PrefixVit()
        m.B, m.P, m.G, m.D, m.heads, m.hidden, m.depth, m.tokens_dims, m.pos_hidden = (B, Pn, G, D, b0.num_heads, ts[18].shape[0], depth, td,
                                                                                        ts[0].shape[0])
        m.eps = float(b0.eps)
        (m.pos_w0, m.pos_b0, m.pos_w1, m.pos_b1, m.pre_w, m.pre_b, m.post_w, m.post_b, m.norm_w, m.norm_b) = [ptr[i] for i in range(10)]
        row = Pn * D * 4                                              # prompt tables [1 | depth-1, Pn, D]: row i of the deep table = layer i + 1
        nb = 10 + _NPB * depth
        toks = (_vp * depth)(*[ptr[nb] if i == 0 else ptr[nb + 2] + (i - 1) * row for i in range(depth)])
        poss = (_vp * depth)(*[ptr[nb + 1] if i == 0 else ptr[nb + 3] + (i - 1) * row for i in range(depth)])
        blks = (_vp * (_NPB * depth))(*[ptr[10 + i] for i in range(_NPB * depth)])
        m.prompt_tok, m.prompt_pos, m.blocks = toks, poss, ctypes.cast(blks, _P(BlockParams))
        n_scratch = int(lib.act_prefix_vit_scratch_floats(ctypes.byref(m)))
        cache = _VIT_STRUCT[tok] = (sig, m, n_scratch, (toks, poss, blks, ptr))
    _, m, n_scratch, _ = cache
    m.drop_p, m.seed_base, m.seed_dev = float(drop_p), int(seed_base) & (2 ** 64 - 1), _p(seed_dev)
    scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
    out = torch.empty(B, G, td, dtype=torch.float32, device=dev)
    ws = K.workspace(dev)
    tokens, center = K._f32c(tokens), K._f32c(center)
    args = (ctypes.byref(m), _p(tokens), _p(center), _p(out), _p(scratch), _p(ws), ws.numel() * 4)
    # (the f32 shapes are collected and tuned in either mode: a shape the split-bf16 kernel does not take falls back to them)
    ensure_tuned(("vit", B, m.P, G, m.D, m.heads, m.hidden, depth, td), lambda: lib.act_prefix_vit_fwd_f32(*args, _C.stream()), dev)
    if TEACHER_BF16X3:
        x3, keep = _vit_planes(tok, ts, depth, B * G * m.hidden + B * max(G, m.P) * m.D, dev)
        check(lib.act_prefix_vit_fwd_bf16x3_f32(args[0], ctypes.byref(x3), *args[1:], _C.stream()), "act_prefix_vit_fwd_bf16x3_f32")
        del keep
        return out
    check(lib.act_prefix_vit_fwd_f32(*args, _C.stream()), "act_prefix_vit_fwd_f32")
    return out


_BLOCK_PLANES = {}     # (id(qkv weight Parameter), transposed) -> (weak reference to it, signature, planes, pointer array) of ONE frozen block; the entry dies with the Parameter
#                        (not a WeakKeyDictionary: tensors as dictionary keys compare element-wise on a hash collision)


def _block_planes(ws_, act_elems, dev, transposed=False):
    """Same for the four weights (qkv, proj, fc1, fc2) of one frozen block of the differentiable Stage-I forward (PrefixBlockFn), cached on the qkv Parameter.
    ``transposed``: the planes its backward multiplies with instead -- fc2^T, fc1^T, proj^T, qkv^T and (qkv rows D..3D)^T (act_prefix_block_bwd_bf16x3_f32)."""
    sig = tuple((w.data_ptr(), w._version) for w in ws_)
    key = (id(ws_[0]), transposed)
    cache = _BLOCK_PLANES.get(key)
    if cache is None or cache[0]() is not ws_[0] or cache[1] != sig:
        with torch.no_grad():
            if transposed:
                wqkv, wproj, w1, w2 = (w.detach() for w in ws_)
                D = wqkv.shape[1]
                src = [w2.t(), w1.t(), wproj.t(), wqkv.t(), wqkv[D:].t()]
            else:
                src = [w.detach() for w in ws_]
            planes = [K.split_bf16x2(w.contiguous()) for w in src]
        parr = (_vp * len(planes))(*[pl.data_ptr() for pl in planes])
        cache = _BLOCK_PLANES[key] = (weakref.ref(ws_[0], lambda _, k=key: _BLOCK_PLANES.pop(k, None)), sig, planes, parr)
    a_planes = torch.empty(2 * act_elems, dtype=torch.bfloat16, device=dev)
    return VitBf16x3(ctypes.cast(cache[3], _vp), a_planes.data_ptr(), a_planes.numel()), (a_planes, cache)


def _vit_planes(tok, ts, depth, act_elems, dev):
    """(hi, lo) bf16 planes of the four Linear weights of every teacher block, split ONCE and cached per tokenizer; re-split when a weight moved or was
    written to (address + version counter of every weight in the signature: load_state_dict copies in place).  -> (VitBf16x3, keep-alive)"""
    ws_ = [ts[10 + _NPB * i + j] for i in range(depth) for j in (2, 4, 8, 10)]                    # qkv_w, proj_w, fc1_w, fc2_w
    sig = tuple((w.data_ptr(), w._version) for w in ws_)
    cache = _VIT_PLANES.get(tok)
    if cache is None or cache[0] != sig:
        with torch.no_grad():
            planes = [K.split_bf16x2(w.detach()) for w in ws_]
        parr = (_vp * len(planes))(*[pl.data_ptr() for pl in planes])
        cache = _VIT_PLANES[tok] = (sig, planes, parr)
    a_planes = torch.empty(2 * act_elems, dtype=torch.bfloat16, device=dev)
    x3 = VitBf16x3(ctypes.cast(cache[2], _vp), a_planes.data_ptr(), a_planes.numel())
    return x3, (a_planes, cache)


# ---- mini-PointNet patch embedding ---------------------------------------------------------------------------------------
def _pointnet_structs(enc, BG, n):
    c1, bn1, _, c2 = enc.first_conv
    c3, bn2, _, c4 = enc.second_conv
    dims = PointnetDims(BG, n, c4.weight.shape[0], float(bn1.eps), float(bn2.eps), float(bn1.momentum), float(bn2.momentum))
    return dims, (c1, bn1, c2, c3, bn2, c4)


class PointnetFn(torch.autograd.Function):
    """Encoder.forward (models/dvae.py:201-215) on rows [BG*n, 3] -> tokens [BG, C]: act_pointnet_fwd_f32 / act_pointnet_bwd_f32.
    ``buffers`` = (bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var), updated in place when training."""

    @staticmethod
    def forward(ctx, x, c1w, c1b, g1, b1, c2w, c2b, c3w, c3b, g2, b2, c4w, c4b, buffers, dims, training, groups=None):
        # ``groups``: int32 ids of the only groups whose tokens are wanted (padded so that len * n % 128 == 0), or None for all
        dev = x.device
        x = K._f32c(x)
        need_grad = any(ctx.needs_input_grad)
        prm = PointnetParams(*[_p(t) for t in (c1w, c1b, g1, b1, c2w, c2b, c3w, c3b, g2, b2, c4w, c4b) + tuple(buffers)])
        saved = torch.empty(int(lib.act_pointnet_saved_floats(ctypes.byref(dims))), dtype=torch.float32, device=dev)
        out = torch.empty(dims.BG, dims.C, dtype=torch.float32, device=dev)
        ws = K.workspace(dev)
        ng = 0 if groups is None else int(groups.numel())
        args = (ctypes.byref(dims), ctypes.byref(prm), _p(x), int(training), int(need_grad), _p(saved), _p(out),
                _p(groups) if ng else None, ng, _p(ws), ws.numel() * 4)
        ensure_tuned(("pn_fwd", dims.BG, dims.n, dims.C, ng), lambda: lib.act_pointnet_fwd_groups_f32(*args, _C.stream()), dev)
        check(lib.act_pointnet_fwd_groups_f32(*args, _C.stream()), "act_pointnet_fwd_groups_f32")
        if need_grad:
            ctx.save_for_backward(x, saved, c1w, c1b, g1, b1, c2w, c2b, c3w, c3b, g2, b2, c4w, c4b)
            ctx.dims, ctx.training = dims, bool(training)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise NotImplementedError("mini-PointNet backward in eval mode (running statistics) is off the training path")
        x, saved = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        dims = ctx.dims
        dev = dout.device
        dout = K._f32c(dout)
        prm = PointnetParams(*([_p(t) for t in params] + [None] * 4))
        grads = tuple(torch.empty_like(t) for t in params)
        gp = PointnetGrads(*[_p(g) for g in grads])
        scratch = torch.empty(int(lib.act_pointnet_bwd_scratch_floats(ctypes.byref(dims))), dtype=torch.float32, device=dev)
        ws = K.workspace(dev)
        args = (ctypes.byref(dims), ctypes.byref(prm), _p(x), _p(saved), _p(dout), ctypes.byref(gp), _p(scratch), _p(ws), ws.numel() * 4)
        ensure_tuned(("pn_bwd", dims.BG, dims.n, dims.C), lambda: lib.act_pointnet_bwd_f32(*args, _C.stream()), dev)
        check(lib.act_pointnet_bwd_f32(*args, _C.stream()), "act_pointnet_bwd_f32")
        return (None,) + grads + (None, None, None, None)


def pointnet_forward(enc, point_groups, need=None):
    """``enc`` = models.dvae.Encoder; point_groups [bs, g, n, 3] -> [bs, g, C].  ``need`` [bs, k] (int64 group indices per cloud, a fixed count per
    cloud) restricts the last conv + max-pool to those groups: the other tokens come back as zeros (csrc/composite.hip act_pointnet_fwd_groups_f32)."""
    bs, g, n, _ = point_groups.shape
    groups = None
    if need is not None and need.shape[1] < g and n in (32, 64):
        ids = (need + torch.arange(bs, device=need.device).unsqueeze(1) * g).reshape(-1).to(torch.int32)
        pad = (-ids.numel()) % (128 // n)                            # whole 128-row tiles: repeat the last group
        groups = torch.cat([ids, ids[-1:].expand(pad)]) if pad else ids
        groups = groups.contiguous()
    dims, (c1, bn1, c2, c3, bn2, c4) = _pointnet_structs(enc, bs * g, n)
    training = enc.training
    if training:
        for bn in (bn1, bn2):
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
    w2 = lambda c: c.weight.view(c.weight.shape[0], c.weight.shape[1])
    out = PointnetFn.apply(point_groups.reshape(bs * g * n, 3), w2(c1), c1.bias, bn1.weight, bn1.bias, w2(c2), c2.bias, w2(c3), c3.bias,
                           bn2.weight, bn2.bias, w2(c4), c4.bias,
                           (bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var), dims, training, groups)
    return out.reshape(bs, g, dims.C)


# ---- DGCNN, inference form -------------------------------------------------------------------------------------------------
def dgcnn_features(dg, f, idx, stacked):
    """``dg`` = models.dvae.DGCNN (frozen / no grad); f [B,G,Cin], idx int64 [B,k,G], stacked = the four [Wa ; Wb-Wa] matrices
    -> pre-norm head rows [B*G, Cout] (everything up to layer5's GroupNorm), one host call (10 launches)."""
    B, G, Cin = f.shape
    dev = f.device
    w5 = dg.layer5[0].weight
    m = Dgcnn()
    gn0 = dg.layer1[1]
    m.B, m.G, m.k, m.Cin, m.Cout, m.groups = B, G, idx.shape[1], Cin, w5.shape[0], gn0.num_groups
    m.eps, m.slope = float(gn0.eps), 0.2
    m.w_in, m.b_in, m.w5 = _p(dg.input_trans.weight), _p(dg.input_trans.bias), _p(w5)
    for l, layer in enumerate((dg.layer1, dg.layer2, dg.layer3, dg.layer4)):
        m.stacked[l], m.gn_w[l], m.gn_b[l] = _p(stacked[l]), _p(layer[1].weight), _p(layer[1].bias)
    scratch = torch.empty(int(lib.act_dgcnn_scratch_floats(ctypes.byref(m))), dtype=torch.float32, device=dev)
    h = torch.empty(B * G, m.Cout, dtype=torch.float32, device=dev)
    ws = K.workspace(dev)
    f = K._f32c(f)
    args = (ctypes.byref(m), _p(f), _p(idx), _p(h), _p(scratch), _p(ws), ws.numel() * 4)
    ensure_tuned(("dgcnn", B, G, Cin, m.Cout), lambda: lib.act_dgcnn_features_f32(*args, _C.stream()), dev)
    check(lib.act_dgcnn_features_f32(*args, _C.stream()), "act_dgcnn_features_f32")
    return h
