"""The composite entry points (csrc/composite.hip: one host call per module) must be BIT-identical to issuing the same kernels
one ctypes call at a time (act_amd.kernels *PerKernel forms, ACT_COMPOSITE=0): same kernels, same order, same launch
configurations.  Checked for the Transformer block (with DropPath gates, position add, weight gradients in line and on the
auxiliary stream), the prompt-prefix block (train + inference), the mini-PointNet Encoder (train fwd/bwd, eval), the DGCNN
inference stack and the whole frozen prompt-tuned Transformer of the teacher (in-kernel Philox dropout active)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _params(dev, D, hidden, qkv_bias, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, off=0.0: (off + torch.randn(*s, generator=g) * 0.05).to(dev).requires_grad_(True)
    return [r(D, off=1.0), r(D), r(3 * D, D), (r(3 * D) if qkv_bias else None), r(D, D), r(D), r(D, off=1.0), r(D), r(hidden, D), r(hidden),
            r(D, hidden), r(D)]


@pytest.mark.parametrize("B,S,D,H,qkv_bias,train_w", [(4, 14, 384, 6, False, 1), (4, 14, 384, 6, False, 2), (3, 64, 128, 2, True, 2),
                                                      (2, 33, 64, 2, True, 0)])
def test_block_composite_is_bit_identical(dev, B, S, D, H, qkv_bias, train_w):
    import act_amd.kernels as K
    import act_amd.composite as CP
    torch.manual_seed(0)
    x0 = torch.randn(B, S, D, device=dev); pos0 = 0.1 * torch.randn(B, S, D, device=dev)
    g1 = torch.floor(0.8 + torch.rand(B, device=dev)) / 0.8; g2 = torch.floor(0.8 + torch.rand(B, device=dev)) / 0.8
    dout = torch.randn(B, S, D, device=dev)
    res = []
    for fn in (K.BlockFnPerKernel, CP.BlockFn):
        ps = _params(dev, D, 4 * D, qkv_bias, 1)
        x, pos = x0.clone().requires_grad_(True), pos0.clone().requires_grad_(True)
        y = fn.apply(x, pos, g1, g2, *ps, H, 1e-5, train_w)
        y.backward(dout)
        torch.cuda.synchronize()
        res.append([y.detach(), x.grad, pos.grad] + [p.grad for p in ps if p is not None])
    for a, b in zip(*res):
        if train_w == 0 and a is None:
            assert b is None
            continue
        assert torch.equal(a, b)
    assert res[0][1].abs().max() > 0


@pytest.mark.parametrize("B,S,D,H,qkv_bias,train_w,depth,with_pos,with_gates", [
    (4, 14, 384, 6, False, 2, 12, True, True),        # the student encoder of the benchmark (weight gradients on the auxiliary stream)
    (4, 64, 384, 6, False, 1, 2, True, True),         # the decoder
    (3, 33, 128, 2, True, 1, 5, True, False),
    (2, 16, 64, 2, True, 1, 3, False, True),          # no position input: dpos is None
    (2, 16, 64, 2, False, 0, 4, True, False),         # frozen stack: dX only
    (2, 16, 64, 2, False, 1, 1, True, True),          # depth 1: the gradient of pos IS dx
])
def test_block_stack_is_bit_identical_to_a_chain_of_block_calls(dev, B, S, D, H, qkv_bias, train_w, depth, with_pos, with_gates):
    """act_block_stack_fwd/bwd_f32 (composite.BlockStackFn: every block of a TransformerEncoder / Decoder in one host call per direction) against
    ``depth`` BlockFn calls chained by autograd (models/act.py:109-112): output, gradient of x, the ACCUMULATED gradient of the shared pos (same
    association order as the autograd engine's input buffer) and every parameter gradient, bit for bit."""
    import act_amd.composite as CP
    torch.manual_seed(0)
    x0 = torch.randn(B, S, D, device=dev); pos0 = 0.1 * torch.randn(B, S, D, device=dev)
    gates = [(torch.floor(0.8 + torch.rand(B, device=dev)) / 0.8, torch.floor(0.8 + torch.rand(B, device=dev)) / 0.8) if (with_gates and l % 2 == 1) else None
             for l in range(depth)]
    dout = torch.randn(B, S, D, device=dev)
    res = []
    for stacked in (False, True):
        ps = [_params(dev, D, 4 * D, qkv_bias, 10 + l) for l in range(depth)]
        if train_w == 0:
            ps = [[p.detach() if p is not None else None for p in blk] for blk in ps]
        x = x0.clone().requires_grad_(True)
        pos = pos0.clone().requires_grad_(True) if with_pos else None
        if stacked:
            y = CP.BlockStackFn.apply(x, pos, gates, H, 1e-5, train_w, False, *[p for blk in ps for p in blk])
        else:
            y = x
            for l in range(depth):
                g1, g2 = gates[l] if gates[l] is not None else (None, None)
                y = CP.BlockFn.apply(y, pos, g1, g2, *ps[l], H, 1e-5, train_w)
        y.backward(dout)
        torch.cuda.synchronize()
        res.append([y.detach(), x.grad, pos.grad if with_pos else None] + [p.grad for blk in ps for p in blk if p is not None])
    for i, (a, b) in enumerate(zip(*res)):
        if a is None:
            assert b is None, i
            continue
        assert torch.equal(a, b), i
    assert res[0][1].abs().max() > 0
    # inference form (no_grad: one saved slab re-used by every block)
    with torch.no_grad():
        ps = [[p.detach() if p is not None else None for p in _params(dev, D, 4 * D, qkv_bias, 10 + l)] for l in range(depth)]
        yi = CP.BlockStackFn.apply(x0, pos0 if with_pos else None, gates, H, 1e-5, train_w, False, *[p for blk in ps for p in blk])
    assert torch.equal(yi, res[0][0])


@pytest.mark.parametrize("depth,split", [(5, (2, 2, 1)), (5, (3, 2)), (5, (1, 1, 1, 1, 1)), (12, (4, 4, 4)), (6, (1, 5))])
def test_block_stack_in_chunks_is_bit_identical_pos_gradient_included(dev, depth, split):
    """a stack differentiated in CHUNKS (what runner_pretrain.wrap_ddp selects under multi-rank DDP: 4 blocks per host call) against the one-call
    stack and against the per-block chain, DropPath gates on, ``pos.requires_grad``: output, dx, every parameter gradient AND the folded gradient of
    the shared pos are torch.equal -- each chunk hands ``pos`` on as a second output, so the deeper chunks' sum re-enters the fold as ``dpos_in``
    and the association stays ((dx_{L-1} + dx_{L-2}) + ...) instead of (s_2 + s_1) + s_0."""
    import act_amd.composite as CP
    B, S, D, H = 3, 14, 128, 2
    torch.manual_seed(1)
    x0 = torch.randn(B, S, D, device=dev); pos0 = 0.1 * torch.randn(B, S, D, device=dev)
    gates = [(torch.floor(0.7 + torch.rand(B, device=dev)) / 0.7, torch.floor(0.7 + torch.rand(B, device=dev)) / 0.7) if l % 3 != 0 else None
             for l in range(depth)]
    dout = torch.randn(B, S, D, device=dev)
    res = []
    for mode in ("chain", "one", "chunks"):
        ps = [_params(dev, D, 4 * D, False, 10 + l) for l in range(depth)]
        x = x0.clone().requires_grad_(True)
        pos = pos0.clone().requires_grad_(True)
        if mode == "chain":
            y = x
            for l in range(depth):
                g1, g2 = gates[l] if gates[l] is not None else (None, None)
                y = CP.BlockFn.apply(y, pos, g1, g2, *ps[l], H, 1e-5, 1)
        else:
            y, pc, c0 = x, pos, 0
            for n in ((depth,) if mode == "one" else split):
                emit = c0 + n < depth
                r = CP.BlockStackFn.apply(y, pc, gates[c0:c0 + n], H, 1e-5, 1, emit, *[p for blk in ps[c0:c0 + n] for p in blk])
                y, pc = r if emit else (r, pc)
                c0 += n
        y.backward(dout)
        torch.cuda.synchronize()
        res.append([y.detach(), x.grad, pos.grad] + [p.grad for blk in ps for p in blk if p is not None])
    for other in res[1:]:
        for i, (a, b) in enumerate(zip(res[0], other)):
            assert torch.equal(a, b), i
    assert res[0][2].abs().max() > 0


def test_block_stack_falls_back_for_hooked_or_overridden_blocks(dev):
    """block_stack bypasses Module.__call__: a block with a forward hook, or a Block subclass that overrides forward, must be CALLED."""
    import act_amd.composite as CP
    from act_amd.models.act import TransformerEncoder, Block
    torch.manual_seed(0)
    enc = TransformerEncoder(embed_dim=64, depth=3, num_heads=2).to(dev)
    x = torch.randn(2, 8, 64, device=dev); pos = torch.randn(2, 8, 64, device=dev)
    y0 = enc(x, pos)
    seen = []
    h = enc.blocks[1].register_forward_hook(lambda m, i, o: seen.append(1))
    y1 = enc(x, pos)
    h.remove()
    assert seen == [1] and torch.equal(y0, y1)

    class Doubling(Block):
        def forward(self, x, pos=None, draws=None, tag="blk", gates=None):
            return 2.0 * super().forward(x, pos, draws, tag, gates)
    for b in enc.blocks:
        b.__class__ = Doubling
    y2 = enc(x, pos)
    assert not torch.equal(y0, y2)                          # the override ran
    for b in enc.blocks:
        b.__class__ = Block
    assert torch.equal(enc(x, pos), y0)


def test_block_stack_backward_follows_a_data_reassignment(dev):
    """``param.data = other`` between forward and backward moves an address without bumping a version: the backward re-derives its pointer array
    from the saved tensors (as BlockFn does), so stack and per-block chain still agree bit for bit."""
    import act_amd.composite as CP
    B, S, D, H, depth = 2, 16, 64, 2, 3
    torch.manual_seed(0)
    x0 = torch.randn(B, S, D, device=dev); dout = torch.randn(B, S, D, device=dev)
    res = []
    for stacked in (False, True):
        ps = [_params(dev, D, 4 * D, False, 10 + l) for l in range(depth)]
        x = x0.clone().requires_grad_(True)
        if stacked:
            y = CP.BlockStackFn.apply(x, None, None, H, 1e-5, 1, False, *[p for blk in ps for p in blk])
        else:
            y = x
            for l in range(depth):
                y = CP.BlockFn.apply(y, None, None, None, *ps[l], H, 1e-5, 1)
        w = ps[1][8]                                         # fc1 weight of the middle block: same values at a NEW address, old storage poisoned
        old = w.data
        w.data = old.clone()
        old.fill_(float("nan"))
        y.backward(dout)
        torch.cuda.synchronize()
        res.append([x.grad] + [p.grad for blk in ps for p in blk if p is not None])
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.isfinite(a).all() and torch.equal(a, b), i


def test_block_stack_training_trajectory_is_bit_identical(dev):
    """the tiny Stage-II model trained for 3 AdamW steps with the stack-level host calls (default) and with one BlockFn per block (ACT_BLOCK_STACK=0
    semantics), DropPath ACTIVE (rate 0.25 set on every block: the tiny YAML has 0), cross-step teacher prefetch on: losses and every parameter
    torch.equal; also in chunks of ONE block per call (depth 2: two chunks, so the pos carry between chunks is on the path), selected per model the way
    wrap_ddp does it (runner_pretrain.set_stack_chunk)."""
    import act_amd.composite as CP
    from act_amd.models import build_model_from_cfg
    from act_amd.tools import builder
    from act_amd.tools.runner_pretrain import train_step, _Single, freeze_unused_heads, set_stack_chunk
    from act_amd.utils.config import EasyDict
    from tests.golden.fill import fill_module, clouds, TINY_STAGE2, TINY_B, TINY_N
    cfg = EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                   scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=10)), step_per_update=1)
    batches = [torch.from_numpy(clouds(60 + i, TINY_B, TINY_N)).to(dev) for i in range(3)]
    saved = (CP.STACK, CP.STACK_CHUNK)
    results = []
    try:
        for stack, chunk in ((False, 0), (True, 0), (True, 1), (False, 0)):
            CP.STACK = stack
            torch.manual_seed(0)
            model = fill_module(build_model_from_cfg(EasyDict(copy.deepcopy(TINY_STAGE2))), "g4.").to(dev).train()
            model.dvae_tokenizer.prompt_dropout.p = 0.0
            for blk in list(model.ACT_encoder.blocks.blocks) + list(model.ACT_decoder.blocks):
                blk.drop_prob = 0.25
            set_stack_chunk(model, chunk)
            freeze_unused_heads(model)
            wrapped = _Single(model)
            opt, _ = builder.build_opti_sche(wrapped, cfg)
            torch.manual_seed(99)
            pts = [b.clone() for b in batches]
            losses = [train_step(wrapped, opt, pts[i], cfg, next_points=(pts[i + 1] if i + 1 < 3 else None)) for i in range(3)]
            torch.cuda.synchronize()
            results.append((torch.stack(losses).cpu(), {n: p.detach().clone().cpu() for n, p in model.named_parameters()}))
    finally:
        CP.STACK, CP.STACK_CHUNK = saved
    ref = results[3]                                         # (the first run also pays the first-use GEMM tuning, which draws random numbers)
    for got in results[1:3]:
        assert torch.equal(ref[0], got[0]), (ref[0], got[0])
        for n, p in ref[1].items():
            assert torch.equal(p, got[1][n]), n


def test_host_caches_survive_deepcopy_and_follow_the_parameters(dev):
    """the host-side caches of the composite path (leaf-module lists, the teacher's ~190-pointer ctypes struct) live in weak dictionaries OUTSIDE the
    modules: a model that has run can still be deep-copied and pickled, the copy computes with ITS OWN parameters (perturbing them changes its loss, not the
    original's), and a parameter that moves to a new address (re-assigned storage) is picked up on the next call."""
    import io
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from tests.golden.fill import fill_module, clouds, TINY_STAGE2, TINY_B, TINY_N
    torch.manual_seed(0)
    model = fill_module(build_model_from_cfg(EasyDict(copy.deepcopy(TINY_STAGE2))), "g4.").to(dev).train()
    model.dvae_tokenizer.prompt_dropout.p = 0.0
    pts = torch.from_numpy(clouds(70, TINY_B, TINY_N)).to(dev)

    from act_amd.utils.draws import Draws
    from act_amd.models.act import random_mask
    torch.manual_seed(3)
    mask = random_mask(TINY_B, 16, 12, dev).cpu()
    gumbel = -torch.empty(TINY_B, 16, 64).exponential_().log()

    def loss_of(m):                                               # every draw pinned: mask + teacher gumbel noise injected, DropPath gates from the seeded RNG
        torch.manual_seed(5)
        with torch.no_grad():
            return m(pts.clone(), draws=Draws({"mask": mask, "gumbel": gumbel}, device=dev)).item()
    loss_of(model)                                                # warm-up: first-use GEMM tuning of an unlisted shape draws random operands
    torch.manual_seed(5)
    model(pts.clone()).backward()                                 # the training path too (stack backward, saved pointer arrays)
    model.zero_grad(set_to_none=True)
    base = loss_of(model)
    assert loss_of(model) == base
    twin = copy.deepcopy(model)
    torch.save(model, io.BytesIO())
    assert loss_of(twin) == base
    with torch.no_grad():
        twin.dvae_tokenizer.visual_embed[0][0].mlp.fc1.weight.mul_(1.7)       # frozen teacher: read through the cached struct
    assert loss_of(twin) != base and loss_of(model) == base
    twin2 = copy.deepcopy(model)
    with torch.no_grad():
        twin2.ACT_encoder.blocks.blocks[0].mlp.fc1.weight.mul_(1.7)           # student stack: read through the cached leaf list
    assert loss_of(twin2) != base and loss_of(model) == base
    # a parameter whose storage is re-assigned (what module.to() / .float() do to param.data): new address, same values -> same loss, no stale pointer
    w = model.dvae_tokenizer.visual_embed[0][1].attn.qkv.weight
    old_ptr = w.data_ptr()
    w.data = w.data.clone()
    assert w.data_ptr() != old_ptr and loss_of(model) == base


def test_composite_shutdown_releases_the_fork_join_events(dev):
    """the library's only state are the fork / join events of streams that produced work for another stream: act_composite_shutdown destroys them,
    and a later call works (and gives the same bits) on fresh ones"""
    import act_amd.composite as CP
    torch.manual_seed(0)
    B, S, D, H = 4, 14, 384, 6
    x0 = torch.randn(B, S, D, device=dev); pos0 = 0.1 * torch.randn(B, S, D, device=dev); dout = torch.randn(B, S, D, device=dev)
    one = torch.ones(B, device=dev)

    def run():
        ps = _params(dev, D, 4 * D, False, 1)
        x = x0.clone().requires_grad_(True)
        y = CP.BlockFn.apply(x, pos0, one, one, *ps, H, 1e-5, 2)          # train_w = 2: weight gradients forked to the auxiliary stream
        y.backward(dout)
        return [y.detach(), x.grad] + [p.grad for p in ps if p is not None]

    a = run()
    n = CP.shutdown()
    assert n >= 1
    assert CP.shutdown() == 0
    b = run()
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert CP.shutdown() >= 1


def test_prefix_block_composite_is_bit_identical(dev):
    import act_amd.kernels as K
    import act_amd.composite as CP
    B, P, G, D, H = 3, 16, 32, 128, 2
    torch.manual_seed(1)
    x0 = torch.randn(B * G, D, device=dev); pos0 = 0.1 * torch.randn(B * G, D, device=dev); prm0 = 0.3 * torch.randn(B * P, D, device=dev)
    dout = torch.randn(B * G, D, device=dev)
    res = []
    for fn, inf in ((K.PrefixBlockFnPerKernel, K.block_forward_prefix_perkernel), (CP.PrefixBlockFn, CP.block_forward_prefix)):
        ps = [p.detach() if p is not None else None for p in _params(dev, D, 4 * D, True, 2)]
        x, pos, prm = (t.clone().requires_grad_(True) for t in (x0, pos0, prm0))
        y = fn.apply(x, pos, prm, B, P, G, *ps, H, 1e-6)
        y.backward(dout)
        with torch.no_grad():
            yi = inf(x0, pos0, prm0, B, P, G, *ps, H, 1e-6)
            n1p, _, _, _ = K.layernorm_fwd(prm0, None, ps[0], ps[1], 1e-6, want_stats=False)
            yj = inf(x0, pos0, None, B, P, G, *ps, H, 1e-6, n1p=n1p)
        torch.cuda.synchronize()
        res.append([y.detach(), x.grad, pos.grad, prm.grad, yi, yj])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.equal(res[1][0], res[1][4]) and torch.equal(res[1][4], res[1][5])


def test_encoder_composite_is_bit_identical(dev):
    import act_amd.composite as CP
    from act_amd.models.dvae import Encoder
    from tests.golden.fill import fill_module
    torch.manual_seed(2)
    nb = 0.2 * torch.randn(6, 16, 32, 3, device=dev)
    dout = torch.randn(6, 16, 96, device=dev)
    res = []
    for enabled in (False, True):
        enc = fill_module(Encoder(96), "cmp.enc.").to(dev).train()
        saved, CP.ENABLED = CP.ENABLED, enabled
        try:
            y = enc(nb); y.backward(dout)
            enc.eval()
            with torch.no_grad():
                ye = enc(nb)
        finally:
            CP.ENABLED = saved
        torch.cuda.synchronize()
        assert int(enc.first_conv[1].num_batches_tracked) == 1 and int(enc.second_conv[1].num_batches_tracked) == 1
        res.append([y.detach(), ye] + [p.grad for p in enc.parameters()] + [b.clone() for b in enc.buffers()])
    for i, (a, b) in enumerate(zip(*res)):
        if i == 1:                  # eval mode: scale/shift from running stats by torch ops vs one HIP launch (rsqrt rounding)
            assert (a - b).abs().max() <= 1e-5 * max(1.0, a.abs().max().item())
        else:
            assert torch.equal(a, b), i


def test_teacher_stack_composite_is_bit_identical(dev):
    """DGCNN inference stack + the whole prompt-tuned Transformer (prompt dropout from the in-kernel Philox, same seeds)."""
    import act_amd.composite as CP
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from tests.golden.fill import fill_module, clouds, TINY_STAGE2
    cfg = copy.deepcopy(TINY_STAGE2["dvae_config"]); cfg["NAME"] = "ACTPromptedDiscreteVAEwithVIT"
    torch.manual_seed(3)
    vae = fill_module(build_model_from_cfg(EasyDict(cfg)), "cmp.vae.").to(dev).train()
    for p in vae.parameters():
        p.requires_grad = False
    pts = torch.from_numpy(clouds(31, 3, 128)).to(dev)
    res = []
    for enabled in (False, True):
        saved, CP.ENABLED = CP.ENABLED, enabled
        vae.__dict__.pop("_rng_state", None)
        torch.manual_seed(77)                      # base seed of the Philox stream is drawn from the host RNG
        try:
            with torch.no_grad():
                nb, c = vae.group_divider(pts)
                f1 = vae.forward_tokenizer_features(nb, c)
                f2 = vae.forward_tokenizer_features(nb, c)      # second call: the device-resident counter advanced
        finally:
            CP.ENABLED = saved
        torch.cuda.synchronize()
        res.append((f1, f2))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert not torch.equal(res[0][0], res[0][1])    # dropout / gumbel noise differ between the two steps


def test_composite_gemm_shape_collection(dev):
    """act_composite_collect_begin/_end: a dry call launches nothing and reports the GEMMs of the module."""
    import ctypes
    import act_amd.composite as CP
    d = CP.BlockDims(2, 14, 64, 2, 256, 1e-5)
    n_saved = CP.lib.act_block_saved_floats(ctypes.byref(d))
    assert n_saved >= 2 * 14 * (8 * 64 + 2 * 256)
    x = torch.zeros(28, 64, device=dev); out = torch.full((28, 64), 7.0, device=dev)
    saved = torch.empty(n_saved, device=dev)
    ws = torch.empty(1 << 20, device=dev)
    w = [torch.zeros(s, device=dev) for s in ((64,), (64,), (192, 64), (192,), (64, 64), (64,), (64,), (64,), (256, 64), (256,), (64, 256), (64,))]
    prm = CP.BlockParams(*[t.data_ptr() for t in w])
    buf = (ctypes.c_int * 50)()
    assert CP.lib.act_composite_collect_begin() == 0
    rc = CP.lib.act_block_fwd_f32(ctypes.byref(d), ctypes.byref(prm), x.data_ptr(), None, None, None, 1, saved.data_ptr(), out.data_ptr(),
                                  ws.data_ptr(), ws.numel() * 4, None)
    n = CP.lib.act_composite_collect_end(buf, 10)
    torch.cuda.synchronize()
    assert rc == 0 and n == 4
    shapes = [tuple(buf[5 * i:5 * i + 5]) for i in range(4)]
    assert shapes == [(1, 1, 28, 192, 64), (1, 1, 28, 64, 64), (1, 1, 28, 256, 64), (1, 1, 28, 64, 256)]
    assert (out == 7.0).all()                       # nothing was launched
    assert CP.lib.act_composite_collect_end(buf, 10) < 0      # not collecting any more


@pytest.mark.parametrize("C,n,bs,g,k", [(384, 32, 8, 16, 4), (384, 32, 3, 64, 13), (768, 64, 2, 8, 3), (192, 32, 2, 8, 2), (128, 32, 4, 16, 4)])
def test_encoder_for_listed_groups_is_exact(dev, C, n, bs, g, k):
    """Encoder.forward(point_groups, need=idx): the last conv + max-pool run only on the k listed groups per cloud (Stage II reads the visible
    patches only, models/act.py:269-275).  The listed tokens are the bits of the full forward, the other tokens are zero, and with a gradient that is
    zero outside the listed groups every parameter gradient and the running statistics are the bits of the full forward + backward (bs * k * n is
    not a multiple of 128 in the second case: the list is padded with a repeated group; C = 192 falls back to the dense backward products)."""
    from act_amd.models.dvae import Encoder
    from tests.golden.fill import fill_module
    torch.manual_seed(11)
    nb = 0.2 * torch.randn(bs, g, n, 3, device=dev)
    need = torch.stack([torch.randperm(g, device=dev)[:k].sort().values for _ in range(bs)])
    dout = torch.zeros(bs, g, C, device=dev)
    dout.scatter_(1, need.unsqueeze(-1).expand(-1, -1, C), torch.randn(bs, k, C, device=dev))
    res = []
    for listed in (False, True):
        enc = fill_module(Encoder(C), "need.enc.").to(dev).train()
        y = enc(nb, need=need if listed else None)
        y.backward(dout)
        torch.cuda.synchronize()
        res.append(dict([("y", y.detach())] + [(kk, p.grad) for kk, p in enc.named_parameters()] + [("buf." + kk, b.clone().float()) for kk, b in enc.named_buffers()]))
    full, part = res
    picked = lambda t: torch.gather(t, 1, need.unsqueeze(-1).expand(-1, -1, C))
    assert torch.equal(picked(part["y"]), picked(full["y"]))
    rest = torch.ones(bs, g, dtype=torch.bool, device=dev).scatter_(1, need, False)
    assert (part["y"][rest] == 0).all()
    for kk in full:
        if kk != "y":
            assert torch.equal(part[kk], full[kk]), kk


@pytest.mark.parametrize("C,n,bs,g", [(384, 32, 8, 16), (128, 32, 4, 64), (192, 64, 2, 24), (768, 64, 2, 8)])
def test_encoder_fused_schedule_matches_plain(dev, C, n, bs, g):
    """mini-PointNet with BatchNorm statistics / apply + ReLU / max-pool fused into the GEMMs (csrc/composite.hip, fused schedule) against
    the one-kernel-per-layer host path: forward, running statistics, every parameter gradient (train mode) and the eval-mode forward.
    Not bit-identical by construction (other summation order of the statistics): 1e-5 relative to the largest element."""
    import act_amd.composite as CP
    from act_amd.models.dvae import Encoder
    from tests.golden.fill import fill_module
    torch.manual_seed(4)
    nb = 0.2 * torch.randn(bs, g, n, 3, device=dev)
    dout = torch.randn(bs, g, C, device=dev)
    res = []
    for enabled in (False, True):
        enc = fill_module(Encoder(C), "fz.enc.").to(dev).train()
        saved, CP.ENABLED = CP.ENABLED, enabled
        try:
            y = enc(nb); y.backward(dout)
            enc.eval()
            with torch.no_grad():
                ye = enc(nb)
        finally:
            CP.ENABLED = saved
        torch.cuda.synchronize()
        res.append(dict([("y", y.detach()), ("y_eval", ye)] + [(k, p.grad) for k, p in enc.named_parameters()] +
                        [("buf." + k, b.clone().float()) for k, b in enc.named_buffers()]))
    rel = lambda a, b: ((a.double() - b.double()).abs().max() / max(1.0, b.double().abs().max().item())).item()
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    for k, a in res[0].items():
        if k in ("y", "y_eval") or k.startswith("buf."):
            assert rel(res[1][k], a) <= 2e-5, (k, rel(res[1][k], a))          # forward values and running statistics: tight
        elif k in ("first_conv.0.bias", "first_conv.3.bias", "second_conv.0.bias"):
            # a conv bias in front of a train-mode BatchNorm has an exactly-zero gradient (first_conv.3.bias too: a constant shift of h2 and
            # of its group max moves h3 by a constant, which BatchNorm-2 removes): both paths return cancellation noise of ~1e-7 x the summands
            assert rel(res[1][k], a) <= 2e-3, (k, rel(res[1][k], a))
        else:
            # gradients: element-wise 2e-5 unless a max-pool winner / ReLU sign flips between the two summation orders of the statistics
            # (then one row of dh is rerouted and every upstream weight moves a little): bounded in the L2 sense
            assert rel(res[1][k], a) <= 2e-5 or l2(res[1][k], a) <= 5e-3, (k, rel(res[1][k], a), l2(res[1][k], a))
