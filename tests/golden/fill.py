"""Deterministic weight / input fillers shared by the golden generator and the tests
(no weight blobs are committed: every tensor is regenerated from its name)."""
import zlib
import numpy as np
import torch


def _rs(name):
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)   # legacy stream: version-stable


def fill_tensor(name, shape, kind=None):
    """kind: 'w' (fan-in scaled), 'norm_w' (1 + small), 'b' (small), 'tok' (0.02 scale)."""
    shape = tuple(shape)
    x = _rs(name).standard_normal(shape).astype(np.float32)
    if kind is None:
        leaf = name.split(".")[-1]
        if len(shape) <= 1:
            kind = "norm_w" if leaf == "weight" else "b"
        elif "token" in name or "prompt" in name or "cls_pos" in name:
            kind = "tok"
        elif name.endswith("codebook"):
            kind = "code"
        else:
            kind = "w"
    if kind == "w":
        fan_in = int(np.prod(shape[1:]))
        x *= np.float32(1.0 / np.sqrt(fan_in))
    elif kind == "norm_w":
        x = (1.0 + 0.1 * x).astype(np.float32)
    elif kind == "b":
        x *= np.float32(0.05)
    elif kind == "tok":
        x *= np.float32(0.2)
    elif kind == "code":
        pass
    return torch.from_numpy(x)


@torch.no_grad()
def fill_module(module, prefix=""):
    """overwrite every parameter (not buffers) of ``module`` by name."""
    for name, p in module.named_parameters():
        p.copy_(fill_tensor(prefix + name, p.shape))
    return module


def clouds(seed, B, N):
    """pc_norm'd gaussian clouds (datasets/ShapeNet55Dataset.py:45-51 semantics)."""
    x = _rs(f"clouds{seed}").standard_normal((B, N, 3)).astype(np.float32)
    x = x - x.mean(axis=1, keepdims=True)
    m = np.sqrt((x ** 2).sum(axis=2)).max(axis=1)
    return (x / m[:, None, None]).astype(np.float32)


TINY_STAGE2 = dict(
    NAME="ACT_PointDistillation", loss="cosine",
    transformer_config=dict(mask_ratio=0.75, mask_type="rand", proj="linear", embed_dim=64, encoder_dims=64,
                            depth=2, drop_path_rate=0.0, cls_dim=32, replace_pob=0.0, num_heads=2,
                            decoder_depth=2, decoder_num_heads=2, return_all_tokens=False, cls_loss=False,
                            register_shallow_hook=9),
    dvae_config=dict(visual_embed_type="vit_base_patch16_384", visual_embed_dim=128, visual_embed_pos="after_dgcnn1",
                     freeze_visual_embed=True, num_prompt_token=8, use_deep_prompt=True, num_group=16,
                     group_size=8, encoder_dims=64, num_tokens=64, tokens_dims=64, decoder_dims=64, ckpt="none",
                     visual_embed_depth=2, visual_embed_heads=2),
)
TINY_B, TINY_N = 2, 128
# non-default configurations of ACTPromptedDiscreteVAEwithVIT (models/dvae.py:513-534), as overrides of TINY_STAGE2["dvae_config"]
PROMPT_VARIANTS = {"shallow": dict(use_deep_prompt=False), "noprompt": dict(use_deep_prompt=False, num_prompt_token=0),
                   "novit": dict(use_deep_prompt=False, num_prompt_token=0, visual_embed_dim="none")}
# the plain Point-BERT tokenizer (models/dvae.py:278-358; keys of cfgs/autoencoder/pointbert_dvae.yaml `model:`)
TINY_DVAE = dict(NAME="DiscreteVAE", group_size=8, num_group=16, num_tokens=64, encoder_dims=64, tokens_dims=64, decoder_dims=64)

# ACT_PointBERT (models/act.py:913-1096): the reference ships no YAML for it; keys as the constructor reads them
TINY_POINTBERT = dict(
    NAME="ACT_PointBERT", m=0.9, T=0.07, K=16,
    transformer_config=dict(mask_ratio=[0.25, 0.45], mask_type="rand", embed_dim=64, encoder_dims=64, depth=2, drop_path_rate=0.0, cls_dim=32,
                            replace_pob=0.2, num_heads=2, moco_loss=True, dvae_loss=True, cutmix_loss=True, return_all_tokens=False),
    dvae_config=dict(TINY_STAGE2["dvae_config"]),
)

TINY_FINETUNE = dict(NAME="PointTransformer", embed_dim=64, depth=2, drop_path_rate=0.0, cls_dim=10, num_heads=2,
                     group_size=8, num_group=16, encoder_dims=32, transfer_type="full")
TINY_FT_LABELS = [3, 1, 4, 1]
