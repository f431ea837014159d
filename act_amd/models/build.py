"""MODELS registry + build_model_from_cfg (reference: models/build.py:1-15)."""
from ..utils import registry

MODELS = registry.Registry("models")


def build_model_from_cfg(cfg, **kwargs):
    """cfg: dict-like with key NAME -> instance of the registered class, constructed as cls(cfg)."""
    return MODELS.build(cfg, **kwargs)
