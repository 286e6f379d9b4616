"""models.build: the registry the reference's builder uses (reference models/build.py:1-17,
tools/builder.py:8,33-35)."""
from utils import registry

MODELS = registry.Registry('models')


def build_model_from_cfg(cfg, **kwargs):
    """cfg.NAME selects the registered class; returns cls(cfg)."""
    return MODELS.build(cfg, **kwargs)
