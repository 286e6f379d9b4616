from .emd import earth_mover_distance as emd  # noqa: F401

__all__ = ['emd']
