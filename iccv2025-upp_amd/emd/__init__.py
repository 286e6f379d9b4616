"""Top-level `emd` package: the reference does `import emd` with extensions/ on sys.path
(models/Point_MAE_unify.py:18), i.e. extensions/emd/__init__.py:1."""
from extensions.emd.emd import earth_mover_distance as emd  # noqa: F401
from extensions.emd.emd import earth_mover_distance, EarthMoverDistanceFunction  # noqa: F401

__all__ = ['emd']
