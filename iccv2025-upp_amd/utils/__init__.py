"""`utils` surface of the reference that sits on the hot path (misc.fps, registry, dist_utils, config)."""
