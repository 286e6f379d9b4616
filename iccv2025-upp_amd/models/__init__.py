from .build import build_model_from_cfg, MODELS  # noqa: F401

from . import Point_MAE_unify  # noqa: F401  (registers Point_MAE_unify)
from . import Point_MAE_unify_segment  # noqa: F401,E402  (registers Point_MAE_unify_seg)
from . import Point_MAE  # noqa: F401,E402  (registers Point_MAE)
from . import Point_MAE_pretask_dev  # noqa: F401,E402  (registers Point_MAE_pretask_dev)
