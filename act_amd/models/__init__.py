from .build import build_model_from_cfg, MODELS  # noqa: F401
from . import dvae  # noqa: F401
from . import act  # noqa: F401
from . import bert  # noqa: F401
