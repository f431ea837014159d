from .build import build_dataset_from_cfg, DATASETS  # noqa: F401
from . import SyntheticDataset  # noqa: F401
from . import FinetuneDatasets  # noqa: F401
