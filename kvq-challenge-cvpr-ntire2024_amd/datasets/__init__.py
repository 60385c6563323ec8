from .fusion_datasets import *  # noqa: F401,F403
from .fusion_datasets import SyntheticKVQDataset, UnifiedFrameSampler, get_spatial_fragments  # noqa: F401
