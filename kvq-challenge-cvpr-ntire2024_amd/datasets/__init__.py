from .fusion_datasets import *  # noqa: F401,F403
from .fusion_datasets import (KVQ_MEAN, KVQ_STD, SIMPLEVQA_MEAN, SIMPLEVQA_STD, SyntheticKVQDataset,  # noqa: F401
                              SyntheticSimpleVQADataset, UnifiedFrameSampler, NpyFrameReader, ViewDecompositionDataset_KVQ,
                              ViewDecompositionDataset_add_forSimpleVQA, open_video, get_resizecrop_video,
                              get_resized_video, get_single_view, get_spatial_fragments)
