"""kvq_amd — MI355X-native KSVQE / SimpleVQA per-video forward path.

Loaded under the module name ``kvq_amd`` by the repo-root shim ``kvq_amd.py`` (this
directory's contract name contains hyphens).  Layout:

  csrc/       hand-written HIP kernels for gfx950 + the C-ABI library (libkvq_hip.so)
  _abi.py     ctypes binding of include/kvq_hip.h (fails loudly when the .so is missing)
  models/     host-side mirror of the reference's models/model.py + models/head.py API
  datasets/   fragment / frame samplers mirroring datasets/fusion_datasets.py
  utils/      procedural weights + clips, metrics
"""
__version__ = "0.1.0"
