"""MI355X (gfx950) implementation of the Neural-Volume-Super-Resolution rendering hot path.

Host-side mirror of the reference's Python surface for this path: the modules carry the reference's
module names (nerf_helpers, volume_rendering_utils, train_utils, models) and export the same function / class names with the
same argument order, so `train_nerf.py`-style callers switch by changing imports.  All arithmetic runs in the hand-written
HIP kernels of csrc/ behind the C ABI of include/nvsr.h (bound with ctypes in capi.py); PyTorch only owns device memory and
streams.  There is no CPU fallback: without a GPU or without the built library every entry point raises.
"""
from . import capi  # noqa: F401
from . import nerf_helpers, volume_rendering_utils, train_utils, models, distributed, plane_store, training  # noqa: F401
from . import load_blender, load_llff  # noqa: F401
from .build import build_extension  # noqa: F401

__all__ = ["capi", "nerf_helpers", "volume_rendering_utils", "train_utils", "models", "distributed", "plane_store", "training", "load_blender", "load_llff", "build_extension"]
