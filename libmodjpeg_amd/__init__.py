"""libmodjpeg_amd — MI355X-native (gfx950) implementation of libmodjpeg's per-8x8-block hot path.

Layers:
    csrc/        hand-written HIP kernels + the C ABI of include/mjx.h  -> libmjx_hip.so
    host/        C shim exporting the libmodjpeg.h API on top of libjpeg + libmjx_hip.so
    api.py       Python mirror of the reference's operator interface for this path
                 (mj_compose_with_mask, mj_convolve, mj_effect_*), used by tests and bench
    batch.py     batches of independent images (device-resident planes, launch plans, sharding)
    entropy.py   ctypes mirror of the baseline Huffman coders (scan data <-> coefficient planes)
"""
from . import _lib  # noqa: F401
from .api import (MJ_OK, MJ_ERR_MEMORY, MJ_ERR_NULL_DATA, MJ_ERR_UNSUPPORTED_COLORSPACE,  # noqa: F401
                  JCS_GRAYSCALE, JCS_RGB, JCS_YCbCr, Jpeg, CompiledDropon, Component,
                  mj_compose_with_mask, mj_convolve, mj_effect_grayscale, mj_effect_pixelate,
                  mj_effect_tint, mj_effect_luminance)

__all__ = ["Jpeg", "CompiledDropon", "Component", "mj_compose_with_mask", "mj_convolve", "mj_effect_grayscale",
           "mj_effect_pixelate", "mj_effect_tint", "mj_effect_luminance"]
