"""dither_pie_amd -- MI355X-native (gfx950) backend for dither_pie's per-pixel hot path.

Host code keeps frames in PyTorch-ROCm uint8 tensors and calls hand-written HIP kernels through
the C ABI declared in include/ditherpie_hip.h (libditherpie_hip.so, built in-tree).
"""
from ._lib import DitherPieError, build, load  # noqa: F401

__version__ = "0.1.0"
