"""fyrox_amd -- MI355X (gfx950) implementation of Fyrox's per-frame skeletal-animation hot path:
pose sampling/blending -> bone palette -> 4-weight linear-blend skinning.

The product is libfyrox_hip.so (hand-written HIP kernels behind the C ABI in include/fyrox_hip.h);
this package is the thin host-side binding used by tests and bench.py.  There is no CPU fallback.
"""
from ._native import FyxError, NativeLibraryMissing, LIB_PATH, exported_symbols  # noqa: F401
from .context import Context, DeviceBuffer  # noqa: F401
from . import synth  # noqa: F401

__all__ = ["Context", "DeviceBuffer", "FyxError", "NativeLibraryMissing", "synth", "LIB_PATH"]
