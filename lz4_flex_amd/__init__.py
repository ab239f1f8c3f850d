"""lz4_flex_amd -- MI355X-native LZ4 block codec behind the lz4_flex API surface.

`lz4_flex_amd.block` mirrors lz4_flex::block, `lz4_flex_amd.frame` mirrors lz4_flex::frame; both are
thin ctypes views over the C ABI in include/lz4flex_amd.h, whose compute is hand-written HIP kernels
for gfx950 (lz4_flex_amd/csrc).  Importing the sub-modules loads the shared library and fails loudly
if it has not been built."""
from . import _lib  # noqa: F401

__all__ = ["block", "frame"]
__version__ = "0.1.0"


def __getattr__(name):
    if name in ("block", "frame"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
