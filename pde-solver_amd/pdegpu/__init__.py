"""pdegpu — MI355X-native structured-grid FEM solve path behind the PDE-Solver tool API.

Host side (Python, mirrors the reference's operator interface) -> ctypes -> libpdegpu.so (C-ABI,
include/pdegpu.h) -> hand-written HIP kernels for gfx950.  No CPU fallback."""
from .fields import PlotResult, SolveResult, TimeSeriesField  # noqa: F401
from ._lib import PdeGpuError, load_library  # noqa: F401

__all__ = ["TimeSeriesField", "SolveResult", "PlotResult", "PdeGpuError", "load_library"]
