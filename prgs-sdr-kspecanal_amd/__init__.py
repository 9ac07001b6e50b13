"""MI355X-native overlapped windowed-FFT spectrum / waterfall engine (kspecanal-compatible).

Python host -> ctypes -> libksa.so (hand-written HIP for gfx950).  Importing this package loads
the library and raises if it is missing: there is no CPU path in the product.
"""
from ._lib import KsaError, lib, LIB_PATH, FMT_C64, FMT_U8, OUT_LINEAR, OUT_DB, OUT_DB_CLIP, HM_ROWS
from .engine import (SpectrumEngine, PinnedBuffer, allreduce_state, scan_allstitch, scan_gather_state, full_size_for,
                     window_starts, window_table, heatmap_width)

__all__ = ["KsaError", "SpectrumEngine", "lib", "LIB_PATH", "FMT_C64", "FMT_U8", "OUT_LINEAR", "OUT_DB",
           "OUT_DB_CLIP", "HM_ROWS", "PinnedBuffer", "allreduce_state", "scan_allstitch", "scan_gather_state", "full_size_for", "window_starts", "window_table", "heatmap_width"]
