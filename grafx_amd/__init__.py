"""grafx_amd — MI355X-native batched audio-graph renderer.

Drop-in for the hot path of sh-lee97/grafx: the per-type batched processor
forwards called from ``render_grafx``.  Graph objects, scheduling and routing
(`data`, `render`, `utils`) are plain Python mirroring the reference API; the
processors in `processors` run hand-written gfx950 HIP kernels through the
C-ABI library ``libgrafx_amd.so`` (see include/grafx_amd.h).
"""
from . import data, render, utils  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # processors import the native library lazily so that graph-only use works without it
    if name in ("processors", "ops"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
