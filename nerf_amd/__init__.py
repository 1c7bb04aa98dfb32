"""nerf_amd: MI355X-native (gfx950) render-and-train hot path behind the reference's Python surface.

Modules mirror the reference files they stand in for: nets.py, renderer.py, helpers.py,
trainer.py, inference.py.  All arithmetic runs in libnerf_hip.so (nerf_amd/csrc, include/nerf_hip.h)."""
from . import _lib  # noqa: F401

__all__ = ['nets', 'renderer', 'helpers', 'trainer', 'inference', 'optim', 'parallel', 'config']
