"""simple_spectral_amd -- MI355X-native core of simple-spectral's per-pixel spectral integrator.

Python is plumbing here: the product is libssx_hip.so (hand-written HIP for gfx950 behind the C
ABI of include/ssx.h) plus the C++ host library libssx_host.so.  This package mirrors the
reference's Renderer interface (src/renderer.hpp:16-34,75-81) on top of that ABI so tests and
bench.py read like the reference's own driver code.
"""
from .renderer import Options, Renderer, Scene, SsxError  # noqa: F401

__all__ = ["Options", "Renderer", "Scene", "SsxError"]
