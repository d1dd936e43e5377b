"""Seeded procedural RGB8 textures (SURVEY.md section 8(d) "synthetic inputs (iii)").

The reference's -srgb scenes open data/scenes/crystal-lizard-4096.png (src/scene.cpp:292,357), a
48 MiB texture that is missing from its repository; a 4096^2 procedural texture reproduces its
memory footprint (beyond the 32 MiB of aggregate L2).  Same bytes as
ssx::procedural_texture (host/image_io.cpp), which the CLI uses for `--texture=procedural:N[:SEED]`.
"""
import numpy as np


def procedural_texture(n=4096, seed=1):
    """uint8 [n, n, 3], rows top to bottom."""
    x = np.arange(n, dtype=np.uint32)[None, :]
    y = np.arange(n, dtype=np.uint32)[:, None]
    with np.errstate(over="ignore"):
        h = (x * np.uint32(0x9E3779B1)) ^ (y * np.uint32(0x85EBCA77)) ^ np.uint32((seed * 0xC2B2AE3D) & 0xFFFFFFFF)
        h ^= h >> np.uint32(15); h *= np.uint32(0x2C1B3C6D); h ^= h >> np.uint32(12); h *= np.uint32(0x297A2D39); h ^= h >> np.uint32(15)
    base = np.where((((x >> np.uint32(5)) ^ (y >> np.uint32(5))) & np.uint32(1)) != 0, np.uint32(200), np.uint32(60))
    out = np.empty((n, n, 3), dtype=np.uint8)
    for c in range(3):
        out[..., c] = ((np.uint32(3) * base + ((h >> np.uint32(8 * c)) & np.uint32(0xFF))) >> np.uint32(2)).astype(np.uint8)
    return out


def resolve(spec):
    """'procedural:N[:SEED]' -> ndarray, anything else unchanged (a path / None / an ndarray)."""
    if isinstance(spec, str) and spec.startswith("procedural:"):
        parts = spec.split(":")
        return procedural_texture(int(parts[1]), int(parts[2]) if len(parts) > 2 else 1)
    return spec
