#!/usr/bin/env python
"""Writes the synthetic latitude-longitude environment map used by the env-map fixtures
(psdr-cuda_amd/data/envmaps/synthetic_sky_64x32.exr): a smooth sky gradient, a warm ground, a bright
"sun" lobe and a dimmer coloured lobe (both away from the u = 0|1 seam, where Bitmap::eval does not wrap)
-- HDR, strictly positive, analytic (no photograph)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "psdr-cuda_amd"))
from psdr_cuda.exr import save_exr_rgb  # noqa: E402

W, H = 64, 32
u = (np.arange(W) + 0.5) / W
v = (np.arange(H) + 0.5) / H
phi, theta = np.meshgrid(u * 2 * np.pi, v * np.pi)                  # [H, W]
d = np.stack([np.sin(phi) * np.sin(theta), np.cos(theta), -np.cos(phi) * np.sin(theta)], -1)


def lobe(direction, sharp):
    direction = np.asarray(direction, dtype=np.float64)
    direction /= np.linalg.norm(direction)
    return np.exp(sharp * ((d * direction).sum(-1) - 1.0))


up = np.clip(d[..., 1], 0, 1)[..., None]
down = np.clip(-d[..., 1], 0, 1)[..., None]
img = 0.15 + up * np.array([0.25, 0.45, 0.9]) + down * np.array([0.35, 0.25, 0.15])
img = img + lobe([-0.4, 0.7, 0.6], 25.0)[..., None] * np.array([30.0, 26.0, 18.0])
img = img + lobe([-0.8, 0.2, 0.5], 6.0)[..., None] * np.array([1.0, 2.5, 1.5])
out = os.path.join(ROOT, "psdr-cuda_amd", "data", "envmaps", "synthetic_sky_64x32.exr")
save_exr_rgb(out, img.astype(np.float32))
print(out, img.min(), img.max())
