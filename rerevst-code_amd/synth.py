"""Seeded synthetic frames / style images (BASELINE.json configs use synthetic video).

Frames are uint8 BGR.  `kind="noise"` is the SURVEY §8(d) definition
(default_rng(1000+i).integers(0,256)); `kind="smooth"` mixes a bilinear-upsampled
low-resolution field with fine noise so the content has image-like spatial structure
(used for parity cases so activations are not all clamped).
"""
import numpy as np


def _smooth_field(rng, H, W, cell):
    gh, gw = H // cell + 2, W // cell + 2
    g = rng.random((gh, gw, 3), dtype=np.float32)
    ys = np.arange(H, dtype=np.float32) / cell
    xs = np.arange(W, dtype=np.float32) / cell
    y0, x0 = ys.astype(np.int64), xs.astype(np.int64)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
    b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
    return a * (1 - fy) + b * fy


def synth_frame(i, H, W, kind="noise", seed=1000):
    rng = np.random.default_rng(seed + i)
    if kind == "noise":
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    f = 0.75 * _smooth_field(rng, H, W, 16) + 0.25 * rng.random((H, W, 3), dtype=np.float32)
    return np.clip(f * 255.0, 0, 255).astype(np.uint8)


def synth_style(H=512, W=512, kind="noise", seed=7):
    return synth_frame(0, H, W, kind=kind, seed=seed)
