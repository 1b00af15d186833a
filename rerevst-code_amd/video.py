"""Driver-side logic of the reference's test/generate_real_video.py, restated for the
drop-in: reflect padding / cropping (ReshapeTool, :61-83, :167), the global-feature
sampling schedule (:129-148) and the frame -> rank sharding used for multi-GPU runs.
Host-side, negligible cost; numpy only.
"""
import numpy as np


def padded_size(n):
    """ReshapeTool.process (:66-76): n+128 rounded up to a multiple of 64."""
    m = n + 128
    if m % 64 != 0:
        m += 64 - m % 64
    return m


class ReshapeTool():
    """Same behaviour as the reference class: the padded size is fixed by the first frame."""

    def __init__(self):
        self.record_H = 0
        self.record_W = 0

    def process(self, img):
        H, W, C = img.shape
        if self.record_H == 0 and self.record_W == 0:
            self.record_H, self.record_W = padded_size(H), padded_size(W)
        return reflect_pad(img, self.record_H, self.record_W)


def reflect_pad(img, PH, PW):
    """cv2.copyMakeBorder(img, 64, PH-64-H, 64, PW-64-W, cv2.BORDER_REFLECT) (:81-82).
    BORDER_REFLECT repeats the edge pixel (fedcba|abcdefgh|hgfedcb) = numpy 'symmetric'."""
    H, W = img.shape[:2]
    return np.pad(img, ((64, PH - 64 - H), (64, PW - 64 - W), (0, 0)), mode="symmetric")


def sample_indices(frame_num, interval=8):
    """Frames fed to add() (:129-143): s*interval for s < (frame_num-1)//interval, then the last."""
    return [s * interval for s in range((frame_num - 1) // interval)] + [frame_num - 1]


def sample_indices_multistyle(frame_num, interval=16):
    """VideoStylization.SeqNormPrePare ("Multi-style Interpolation/test.py":72-85): cached features
    s*interval for s < (frame_num-1)//interval + 1, then the last frame — AGAIN when it was already sampled."""
    return [s * interval for s in range((frame_num - 1) // interval + 1)] + [frame_num - 1]


def ramp_weights(i, frame_num, n_styles=2, blend="pair"):
    """Per-frame style weights of the multi-style driver loop (test.py:127-131): for two styles exactly the
    reference's [w, 1-w] with w = i/(frame_num-1).  The reference only ever ramps TWO styles; for more this build chains
    the same ramp through the styles in reverse order (the reference ramps from style 1 towards style 0): the video
    starts on the last style and ends on style 0.  blend="pair": two neighbouring styles at a time (piecewise linear);
    blend="all": a smooth partition of unity (normalised Gaussian bumps of width 1.5 styles around the same position),
    so EVERY style has a non-zero weight in every frame — what bench.py's 4-style configuration uses.  Weights sum to 1."""
    w = i / (frame_num - 1.0) if frame_num > 1 else 1.0
    if n_styles == 1:
        return [1.0]
    if n_styles == 2:
        return [w, 1.0 - w]
    pos = (1.0 - w) * (n_styles - 1)           # 0 -> style 0, n_styles-1 -> the last style
    if blend == "all":
        b = [float(np.exp(-((pos - k) / 1.5) ** 2)) for k in range(n_styles)]
        t = sum(b)
        return [v / t for v in b]
    lo = min(int(pos), n_styles - 2)
    f = pos - lo
    out = [0.0] * n_styles
    out[lo], out[lo + 1] = 1.0 - f, f
    return out


def resize_bilinear(img, size):
    """cv2.resize(img, (w, h)) with the default INTER_LINEAR geometry (half-pixel centres, edge clamp) used for the
    style images (test.py:53: 384x384).  cv2 evaluates it in 11-bit fixed point, so single grey levels may differ."""
    w, h = size
    H, W = img.shape[:2]
    ys = np.clip((np.arange(h) + 0.5) * (H / h) - 0.5, 0, H - 1)
    xs = np.clip((np.arange(w) + 0.5) * (W / w) - 0.5, 0, W - 1)
    y0, x0 = np.floor(ys).astype(np.int64), np.floor(xs).astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    f = img.astype(np.float64)
    top = f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx
    bot = f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx
    return np.clip(np.rint(top * (1 - fy) + bot * fy), 0, 255).astype(np.uint8)


def shard_range(frame_num, rank, world):
    """Contiguous block of frames owned by `rank`."""
    lo = frame_num * rank // world
    hi = frame_num * (rank + 1) // world
    return lo, hi


def stylize_video(model, frames, style, rank=0, world=1, broadcast=None, interval=8, chunk=32):
    """generate_real_video.py main flow on one rank of `world`.

    `frames`: list of uint8 BGR HWC arrays (the whole video; only the sampled frames and
    this rank's shard are touched).  `broadcast(blob, src)` ships the state blob from rank 0
    (an RCCL broadcast in bench.py / dist.py; None for single GPU).  Returns
    {frame index: float32 BGR HWC stylized frame cropped back to the input size}.
    """
    n = len(frames)
    if rank == 0:
        model.prepare_style(style)
        model.clean()
        for i in sample_indices(n, interval):
            model.add(frames[i])                 # unpadded, as the reference does
        model.compute()
        blob = model.get_state()
    else:
        blob = None
    if world > 1:
        blob = broadcast(blob, 0)
        if rank != 0:
            model.set_state(blob)
    tool = ReshapeTool()
    lo, hi = shard_range(n, rank, world)
    out = {}
    on_device = getattr(model, "transfer_frames", None)     # pad / crop inside the first / last kernel
    same = all(f.shape == frames[lo].shape for f in frames[lo:hi]) if hi > lo else False
    if on_device is not None and same and getattr(model, "use_Global", True):
        for c0 in range(lo, hi, chunk):
            idx = list(range(c0, min(hi, c0 + chunk)))
            styled = on_device([frames[i] for i in idx])
            for j, i in enumerate(idx):
                out[i] = styled[j]
        return out
    # the host-buffer batch entry pipelines sub-batches (copy in / kernels / copy out) inside one call
    batch = getattr(model, "transfer_batch", None)     # a model with only the reference's per-frame transfer() works too
    if batch is None:
        batch = lambda fs: np.stack([model.transfer(f) for f in fs])
    for c0 in range(lo, hi, chunk):
        idx = list(range(c0, min(hi, c0 + chunk)))
        styled = batch([tool.process(frames[i]) for i in idx])
        for j, i in enumerate(idx):
            H, W, _ = frames[i].shape
            out[i] = styled[j, 64:64 + H, 64:64 + W, :]
    return out


def stylize_video_multistyle(model, frames, styles, weights_of=None, interval=16, style_size=(384, 384)):
    """"Multi-style Interpolation/test.py" main flow (VideoStylization :40-111 + the driver loop :114-131) on a model
    with the multi-style call surface (MultiStyleStylization, or the oracle's MultiStylization):
    styles resized to 384x384 (:53) -> prepare_style; every frame padded (ChangeShapeTool) and encoded ONCE, the
    feature cached (:87-101; in HBM here, on disk in the reference); every `interval`-th cached feature plus the last
    one again -> add_patch -> compute_norm (:72-85); then per frame the decoder alone with the blended state of
    weight vector weights_of(i, n) (default: the reference's ramp) and the crop (:110).
    Returns {frame index: float32 BGR HWC stylized frame}."""
    n = len(frames)
    S = len(styles)
    if weights_of is None:
        weights_of = lambda i, num: ramp_weights(i, num, S)
    model.prepare_style([resize_bilinear(s, style_size) if style_size and tuple(s.shape[:2]) != tuple(style_size[::-1]) else s for s in styles])
    tool = ReshapeTool()
    feats = [model.generate_content_features(tool.process(f)) for f in frames]
    model.clean()
    for i in sample_indices_multistyle(n, interval):
        model.add_patch(feats[i])
    model.compute_norm()
    out = {}
    for i, f in enumerate(frames):
        H, W, _ = f.shape
        out[i] = model.transfer(feats[i], weights_of(i, n))[64:64 + H, 64:64 + W, :]
    return out
