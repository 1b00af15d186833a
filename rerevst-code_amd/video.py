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
