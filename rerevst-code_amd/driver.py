"""Command-line driver: the flow of the reference's test/generate_real_video.py (:86-186) on top of the HIP path.

    python -m rerevst-code_amd.driver --style style.jpg --frames 'video/*.png' --checkpoint style_net-TIP-final.pth \
           --out result_frames/name [--video result.avi --fps 24] [--no-global]

Reads the style image and the content frames (PNG / JPEG, through Pillow; BGR uint8 arrays as cv2.imread gives the
reference), runs prepare_style / clean / add (every 8th frame + the last, :129-143) / compute, stylizes every frame with
pad + crop on the device (`Stylization.transfer_frames`), writes the stylized frames under the input file names
(saturating float -> uint8 as cv2.imwrite does, :170-171) and optionally a Motion-JPEG AVI (:175-186).

Differences from the reference script, on purpose: the frame list is sorted (the reference's comment asks for it, its
code forgets the sort()); image decoding / encoding is Pillow's, not OpenCV's (neither is part of the measured path).
"""
import argparse
import glob
import io
import os
import struct
import sys

import numpy as np


def _pil():
    try:
        from PIL import Image
    except ImportError as e:                       # no silent fallback: the driver needs an image codec
        raise RuntimeError("the driver needs Pillow for PNG/JPEG I/O") from e
    return Image


def read_image_bgr(path):
    """uint8 [H][W][3] in BGR order, like cv2.imread(path) (generate_real_video.py:54-55)."""
    with _pil().open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def image_size(path):
    """(H, W) from the file's header, without decoding the pixels."""
    with _pil().open(path) as im:
        w, h = im.size
    return h, w


def to_uint8(img):
    """cv2.imwrite's conversion of a float image: round to nearest, saturate to 0..255."""
    if img.dtype == np.uint8:
        return img
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def write_image_bgr(path, img):
    """Write a BGR image (uint8, or float in 0..255) as cv2.imwrite(path, img) would; format from the extension."""
    rgb = np.ascontiguousarray(to_uint8(img)[:, :, ::-1])
    if path.lower().endswith(".png"):
        _pil().fromarray(rgb, "RGB").save(path, compress_level=1)      # cv2.imwrite's default IMWRITE_PNG_COMPRESSION is 1 (best speed)
    else:
        _pil().fromarray(rgb, "RGB").save(path)


def list_frames(pattern):
    """Sorted frame list of glob pattern `pattern` (generate_real_video.py:23-25, 102)."""
    frames = sorted(glob.glob(pattern))
    if not frames:
        raise FileNotFoundError("no content frames match %r" % pattern)
    return frames


class MJPGWriter:
    """Minimal Motion-JPEG AVI muxer (cv2.VideoWriter(..., fourcc('M','J','P','G'), fps, (W, H)), :179-186):
    RIFF 'AVI ' = hdrl (avih + one video stream: strh/strf) + movi ('00dc' JPEG chunks) + idx1."""

    def __init__(self, path, fps, width, height, quality=95):
        self.path, self.fps, self.w, self.h, self.q = path, float(fps), int(width), int(height), int(quality)
        self.f = open(path, "wb")
        self.index = []          # (offset inside movi, size)
        self.f.write(b"\0" * self._header_size())          # header is written at close, when the frame count is known
        self.movi_start = self.f.tell()
        self.f.write(b"LIST" + struct.pack("<I", 0) + b"movi")

    @staticmethod
    def _header_size():
        return 12 + (8 + 4 + (8 + 56) + (8 + 4 + (8 + 56) + (8 + 40)))

    def write(self, frame_bgr):
        self.write_jpeg(_encode_jpeg(frame_bgr, self.q), frame_bgr.shape)

    def write_jpeg(self, data, shape):
        """Append one already encoded frame (the encoders run on worker threads; the muxer only appends, in order)."""
        if shape[0] != self.h or shape[1] != self.w:
            raise ValueError("frame size %r does not match the video size %r" % (tuple(shape[:2]), (self.h, self.w)))
        off = self.f.tell() - (self.movi_start + 8)
        self.f.write(b"00dc" + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b""))
        self.index.append((off, len(data)))

    def release(self):
        n = len(self.index)
        movi_end = self.f.tell()
        idx = b"".join(b"00dc" + struct.pack("<III", 0x10, off, size) for off, size in self.index)
        self.f.write(b"idx1" + struct.pack("<I", len(idx)) + idx)
        end = self.f.tell()
        usec = int(round(1e6 / self.fps))
        max_bytes = max((s for _, s in self.index), default=0)
        avih = struct.pack("<IIIIIIIIIIIIII", usec, int(max_bytes * self.fps), 0, 0x10, n, 0, 1, max_bytes, self.w, self.h, 0, 0, 0, 0)
        rate, scale = (int(round(self.fps * 1000)), 1000)
        strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, scale, rate, 0, n, max_bytes, 0xFFFFFFFF, 0, 0, 0, self.w, self.h)
        strf = struct.pack("<IiiHHIIiiII", 40, self.w, self.h, 1, 24, 0x47504A4D, self.w * self.h * 3, 0, 0, 0, 0)   # BITMAPINFOHEADER, 'MJPG'
        strl = b"LIST" + struct.pack("<I", 4 + 8 + len(strh) + 8 + len(strf)) + b"strl" + b"strh" + struct.pack("<I", len(strh)) + strh + \
               b"strf" + struct.pack("<I", len(strf)) + strf
        hdrl = b"LIST" + struct.pack("<I", 4 + 8 + len(avih) + len(strl)) + b"hdrl" + b"avih" + struct.pack("<I", len(avih)) + avih + strl
        head = b"RIFF" + struct.pack("<I", end - 8) + b"AVI " + hdrl
        assert len(head) == self._header_size(), (len(head), self._header_size())
        self.f.seek(0)
        self.f.write(head)
        self.f.seek(self.movi_start + 4)
        self.f.write(struct.pack("<I", movi_end - self.movi_start - 8))
        self.f.close()


def default_io_threads():
    """Worker threads for image decode / encode (Pillow releases the GIL inside its codecs): half the host cores, 4..48."""
    return max(4, min(48, (os.cpu_count() or 8) // 2))


def _host_buffer(pkg_empty, shape, dtype):
    """Page-locked when the HIP library can allocate it (direct DMA by the host entries), else a plain array."""
    if pkg_empty is not None:
        try:
            return pkg_empty(shape, dtype)
        except Exception:
            pass
    return np.empty(shape, dtype)


def _encode_jpeg(img, quality):
    buf = io.BytesIO()
    _pil().fromarray(np.ascontiguousarray(to_uint8(img)[:, :, ::-1]), "RGB").save(buf, "JPEG", quality=quality)
    return buf.getvalue()


def mux_video_from_files(paths, video_path, fps, pool, quality=95):
    """generate_real_video.py:175-186 as written there: re-read the result frames (sorted) and write the MJPG AVI.  Used when
    several ranks wrote the frames; decode + JPEG encode on the worker threads, the muxer appends in order."""
    writer = None
    for data, shape in pool.map(lambda p: (lambda im: (_encode_jpeg(im, quality), im.shape))(read_image_bgr(p)), paths):
        if writer is None:
            writer = MJPGWriter(video_path, fps, shape[1], shape[0], quality)
        writer.write_jpeg(data, shape)
    if writer is not None:
        writer.release()


def stylize_files(model, style_path, frame_paths, out_dir, video_path=None, fps=24, chunk=32, log=print,
                  io_threads=None, rank=0, world=1, broadcast=None, barrier=None, stats=None):
    """The reference script's main flow for one style and one frame list; returns the frame paths THIS rank wrote.

    Throughput form: the three stages of the reference's loop — cv2.imread, framework.transfer,
    cv2.imwrite (generate_real_video.py:152-171) — overlap: worker threads decode the frames of chunk k+1, k+2 straight into
    page-locked input buffers and encode / write chunk k-1, k-2 out of page-locked output buffers while the calling
    thread runs chunk k through `transfer_frames` (pad + crop on the device; ctypes releases the GIL for the call).
    rank / world: this rank stylizes its contiguous shard of the frames (video.shard_range); rank 0 runs the preparation
    and `broadcast(blob, 0)` ships the 70 KB state (dist.broadcast_state); `barrier()` orders the optional video muxing
    on rank 0 behind every rank's frame files.  `stats` (a dict) receives the stage timings."""
    import importlib
    import time
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    video = importlib.import_module("rerevst-code_amd.video")
    os.makedirs(out_dir, exist_ok=True)
    n = len(frame_paths)
    nthreads = int(io_threads or default_io_threads())
    use_global = getattr(model, "use_Global", True)
    t_start = time.perf_counter()
    with ThreadPoolExecutor(nthreads) as pool:
        # ---- once per video (generate_real_video.py:95-146); the sampled frames are decoded in parallel
        if rank == 0 or not use_global:
            model.prepare_style(read_image_bgr(style_path))
        if use_global:
            blob = None
            if rank == 0:
                ids = video.sample_indices(n)
                log("Preparations for Sequence-Level Global Feature Sharing (%d sampled frames)" % len(ids))
                model.clean()
                for f in pool.map(read_image_bgr, [frame_paths[i] for i in ids]):
                    model.add(f)
                model.compute()
                if world > 1:
                    blob = model.get_state()
            if world > 1:
                if broadcast is None:
                    raise ValueError("world > 1 needs a broadcast(blob, src) function (dist.broadcast_state)")
                blob = broadcast(blob, 0)
                if rank != 0:
                    model.set_state(blob)
        t_prep = time.perf_counter()

        lo, hi = video.shard_range(n, rank, world)
        written = []
        jpeg_q = deque()          # (chunk, [future -> (jpeg bytes, shape)]) in frame order, for the AVI of a single-rank run
        writer = None
        inline_video = video_path is not None and world == 1

        def drain_video(all_of_them):
            nonlocal writer
            while jpeg_q and (all_of_them or all(f.done() for f in jpeg_q[0])):
                for f in jpeg_q.popleft():
                    data, shape = f.result()
                    if writer is None:
                        writer = MJPGWriter(video_path, fps, shape[1], shape[0])
                    writer.write_jpeg(data, shape)

        def save(path, img):                     # worker: float -> uint8 (cv2.imwrite's saturation), PNG / JPEG by extension
            u8 = to_uint8(img)
            write_image_bgr(path, u8)
            return (_encode_jpeg(u8, 95), u8.shape) if inline_video else None

        on_device = getattr(model, "transfer_frames", None)      # absent on a model with the reference's surface only
        fast = on_device is not None and use_global and hi > lo
        gpu_s = 0.0
        if fast:
            # The reference reshapes every frame on its own (generate_real_video.py:152-171) and so accepts a list of mixed
            # sizes; the batched entry needs equal sizes per call.  The headers are read up front (no decode) and the chunks
            # are cut from RUNS of equally sized frames: a uniform video is one run, a mixed list degrades to shorter chunks.
            sizes = list(pool.map(image_size, frame_paths[lo:hi]))          # (H, W) per frame of this shard
            pkg_empty = getattr(importlib.import_module("rerevst-code_amd"), "pinned_empty", None)
            nbuf = 3
            chunk = max(1, min(int(chunk), hi - lo))
            chunks = []
            for i in range(lo, hi):
                if chunks and len(chunks[-1]) < chunk and sizes[chunks[-1][0] - lo] == sizes[i - lo]:
                    chunks[-1].append(i)
                else:
                    chunks.append([i])
            biggest = max(h * w for h, w in sizes)
            in_flat = [_host_buffer(pkg_empty, (chunk * biggest * 3,), np.uint8) for _ in range(nbuf)]
            out_flat = [_host_buffer(pkg_empty, (chunk * biggest * 3,), np.float32) for _ in range(nbuf)]

            def views(k):                        # chunk k's input / output slots: [frames][H][W][3] views of buffer set k % nbuf
                (H, W), m = sizes[chunks[k][0] - lo], len(chunks[k])
                return in_flat[k % nbuf][:m * H * W * 3].reshape(m, H, W, 3), out_flat[k % nbuf][:m * H * W * 3].reshape(m, H, W, 3)

            def load(path, dst):                 # worker: decode straight into the page-locked input slot
                img = read_image_bgr(path)
                if img.shape != dst.shape:
                    raise ValueError("frame %s decodes to %r, its header said %r" % (path, img.shape, dst.shape))
                dst[...] = img

            dec = {}                             # chunk -> decode futures
            enc = [[] for _ in range(nbuf)]      # output buffer -> encode futures still reading it

            def submit_decode(k):
                if k < len(chunks) and k not in dec:
                    dst = views(k)[0]
                    dec[k] = [pool.submit(load, frame_paths[i], dst[j]) for j, i in enumerate(chunks[k])]
            submit_decode(0); submit_decode(1)
            for k, idx in enumerate(chunks):
                for f in dec.pop(k):
                    f.result()
                for f in enc[k % nbuf]:          # the encoders of chunk k-3 have left this output buffer
                    f.result()
                t0 = time.perf_counter()
                src, dst = views(k)
                styled = on_device(src, out=dst)
                gpu_s += time.perf_counter() - t0
                submit_decode(k + 2)             # its input buffer was chunk k-1's: consumed
                futs = []
                for j, i in enumerate(idx):
                    out_path = os.path.join(out_dir, os.path.basename(frame_paths[i]))
                    futs.append(pool.submit(save, out_path, styled[j]))
                    written.append(out_path)
                enc[k % nbuf] = futs
                if inline_video:
                    jpeg_q.append(futs)
                    drain_video(False)
                log("stylized frames %d..%d of %d" % (idx[0], idx[-1], n))
            for fl in enc:
                for f in fl:
                    f.result()
        else:
            # frame mode, or a model with only the reference's transfer(): per frame, decode and encode still on the worker threads
            tool = video.ReshapeTool()
            pending = deque()
            look = deque(pool.submit(read_image_bgr, frame_paths[i]) for i in range(lo, min(hi, lo + 2 * nthreads)))
            nxt = lo + len(look)
            for i in range(lo, hi):
                f = look.popleft().result()
                if nxt < hi:
                    look.append(pool.submit(read_image_bgr, frame_paths[nxt])); nxt += 1
                t0 = time.perf_counter()
                styled = model.transfer(tool.process(f))[64:64 + f.shape[0], 64:64 + f.shape[1], :]
                gpu_s += time.perf_counter() - t0
                out_path = os.path.join(out_dir, os.path.basename(frame_paths[i]))
                fut = pool.submit(save, out_path, np.array(styled))
                written.append(out_path)
                pending.append(fut)
                if inline_video:
                    jpeg_q.append([fut])
                    drain_video(False)
                while len(pending) > 4 * nthreads:
                    pending.popleft().result()
            for f in pending:
                f.result()
        if inline_video:
            drain_video(True)
            if writer is not None:
                writer.release()
        t_frames = time.perf_counter()
        if barrier is not None:
            barrier()
        if video_path is not None and world > 1 and rank == 0:
            mux_video_from_files([os.path.join(out_dir, os.path.basename(p)) for p in sorted(frame_paths)], video_path, fps, pool)
    if stats is not None:
        stats.update(prep_s=t_prep - t_start, frames=hi - lo, frames_s=t_frames - t_prep, gpu_call_s=gpu_s, io_threads=nthreads,
                     frames_per_s=(hi - lo) / max(t_frames - t_prep, 1e-9))
    return written


def stylize_files_multistyle(model, style_paths, frame_paths, out_dir, video_path=None, fps=24, chunk=16, style_size=(384, 384), log=print,
                             io_threads=None):
    """"Multi-style Interpolation/test.py" on files (VideoStylization :40-111 + the loop :114-131): styles resized to
    384x384, every frame padded and encoded once (features cached by the model), every 16th + the last feature sampled,
    then frame i with the weight ramp; output frames are numbered %d.png as the reference writes them (:131).
    Decoding (ahead of the encoder pass) and file writing (behind the decoder pass) run on worker threads."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    video = __import__("importlib").import_module("rerevst-code_amd.video")
    os.makedirs(out_dir, exist_ok=True)
    n, S = len(frame_paths), len(style_paths)
    nthreads = int(io_threads or default_io_threads())
    with ThreadPoolExecutor(nthreads) as pool:
        model.prepare_style([video.resize_bilinear(im, style_size) for im in pool.map(read_image_bgr, style_paths)])
        tool = video.ReshapeTool()
        shapes, feats = [], []
        batch_encode = getattr(model, "generate_content_features_batch", None)     # absent on a model with the reference's surface only
        look = deque(pool.submit(read_image_bgr, p) for p in frame_paths[:2 * nthreads])
        group = []                               # padded frames of one size waiting for one batched encoder call

        def flush():
            if group:
                feats.extend(batch_encode(group) if batch_encode else [model.generate_content_features(g) for g in group])      # (a group of one too: every frame of a video through ONE entry, one encoder arithmetic)
                group.clear()
        for i in range(n):
            f = look.popleft().result()
            if i + 2 * nthreads < n:
                look.append(pool.submit(read_image_bgr, frame_paths[i + 2 * nthreads]))
            shapes.append(f.shape)
            padded = tool.process(f)
            if group and (padded.shape != group[0].shape or len(group) >= chunk):
                flush()
            group.append(padded)
        flush()
        log("Encoded %d frames; statistics from %d of them" % (n, len(video.sample_indices_multistyle(n))))
        model.clean()
        for i in video.sample_indices_multistyle(n):
            model.add_patch(feats[i])
        model.compute_norm()
        many = getattr(model, "transfer_many", None)          # absent on a model with the reference's surface only
        written, writer, pending = [], None, deque()

        def save(path, img):
            u8 = to_uint8(img)
            write_image_bgr(path, u8)
            return (_encode_jpeg(u8, 95), u8.shape) if video_path else None

        def drain(keep):
            nonlocal writer
            while len(pending) > keep:
                res = pending.popleft().result()
                if res is not None:
                    if writer is None:
                        writer = MJPGWriter(video_path, fps, res[1][1], res[1][0])
                    writer.write_jpeg(*res)
        for c0 in range(0, n, chunk):
            idx = list(range(c0, min(n, c0 + chunk)))
            wts = [video.ramp_weights(i, n, S) for i in idx]
            styled = many([feats[i] for i in idx], wts) if many is not None else [model.transfer(feats[i], w) for i, w in zip(idx, wts)]
            for j, i in enumerate(idx):
                H, W, _ = shapes[i]
                out_path = os.path.join(out_dir, "%d.png" % i)
                pending.append(pool.submit(save, out_path, np.array(styled[j][64:64 + H, 64:64 + W, :])))      # own copy: `styled` may be a recycled block
                written.append(out_path)
            drain(4 * nthreads)
            log("stylized frames %d..%d of %d" % (idx[0], idx[-1], n))
        drain(0)
        if writer is not None:
            writer.release()
    return written


def _self_launch(n_ranks, argv):
    """`--gpus N` without a launcher: start the N ranks (one process per GPU), wait; a rank that dies takes the others down."""
    import socket
    import subprocess
    import time
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, "-m", "rerevst-code_amd.driver"] + list(argv), env=env,
                                      cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    rc, live = 0, list(procs)
    while live and not rc:
        time.sleep(0.1)
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                rc = rc or r
    for p in live:
        p.kill()
    for p in live:
        p.wait()
    return rc


def main(argv=None, model_factory=None):
    """`model_factory(args, device)` builds the model (default: the HIP Stylization / MultiStyleStylization); the CPU tests
    of the multi-rank control flow pass their own."""
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--style", required=True, nargs="+", help="one style image, or several for multi-style interpolation")
    ap.add_argument("--frames", required=True, help="glob pattern of the content frames")
    ap.add_argument("--checkpoint", required=True, help="style_net-TIP-final.pth (or 'synthetic' for seeded weights)")
    ap.add_argument("--out", required=True, help="directory for the stylized frames")
    ap.add_argument("--video", default=None, help="also write a Motion-JPEG .avi here")
    ap.add_argument("--fps", type=float, default=24)
    ap.add_argument("--no-global", action="store_true", help="per-frame statistics (use_Global=False)")
    ap.add_argument("--device", type=int, default=None, help="HIP device (default: LOCAL_RANK, else 0)")
    ap.add_argument("--gpus", type=int, default=1, help="shard the frames over N GPUs of this node: one process per GPU, one broadcast of the "
                    "70 KB state per style (RCCL), every rank writes its own frames")
    ap.add_argument("--io-threads", type=int, default=0, help="decode / encode worker threads per rank (default: half the host cores, 4..48)")
    ap.add_argument("--chunk", type=int, default=32, help="frames per transfer_frames call")
    args = ap.parse_args(argv)
    for sp in args.style:
        if not os.path.exists(sp):
            sys.exit("Style image %s not exists" % sp)                # generate_real_video.py:93-94, test.py:51-52
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if len(args.style) > 1:
            sys.exit("--gpus N shards the single-style flow; the multi-style flow runs on one GPU")
        sys.exit(_self_launch(args.gpus, argv))
    import importlib
    import time
    pkg = importlib.import_module("rerevst-code_amd")
    D = importlib.import_module("rerevst-code_amd.dist")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    broadcast = barrier = None
    rank = local = 0
    if world > 1:
        if world != args.gpus:
            sys.exit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        # RRV_DRIVER_BACKEND=gloo: all ranks may share a GPU (control-flow check on a one-GPU box); default nccl = RCCL
        rank, world, local = D.init_from_env(os.environ.get("RRV_DRIVER_BACKEND"))
        import torch.distributed as dist
        broadcast, barrier = D.broadcast_state, dist.barrier
    if args.gpus > 1 and args.device is not None:
        sys.exit("--device picks the GPU of a single-process run; with --gpus N every rank takes the GPU of its LOCAL_RANK")
    device = args.device if args.device is not None else local
    if model_factory is None:
        if world > 1 and os.environ.get("RRV_DRIVER_BACKEND", "nccl") == "nccl":
            import torch                      # (a single-process run needs no torch at all)
            if world > torch.cuda.device_count():
                sys.exit("--gpus %d but this node has %d GPUs (one rank per GPU over RCCL)" % (world, torch.cuda.device_count()))
        elif world > 1:                       # gloo control-flow runs: the ranks may share the GPUs there are
            import torch
            device = device % max(1, torch.cuda.device_count())
        ckpt = pkg.synthetic_weights(0) if args.checkpoint == "synthetic" else args.checkpoint
        if len(args.style) > 1:     # "Multi-style Interpolation/test.py"
            model = pkg.MultiStyleStylization(ckpt, cuda=True, style_num=len(args.style), device=device)
        else:
            model = pkg.Stylization(ckpt, cuda=True, use_Global=not args.no_global, device=device)
    else:
        model = model_factory(args, device)
    log = print if rank == 0 else (lambda *a, **k: None)
    stats = {}
    t0 = time.perf_counter()
    if len(args.style) > 1:
        stylize_files_multistyle(model, args.style, list_frames(args.frames), args.out, args.video, args.fps, io_threads=args.io_threads or None, log=log)
    else:
        stylize_files(model, args.style[0], list_frames(args.frames), args.out, args.video, args.fps, chunk=args.chunk, log=log,
                      io_threads=args.io_threads or None, rank=rank, world=world, broadcast=broadcast, barrier=barrier, stats=stats)
        if stats:
            print("[rank %d] %d frames in %.2f s = %.1f frames/s file -> file (%d I/O threads; GPU calls %.2f s; preparation %.2f s)"
                  % (rank, stats["frames"], stats["frames_s"], stats["frames_per_s"], stats["io_threads"], stats["gpu_call_s"], stats["prep_s"]), flush=True)
    if hasattr(model, "close"):
        model.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
