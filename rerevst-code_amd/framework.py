"""Host-side mirror of the reference drop-in boundary ``Stylization``
(test/framework.py:56-118): same method names, argument meaning and return types, so a
``generate_real_video.py``-style driver runs unchanged.  All compute happens in
librerevst_hip.so on one MI355X; this class only marshals numpy buffers through the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from .weights import weight_table, load_checkpoint


class RRVError(RuntimeError):
    pass


def pinned_empty(shape, dtype=np.float32):
    """A numpy array in page-locked host memory (rrv_host_alloc): the host-buffer entries DMA straight from / into
    it instead of staging through the library's own pinned buffers.  Freed when the last view of it dies."""
    import weakref
    lib = _lib.load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = C.c_void_p()
    if lib.rrv_host_alloc(max(n, 1), C.byref(ptr)) != 0:
        raise MemoryError("rrv_host_alloc(%d) failed" % n)
    buf = (C.c_char * max(n, 1)).from_address(ptr.value)
    weakref.finalize(buf, lib.rrv_host_free, ptr)
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def _u8_image(img, what):
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("%s must be uint8 HWC BGR with 3 channels, got %s %s" % (what, a.dtype, a.shape))
    return a


class _OutputPool:
    """Output arrays of the host entries come out of recycled page-locked blocks: the reference's call surface returns a
    fresh array per frame (framework.py:40-49), and a fresh pageable 5 MB array costs ~0.2 ms of page faults plus a
    staging copy per call — a tenth of a one-frame transfer().  A block goes back to the pool when the LAST array that
    views it dies (numpy collapses the base of every derived view onto the lease object below), so a recycled block is
    never visible to the caller.  Bounded: beyond `cap_bytes` of blocks handed out or parked, arrays are plain numpy."""
    cap_bytes = 2 << 30
    keep_per_size = 8

    def __init__(self):
        import threading
        self.free = {}            # nbytes -> [address]
        self.live = 0             # bytes of blocks in existence (handed out + parked)
        self.lock = threading.RLock()     # finalizers run on whichever thread drops the last reference (re-entrant: a GC pass inside empty())

    def _give_back(self, lib, addr, nbytes):
        with self.lock:
            lst = self.free.setdefault(nbytes, [])
            if len(lst) < self.keep_per_size:
                lst.append(addr)
                return
            self.live -= nbytes
        lib.rrv_host_free(C.c_void_p(addr))

    def empty(self, shape, dtype=np.float32):
        import weakref
        dt = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = max(count * dt.itemsize, 1)
        lib = _lib.load()
        with self.lock:
            lst = self.free.get(nbytes)
            addr = lst.pop() if lst else None
            if addr is None:
                for size in sorted(self.free, reverse=True):          # over the cap: parked blocks of other sizes go first
                    while self.free[size] and self.live + nbytes > self.cap_bytes:
                        lib.rrv_host_free(C.c_void_p(self.free[size].pop()))
                        self.live -= size
                if self.live + nbytes > self.cap_bytes:
                    return np.empty(shape, dtype=dt)
                self.live += nbytes               # reserved before the allocation (released below if it fails)
        if addr is None:
            ptr = C.c_void_p()
            if lib.rrv_host_alloc(nbytes, C.byref(ptr)) != 0:
                with self.lock:
                    self.live -= nbytes
                return np.empty(shape, dtype=dt)
            addr = ptr.value
        lease = (C.c_char * nbytes).from_address(addr)
        weakref.finalize(lease, self._give_back, lib, addr, nbytes)
        return np.frombuffer(lease, dtype=dt, count=count).reshape(shape)


_outputs = _OutputPool()


class Stylization():
    """``Stylization(checkpoint, cuda=True, use_Global=True)`` (test/framework.py:57).

    `checkpoint` is a path to the reference ``.pth`` state_dict, or a ``{key: ndarray}``
    dict with the same keys (used with the seeded synthetic weights, since the released
    checkpoint is download-only).  `device` picks the HIP device ordinal (one process per
    GPU; defaults to LOCAL_RANK or 0).
    """

    def __init__(self, checkpoint, cuda=True, use_Global=True, device=None, style_num=1):
        if not cuda:
            raise RRVError("this implementation only runs on an MI355X GPU (cuda=False has no CPU fallback)")
        self.use_Global = bool(use_Global)   # False: per-frame statistics model of test/style_network_frame.py
        self._lib = _lib.load()
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = int(device)
        self.style_num = int(style_num)
        self._open = {}           # ticket id -> output array of an open transfer_async(): the GPU (or the retiring host copy)
                                  # writes into it until the ticket is collected or retired, whatever the caller keeps
        self._h = C.c_void_p()
        rc = self._lib.rrv_create(self.device, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise RRVError("rrv_create(device=%d) failed with %d (no HIP device?)" % (self.device, rc))
        weights = checkpoint if isinstance(checkpoint, dict) else load_checkpoint(checkpoint)
        for key, shape in weight_table().items():     # strict: every key, exact shape
            if key not in weights:
                raise KeyError("missing weight %s" % key)
            w = np.ascontiguousarray(weights[key], dtype=np.float32)
            if tuple(w.shape) != tuple(shape):
                raise ValueError("weight %s has shape %s, expected %s" % (key, w.shape, shape))
            shp = (C.c_int64 * w.ndim)(*w.shape)
            self._chk(self._lib.rrv_load_weight(self._h, key.encode(), w.ctypes.data_as(C.c_void_p), shp, w.ndim))
        self._chk(self._lib.rrv_finalize_weights(self._h))

    # ------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            msg = self._lib.rrv_last_error(self._h)
            err = RRVError("librerevst_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
            err.code = rc          # RRV_E_* of include/rerevst_hip.h (-5 = RRV_E_NOMEM)
            raise err

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rrv_destroy(self._h)      # waits for every stream: nothing writes the open tickets' outputs after it
            self._h = None
        self._open = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ===== Sequence-Level Global Feature Sharing (test/framework.py:82-95) =====
    def _global_only(self, what):
        if not self.use_Global:   # the reference's frame-mode TransformerNet has no add/compute/clean either
            raise RRVError("%s() belongs to Sequence-Level Global Feature Sharing (use_Global=True)" % what)

    def add(self, patch):
        self._global_only("add")
        a = _u8_image(patch, "patch")
        self._chk(self._lib.rrv_add(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1]))

    def compute(self):
        self._global_only("compute")
        self._chk(self._lib.rrv_compute(self._h))

    def set_debug(self, level):
        """Bounds-checked debug mode: 0 off, 1 verify guard bands / zero rings after every call, 2 after every kernel."""
        self._chk(self._lib.rrv_set_debug(self._h, int(level)))

    def debug_selftest(self):
        self._chk(self._lib.rrv_debug_selftest(self._h))

    def debug_fail_alloc(self, nth):
        """Failure injection: the nth next device allocation reports out-of-memory (0 disarms)."""
        self._chk(self._lib.rrv_debug_fail_alloc(self._h, int(nth)))

    def set_workspace_cap(self, nbytes):
        """compute() keeps all sampled frames' activations resident while they fit `nbytes` (default 64 GiB); beyond
        that it streams groups of frames one synchronisation point at a time (workspace independent of the frame count)."""
        self._chk(self._lib.rrv_set_workspace_cap(self._h, int(nbytes)))

    def last_compute_info(self):
        """(groups, frames per group, workspace bytes) of the last compute()."""
        g, n, b = C.c_int(), C.c_int(), C.c_size_t()
        self._chk(self._lib.rrv_last_compute_info(self._h, C.byref(g), C.byref(n), C.byref(b)))
        return g.value, n.value, b.value

    def clean(self):
        self._global_only("clean")
        self._chk(self._lib.rrv_clean(self._h))

    # ===== Style Transfer (test/framework.py:99-118) =====
    def prepare_style(self, style):
        """Single style image (test/framework.py:99) or a list of them
        ("Multi-style Interpolation/stylization.py":71)."""
        styles = style if isinstance(style, (list, tuple)) else [style]
        for sid, s in enumerate(styles):
            a = _u8_image(s, "style")
            self._chk(self._lib.rrv_prepare_style(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], sid))

    def transfer(self, frame, style_weight=None):
        """uint8 BGR HWC frame -> float32 BGR HWC in 0..255 (test/framework.py:106-118).
        With `style_weight` (list of floats) the saved state of the prepared styles is
        blended first (stylization.py:94-100)."""
        a = _u8_image(frame, "frame")
        H, W = a.shape[:2]
        out = _outputs.empty((H // 8 * 8, W // 8 * 8, 3))     # the max pools floor the size, as in the reference
        if not self.use_Global:
            self._chk(self._lib.rrv_transfer_frame_mode(self._h, a.ctypes.data_as(C.c_void_p), H, W, out.ctypes.data_as(C.c_void_p)))
            return out
        if style_weight is None:      # the reference's hot call: plain addresses (ctypes' data_as() objects cost ~10 us a call)
            rc = self._lib.rrv_transfer(self._h, a.__array_interface__["data"][0], H, W, out.__array_interface__["data"][0])
            if rc != 0:
                self._chk(rc)
            return out
        w = (C.c_float * len(style_weight))(*[float(v) for v in style_weight])
        self._chk(self._lib.rrv_transfer_blend(self._h, a.ctypes.data_as(C.c_void_p), H, W, w, len(style_weight),
                                               out.ctypes.data_as(C.c_void_p)))
        return out

    # ===== look-ahead form of transfer() for a one-frame-per-call loop =====
    def transfer_async(self, frame, out=None):
        """Queue one frame (H2D copy, kernels, D2H copy) and return a ticket at once; ``result(ticket)`` returns the
        stylized frame.  A driver loop written as

            prev = None
            for frame in frames:
                t = framework.transfer_async(frame)
                if prev is not None: write(framework.result(prev))
                prev = t
            write(framework.result(prev))

        overlaps frame i+1's copy-in and kernels with frame i's kernel tails, copy-out and file write.  Up to four
        tickets may be open (each on its own stream with a quarter of the CUs per launch): keeping three frames
        submitted ahead of the one being collected gives the best rate.  Same arithmetic as transfer(): bit-identical results."""
        if not self.use_Global:
            raise RRVError("transfer_async() needs the global-feature-sharing model (use_Global=True)")
        a = _u8_image(frame, "frame")
        H, W = a.shape[:2]
        oshape = (H // 8 * 8, W // 8 * 8, 3)
        if out is None:
            out = _outputs.empty(oshape)
        elif out.dtype != np.float32 or out.shape != oshape or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape %r" % (oshape,))
        t = C.c_long(-1)
        self._chk(self._lib.rrv_transfer_async(self._h, a.ctypes.data_as(C.c_void_p), H, W, out.ctypes.data_as(C.c_void_p), C.byref(t)))
        # The library owns `out` until the ticket is collected or retired: a caller that drops the ticket (an exception in
        # its loop, `prev` overwritten) must not hand the block back to the pool under a running kernel.  Submitting
        # ticket t retired ticket t - 4 (four staging sets), so only the last four stay referenced here.
        self._open[t.value] = out
        for old in [k for k in self._open if k <= t.value - 4]:
            del self._open[old]
        return (t.value, out)

    def result(self, ticket):
        tid, out = ticket
        self._chk(self._lib.rrv_transfer_wait(self._h, tid))
        self._open.pop(tid, None)
        return out

    # ===== device-resident entry (what bench.py times) =====
    def transfer_device(self, d_in_ptr, H, W, d_out_ptr):
        self._chk(self._lib.rrv_transfer_device(self._h, C.c_void_p(d_in_ptr), H, W, C.c_void_p(d_out_ptr)))

    def transfer_batch_device(self, d_in_ptr, B, H, W, d_out_ptr):
        """[B][H][W][3] uint8 in HBM -> [B][H][W][3] float32 in HBM, asynchronous on the library stream."""
        self._chk(self._lib.rrv_transfer_batch_device(self._h, C.c_void_p(d_in_ptr), B, H, W, C.c_void_p(d_out_ptr)))

    def transfer_batch(self, frames, out=None):
        """Stylize equally sized uint8 BGR frames (a list, or one [B][H][W][3] array) in one call; sub-batches are
        pipelined inside the library (copy in / kernels / copy out).  `out`: optional float32 [B][H][W][3] array to
        fill instead of allocating a fresh one."""
        if isinstance(frames, np.ndarray) and frames.ndim == 4 and frames.dtype == np.uint8 and frames.shape[3] == 3:
            a = np.ascontiguousarray(frames)
        else:
            a = np.stack([_u8_image(f, "frame") for f in frames])
        B, H, W, _ = a.shape
        oshape = (B, H // 8 * 8, W // 8 * 8, 3)
        if out is None:
            out = _outputs.empty(oshape)
        elif out.dtype != np.float32 or out.shape != oshape or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape %r" % (oshape,))
        self._chk(self._lib.rrv_transfer_batch(self._h, a.ctypes.data_as(C.c_void_p), B, H, W, out.ctypes.data_as(C.c_void_p)))
        return out

    def transfer_frames(self, frames, out=None):
        """UNPADDED uint8 BGR frames (a list, or one [B][H][W][3] array) -> [B][H][W][3] float32 stylized frames.
        The reference driver's ReshapeTool.process + crop (test/generate_real_video.py:61-83, :167) run on the
        device, without the padded copies on the host or over PCIe: the same picture as pad -> transfer -> crop (bit-identical
        for a fixed kernel choice, set_f43(0) / set_f43(2); the default picks kernels per launch geometry, the crop window included)."""
        if isinstance(frames, np.ndarray) and frames.ndim == 4 and frames.dtype == np.uint8 and frames.shape[3] == 3:
            a = np.ascontiguousarray(frames)
        else:
            a = np.stack([_u8_image(f, "frame") for f in frames])
        B, H, W, _ = a.shape
        if out is None:
            out = _outputs.empty((B, H, W, 3))
        elif out.dtype != np.float32 or out.shape != (B, H, W, 3) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape %r" % ((B, H, W, 3),))
        self._chk(self._lib.rrv_transfer_frames(self._h, a.ctypes.data_as(C.c_void_p), B, H, W, out.ctypes.data_as(C.c_void_p)))
        return out

    def transfer_frames_device(self, d_in_ptr, B, H, W, d_out_ptr):
        """Same on HBM buffers ([B][H][W][3] uint8 -> [B][H][W][3] float32), asynchronous on the library stream."""
        self._chk(self._lib.rrv_transfer_frames_device(self._h, C.c_void_p(d_in_ptr), B, H, W, C.c_void_p(d_out_ptr)))

    def sync(self):
        self._chk(self._lib.rrv_sync(self._h))

    def set_caller_stream(self, stream_ptr, enable=True):
        """Order the *_device entries against the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream):
        they wait for what is queued on it and it waits for their output — no host synchronisation needed."""
        self._chk(self._lib.rrv_set_caller_stream(self._h, C.c_void_p(stream_ptr), 1 if enable else 0))

    def set_f43(self, mode):
        """Kernel choice for the ten layers with a Winograd F(4x4,3x3) pack (rrv_set_f43): 0 never; 1 (default) per layer where
        the launch geometry — frames per launch, frame size, CUs the launch may use — lets it win, so a frame's low-order bits
        depend on how it was submitted (inside the parity bounds either way); 2 always.  With 0 or 2 every entry delivers
        the same bits for a frame."""
        self._chk(self._lib.rrv_set_f43(self._h, int(mode)))

    def set_grid_share(self, share):
        """Persistent grids use 1/share of the CUs (share 1..4): launches of several streams run side by side."""
        self._chk(self._lib.rrv_set_grid_share(self._h, int(share)))

    def debug_tensor(self, slot, index, H, W):
        """Activation tensor `index` of workspace slot `slot` for H x W frames (rrv_debug_copy_tensor), flat float32."""
        n = C.c_size_t(0)
        self._chk(self._lib.rrv_debug_copy_tensor(self._h, int(slot), int(index), int(H), int(W), None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        self._chk(self._lib.rrv_debug_copy_tensor(self._h, int(slot), int(index), int(H), int(W), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def set_host_io(self, mode):
        """0 (default): staged H2D / D2H copies; 1: zero copy — kernels read / write page-locked host memory directly."""
        self._chk(self._lib.rrv_set_host_io(self._h, int(mode)))

    def set_pipeline(self, n_slots):
        """1: every device-entry call runs on one stream; 2 (default): consecutive calls alternate over two
        (stream, workspace) pairs so two independent batches are in flight."""
        self._chk(self._lib.rrv_set_pipeline(self._h, int(n_slots)))

    # ===== shared state (RCCL broadcast payload / golden comparison) =====
    def get_state(self, style_id=0):
        out = np.empty(_lib.STATE_FLOATS, dtype=np.float32)
        self._chk(self._lib.rrv_get_state(self._h, out.ctypes.data_as(C.c_void_p), out.size, style_id))
        return out

    def set_state(self, blob, style_id=0):
        b = np.ascontiguousarray(blob, dtype=np.float32).reshape(-1)
        if b.size != _lib.STATE_FLOATS:
            raise ValueError("state blob must have %d floats" % _lib.STATE_FLOATS)
        self._chk(self._lib.rrv_set_state(self._h, b.ctypes.data_as(C.c_void_p), b.size, style_id))

    # ===== the path's one collective through the C ABI (RCCL; rerevst_hip.h: rrv_comm_*, rrv_broadcast_state) =====
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        if self._lib.rrv_comm_unique_id(buf) != 0:
            raise RRVError("rrv_comm_unique_id failed (librccl.so not found? set RRV_RCCL_PATH)")
        return bytes(buf.raw)

    def comm_init_rank(self, unique_id, nranks, rank):
        comm = C.c_void_p()
        self._chk(self._lib.rrv_comm_init_rank(self._h, C.create_string_buffer(bytes(unique_id), 128), int(nranks), int(rank), C.byref(comm)))
        return comm

    def comm_destroy(self, comm):
        if self._lib.rrv_comm_destroy(comm) != 0:
            raise RRVError("rrv_comm_destroy failed")

    def broadcast_state(self, comm, root, rank, style_id=0):
        """ncclBroadcast of the style's 17 536-float state from `root`; afterwards this rank holds it as after set_state."""
        self._chk(self._lib.rrv_broadcast_state(self._h, comm, int(root), int(rank), int(style_id)))

    def preclamp(self, H, W, image=0):
        """Pre-clamp network output of the last transfer (image `image` of its last launch), NHWC RGB normalised units."""
        out = _outputs.empty((H, W, 3))
        self._chk(self._lib.rrv_get_preclamp_image(self._h, out.ctypes.data_as(C.c_void_p), H, W, int(image)))
        return out

    # ===== per-launch timing (HIP events on the library's stream) =====
    def profile_begin(self):
        self._chk(self._lib.rrv_profile_begin(self._h))

    def profile_end(self):
        self._chk(self._lib.rrv_profile_end(self._h))
        n = self._lib.rrv_profile_count(self._h)
        rows = []
        name, ms, fl, by, fx = C.c_char_p(), C.c_float(), C.c_double(), C.c_double(), C.c_double()
        for i in range(n):
            self._chk(self._lib.rrv_profile_entry(self._h, i, C.byref(name), C.byref(ms), C.byref(fl), C.byref(by), C.byref(fx)))
            rows.append((name.value.decode(), ms.value, fl.value, by.value, fx.value))
        return rows


class ContentFeature():
    """Handle to an encoder output cached in HBM (what the reference stores as cache/%d.pt)."""

    def __init__(self, fid, shape):
        self.id, self.shape = fid, shape


class MultiStyleStylization(Stylization):
    """Mirror of "Multi-style Interpolation/stylization.py":42-100: ``Stylization(checkpoint, cuda, style_num)``
    with ``prepare_style(list)``, ``generate_content_features(img)``, ``add_patch(feature)``,
    ``compute_norm()``, ``clean()`` and ``transfer(feature, style_weight)``."""

    def __init__(self, checkpoint="", cuda=True, style_num=1, device=None):
        super().__init__(checkpoint, cuda=cuda, use_Global=True, device=device, style_num=style_num)

    def generate_content_features(self, content):
        a = _u8_image(content, "content")
        fid = C.c_int(-1)
        self._chk(self._lib.rrv_generate_content_features(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], C.byref(fid)))
        return ContentFeature(fid.value, a.shape)

    def generate_content_features_batch(self, frames):
        """`generate_content_features` for a run of equally sized frames (a list, or one [B][H][W][3] array) in one call:
        the reference's caching loop ("Multi-style Interpolation/test.py":87-101) pipelined inside the library
        (rrv_generate_content_features_batch).  Returns one ContentFeature per frame."""
        if isinstance(frames, np.ndarray) and frames.ndim == 4 and frames.dtype == np.uint8 and frames.shape[3] == 3:
            a = np.ascontiguousarray(frames)
        else:
            a = np.stack([_u8_image(f, "content") for f in frames])
        B, H, W, _ = a.shape
        ids = (C.c_int * B)()
        self._chk(self._lib.rrv_generate_content_features_batch(self._h, a.ctypes.data_as(C.c_void_p), B, H, W, ids))
        return [ContentFeature(int(i), (H, W, 3)) for i in ids]

    def add_patch(self, patch_feature):
        self._chk(self._lib.rrv_add_patch(self._h, patch_feature.id))

    def compute_norm(self):
        self.compute()

    def transfer(self, cur_feature, style_weight=[1.], out=None):
        H, W = cur_feature.shape[0] // 8 * 8, cur_feature.shape[1] // 8 * 8
        if out is None:
            out = _outputs.empty((H, W, 3))
        elif out.dtype != np.float32 or out.shape != (H, W, 3) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape %r" % ((H, W, 3),))
        w = (C.c_float * len(style_weight))(*[float(v) for v in style_weight])
        self._chk(self._lib.rrv_transfer_features(self._h, cur_feature.id, w, len(style_weight), out.ctypes.data_as(C.c_void_p)))
        return out

    def transfer_many(self, features, style_weights, out=None):
        """`transfer` for a run of cached features, one weight vector each, pipelined inside the library
        (rrv_transfer_features_batch).  Returns / fills a float32 [n][H][W][3] array."""
        n = len(features)
        H, W = features[0].shape[0] // 8 * 8, features[0].shape[1] // 8 * 8
        ns = len(style_weights[0])
        if out is None:
            out = _outputs.empty((n, H, W, 3))
        elif out.dtype != np.float32 or out.shape != (n, H, W, 3) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape %r" % ((n, H, W, 3),))
        ids = (C.c_int * n)(*[f.id for f in features])
        w = (C.c_float * (n * ns))(*[float(v) for row in style_weights for v in row])
        self._chk(self._lib.rrv_transfer_features_batch(self._h, ids, w, n, ns, out.ctypes.data_as(C.c_void_p)))
        return out

    def release_features(self):
        self._chk(self._lib.rrv_release_features(self._h))

    def set_multistyle_group(self, frames):
        """Frames per launch sequence of transfer_many (1..16; 0 = by the frame size, the default): per-image blended state inside one launch."""
        self._chk(self._lib.rrv_set_multistyle_group(self._h, int(frames)))

    def set_feature_cache_cap(self, nbytes):
        """Features are cached in HBM up to `nbytes` (default 64 GiB); beyond that a frame is kept as uint8 pixels and
        re-encoded at every use (the reference's cache is on disk: "Multi-style Interpolation/test.py":87-101)."""
        self._chk(self._lib.rrv_set_feature_cache_cap(self._h, int(nbytes)))

    def feature_cache_info(self):
        """(cached features, spilled features, bytes held by the cached ones)."""
        r, sp, b = C.c_int(), C.c_int(), C.c_size_t()
        self._chk(self._lib.rrv_feature_cache_info(self._h, C.byref(r), C.byref(sp), C.byref(b)))
        return r.value, sp.value, b.value
