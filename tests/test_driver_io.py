"""Host-side driver (rerevst-code_amd/driver.py): image I/O conventions of cv2.imread / cv2.imwrite, the sorted frame
list, the Motion-JPEG AVI container, and the reference script's flow (test/generate_real_video.py:86-186) end to end
with the CPU oracle standing in for the GPU model."""
import importlib
import io
import os
import struct

import numpy as np
import pytest

D = importlib.import_module("rerevst-code_amd.driver")
Image = pytest.importorskip("PIL.Image")


def test_png_roundtrip_is_bgr_and_exact(tmp_path):
    img = np.random.default_rng(0).integers(0, 256, (13, 17, 3), dtype=np.uint8)     # BGR
    p = str(tmp_path / "a.png")
    D.write_image_bgr(p, img)
    np.testing.assert_array_equal(D.read_image_bgr(p), img)
    with Image.open(p) as im:                                                         # the file itself holds RGB
        np.testing.assert_array_equal(np.asarray(im), img[:, :, ::-1])


def test_float_images_saturate_and_round_like_imwrite(tmp_path):
    f = np.array([[[-3.0, 0.49, 0.5], [254.5, 255.4, 300.0]]], dtype=np.float32)
    np.testing.assert_array_equal(D.to_uint8(f), np.array([[[0, 0, 0], [254, 255, 255]]], dtype=np.uint8))   # rint: ties to even
    p = str(tmp_path / "f.png")
    D.write_image_bgr(p, np.full((4, 4, 3), 127.6, np.float32))
    assert (D.read_image_bgr(p) == 128).all()


def test_frame_list_is_sorted_and_missing_pattern_fails(tmp_path):
    for n in ("frame_0010.png", "frame_0002.png", "frame_0001.png"):
        D.write_image_bgr(str(tmp_path / n), np.zeros((4, 4, 3), np.uint8))
    assert [os.path.basename(p) for p in D.list_frames(str(tmp_path / "*.png"))] == ["frame_0001.png", "frame_0002.png", "frame_0010.png"]
    with pytest.raises(FileNotFoundError):
        D.list_frames(str(tmp_path / "*.jpg"))


def test_mjpg_avi_container(tmp_path):
    p = str(tmp_path / "v.avi")
    w = D.MJPGWriter(p, 24, 32, 20)
    frames = [np.full((20, 32, 3), v, np.uint8) for v in (10, 120, 240)]
    for f in frames:
        w.write(f)
    with pytest.raises(ValueError):
        w.write(np.zeros((8, 8, 3), np.uint8))
    w.release()
    b = open(p, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"AVI " and struct.unpack("<I", b[4:8])[0] == len(b) - 8
    assert b[12:16] == b"LIST" and b[20:24] == b"hdrl" and b[24:28] == b"avih"
    usec, _, _, flags, nframes, _, nstreams, _, width, height = struct.unpack("<10I", b[32:72])
    assert (usec, flags & 0x10, nframes, nstreams, width, height) == (41667, 0x10, 3, 1, 32, 20)
    assert b[108:112] == b"vids" and b[112:116] == b"MJPG"
    movi = b.index(b"movi")
    assert b[movi + 4:movi + 8] == b"00dc"
    size = struct.unpack("<I", b[movi + 8:movi + 12])[0]
    with Image.open(io.BytesIO(b[movi + 12:movi + 12 + size])) as im:                 # the first chunk is a JPEG of frame 0
        a = np.asarray(im.convert("RGB"))
    assert a.shape == (20, 32, 3) and abs(int(a.mean()) - 10) <= 2
    idx = b.index(b"idx1")
    assert struct.unpack("<I", b[idx + 4:idx + 8])[0] == 3 * 16
    ck, fl, off, sz = struct.unpack("<4sIII", b[idx + 8:idx + 24])
    assert (ck, fl, off, sz) == (b"00dc", 0x10, 4, size)


def test_reference_script_flow_with_the_oracle(tmp_path, pkg, oracle):
    """prepare_style -> clean -> add(sampled) -> compute -> transfer(pad) -> crop -> write, on files."""
    src, out = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    frames = [pkg.synth_frame(i, 24, 32, kind="smooth") for i in range(3)]
    for i, f in enumerate(frames):
        D.write_image_bgr(str(src / ("f%02d.png" % i)), f)
    D.write_image_bgr(str(tmp_path / "style.png"), pkg.synth_style(32, 32, kind="smooth"))
    model = oracle.Stylization(pkg.synthetic_weights(0))
    written = D.stylize_files(model, str(tmp_path / "style.png"), D.list_frames(str(src / "*.png")), str(out),
                              video_path=str(tmp_path / "v.avi"), fps=12, log=lambda *_: None)
    assert [os.path.basename(p) for p in written] == ["f00.png", "f01.png", "f02.png"]
    V = importlib.import_module("rerevst-code_amd.video")
    ref_model = oracle.Stylization(pkg.synthetic_weights(0))
    ref = V.stylize_video(ref_model, frames, pkg.synth_style(32, 32, kind="smooth"))
    for i, p in enumerate(written):
        got = D.read_image_bgr(p)
        assert got.shape == (24, 32, 3)
        np.testing.assert_array_equal(got, D.to_uint8(ref[i]))
    assert os.path.getsize(str(tmp_path / "v.avi")) > 212


def test_multistyle_script_flow_with_the_oracle(tmp_path, pkg, oracle):
    """"Multi-style Interpolation/test.py" on files: styles resized, frames padded and encoded once, sampled features,
    weight ramp, crop, frames written as %d.png — with the CPU oracle as the model."""
    src, out = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    frames = [pkg.synth_frame(i, 24, 32, kind="smooth") for i in range(3)]
    for i, f in enumerate(frames):
        D.write_image_bgr(str(src / ("f%02d.png" % i)), f)
    styles = [pkg.synth_style(20, 28, kind="smooth", seed=5), pkg.synth_style(28, 20, kind="smooth", seed=6)]
    for k, s in enumerate(styles):
        D.write_image_bgr(str(tmp_path / ("style%d.png" % k)), s)
    model = oracle.MultiStylization(pkg.synthetic_weights(0), 2)
    written = D.stylize_files_multistyle(model, [str(tmp_path / "style0.png"), str(tmp_path / "style1.png")], D.list_frames(str(src / "*.png")),
                                         str(out), video_path=str(tmp_path / "v.avi"), fps=12, style_size=(32, 32), log=lambda *_: None)
    assert [os.path.basename(p) for p in written] == ["0.png", "1.png", "2.png"]
    V = importlib.import_module("rerevst-code_amd.video")
    ref = V.stylize_video_multistyle(oracle.MultiStylization(pkg.synthetic_weights(0), 2), frames, styles, style_size=(32, 32))
    for i, p in enumerate(written):
        np.testing.assert_array_equal(D.read_image_bgr(p), D.to_uint8(ref[i]))
    assert os.path.getsize(str(tmp_path / "v.avi")) > 212


class _FramesModel:
    """The oracle behind the HIP model's batched driver entry (`transfer_frames(frames, out=)`: unpadded in, cropped
    out), so that the threaded chunk pipeline of stylize_files runs on CPU.  Counts the calls."""

    def __init__(self, oracle, weights):
        self.o = oracle.Stylization(weights)
        self.O = oracle
        self.use_Global = True
        self.calls = []
        for name in ("prepare_style", "clean", "add", "compute", "get_state", "set_state", "transfer"):
            setattr(self, name, getattr(self.o, name))

    def transfer_frames(self, frames, out=None):
        frames = np.asarray(frames)
        B, H, W, _ = frames.shape
        self.calls.append(B)
        if out is None:
            out = np.empty((B, H, W, 3), np.float32)
        PH, PW = self.O.padded_size(H), self.O.padded_size(W)
        for b in range(B):
            out[b] = self.o.transfer(self.O.reflect_pad(frames[b], PH, PW))[64:64 + H, 64:64 + W]
        return out


def _write_inputs(tmp_path, pkg, n, hw=(24, 32)):
    src = tmp_path / "in"
    src.mkdir()
    frames = [pkg.synth_frame(i, hw[0], hw[1], kind="smooth") for i in range(n)]
    for i, f in enumerate(frames):
        D.write_image_bgr(str(src / ("f%02d.png" % i)), f)
    style = pkg.synth_style(32, 32, kind="smooth")
    D.write_image_bgr(str(tmp_path / "style.png"), style)
    return src, frames, style


def test_threaded_chunk_pipeline_equals_serial_flow(tmp_path, pkg, oracle):
    """Decode workers -> page-locked-style input buffers -> transfer_frames per chunk -> encode workers, three rotating
    buffer sets, ragged last chunk, AVI appended in frame order: the files equal the serial reference flow."""
    src, frames, style = _write_inputs(tmp_path, pkg, 7)
    model = _FramesModel(oracle, pkg.synthetic_weights(0))
    stats = {}
    written = D.stylize_files(model, str(tmp_path / "style.png"), D.list_frames(str(src / "*.png")), str(tmp_path / "out"),
                              video_path=str(tmp_path / "v.avi"), fps=12, chunk=2, io_threads=3, log=lambda *_: None, stats=stats)
    assert model.calls == [2, 2, 2, 1] and stats["frames"] == 7 and stats["io_threads"] == 3
    V = importlib.import_module("rerevst-code_amd.video")
    ref = V.stylize_video(oracle.Stylization(pkg.synthetic_weights(0)), frames, style)
    assert [os.path.basename(p) for p in written] == ["f%02d.png" % i for i in range(7)]
    for i, p in enumerate(written):
        np.testing.assert_array_equal(D.read_image_bgr(p), D.to_uint8(ref[i]))
    b = open(str(tmp_path / "v.avi"), "rb").read()
    assert struct.unpack("<I", b[48:52])[0] == 7                                      # avih.dwTotalFrames
    # frames of other sizes in the list: as the reference (generate_real_video.py:152-171 reshapes every frame on its own)
    # they are stylized, in shorter chunks of equal size — f03 and f04 smaller, then back to the video's size
    for i in (3, 4):
        frames[i] = pkg.synth_frame(i, 16, 40, kind="smooth")
        D.write_image_bgr(str(src / ("f%02d.png" % i)), frames[i])
    model = _FramesModel(oracle, pkg.synthetic_weights(0))
    written = D.stylize_files(model, str(tmp_path / "style.png"), D.list_frames(str(src / "*.png")), str(tmp_path / "out2"), chunk=4, io_threads=2,
                              log=lambda *_: None)
    assert model.calls == [3, 2, 2]
    o = oracle.Stylization(pkg.synthetic_weights(0))
    o.prepare_style(style); o.clean()
    for i in V.sample_indices(7):
        o.add(frames[i])
    o.compute()
    tool = V.ReshapeTool()
    for i, p in enumerate(written):
        H, W = frames[i].shape[:2]
        tool = V.ReshapeTool()                      # the reference's tool fixes the padded size at its first frame; per size here
        ref_i = o.transfer(tool.process(frames[i]))[64:64 + H, 64:64 + W]
        np.testing.assert_array_equal(D.read_image_bgr(p), D.to_uint8(ref_i))


def _cli_rank(rank, world, port, argv):
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RRV_DRIVER_BACKEND="gloo")
    import torch
    torch.set_num_threads(2)
    import rerevst_oracle as O
    pkg = importlib.import_module("rerevst-code_amd")
    drv = importlib.import_module("rerevst-code_amd.driver")
    rc = drv.main(argv, model_factory=lambda args, device: _FramesModel(O, pkg.synthetic_weights(0)))
    assert rc == 0


def test_cli_two_ranks_gloo_equals_single_process(tmp_path, pkg, oracle):
    """`driver --gpus 2` control flow on CPU (gloo): rank 0 prepares, one state broadcast, contiguous shards, every rank
    writes its own frames, rank 0 muxes the AVI from the files behind a barrier — same files as the single-process run."""
    import torch.multiprocessing as mp
    src, frames, style = _write_inputs(tmp_path, pkg, 5)
    common = ["--style", str(tmp_path / "style.png"), "--frames", str(src / "*.png"), "--checkpoint", "synthetic", "--io-threads", "2", "--chunk", "2"]
    port = 29900 + os.getpid() % 90
    mp.spawn(_cli_rank, args=(2, port, common + ["--out", str(tmp_path / "o2"), "--gpus", "2", "--video", str(tmp_path / "v2.avi")]), nprocs=2, join=True)
    rc = D.main(common + ["--out", str(tmp_path / "o1"), "--video", str(tmp_path / "v1.avi")],
                model_factory=lambda args, device: _FramesModel(oracle, pkg.synthetic_weights(0)))
    assert rc == 0
    names = sorted(os.listdir(str(tmp_path / "o1")))
    assert names == sorted(os.listdir(str(tmp_path / "o2"))) == ["f%02d.png" % i for i in range(5)]
    for nm in names:
        np.testing.assert_array_equal(D.read_image_bgr(str(tmp_path / "o2" / nm)), D.read_image_bgr(str(tmp_path / "o1" / nm)))
    b1, b2 = open(str(tmp_path / "v1.avi"), "rb").read(), open(str(tmp_path / "v2.avi"), "rb").read()
    assert struct.unpack("<I", b2[48:52])[0] == 5 and len(b1) == len(b2)             # the JPEGs of identical frames
