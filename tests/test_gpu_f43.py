"""GPU tests (-m gpu) of the F(4x4,3x3) kernel choice (conv_f43_k; rrv_set_f43, include/rerevst_hip.h): by default the
library runs the encoder convs conv1_2 .. conv3_4 and the three ResidualBlock.conv2 in F(4x4,3x3) when a launch has
enough work items.  The suite runs in that default (the BASELINE configurations at full size: tests/test_gpu_default_choice.py);
this module forces the kernel onto every packed layer in every launch (mode 2) so that it meets the reference goldens, the
oracle, partial tiles, the crop windows and the debug mode also at the small sizes where the default rule would not pick it."""
import os

import numpy as np
import pytest

from conftest import load_golden, golden_inputs, decode_png, assert_pre_close, pre_worst, IMG_ATOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(pkg, weights):
    s = pkg.Stylization(weights, cuda=True)
    s.set_state(load_golden("global_a")["state"])
    yield s
    s.close()


def _batched(hip, padded, n=4):
    out = hip.transfer_batch([padded] * n)
    pre = hip.preclamp(*padded.shape[:2])
    for k in range(1, n):
        np.testing.assert_array_equal(out[k], out[0])         # a frame's arithmetic does not depend on its place in the launch
    return np.array(out[0]), pre


def test_f43_meets_the_reference_goldens(pkg, weights, oracle):
    """conv_f43_k on every layer that has an F(4x4,3x3) pack (mode 2 for the small fixtures, whose launches the default
    rule leaves to F(2x2,3x3); the default mode 1 for the reference's own 436 x 1024 frames at four per launch): reference
    goldens at the stated bounds, through the batched host entry the bench times.  Margins: tools/parity_margin.py
    (profiles/r04_parity_margin.txt)."""
    s = pkg.Stylization(weights, cuda=True)
    s.set_f43(2)
    g = load_golden("global_a")
    style, frames, ids, tid = golden_inputs(pkg, g)
    s.set_state(g["state"])
    out, pre = _batched(s, oracle.reflect_pad(frames[tid], 192, 192))
    assert_pre_close(pre, g["pre"])
    assert np.abs(out - g["out"]).max() <= IMG_ATOL
    s.set_f43(0)
    ref = s.transfer(oracle.reflect_pad(frames[tid], 192, 192))
    assert not np.array_equal(ref, out)                          # the other kernels really ran
    s.set_f43(1)
    np.testing.assert_array_equal(_batched(s, oracle.reflect_pad(frames[tid], 192, 192), n=1)[0], ref)     # default rule: too few work items in one 192 x 192 frame (from two frames on ResidualBlock.conv2 wins on its P8 input)
    np.testing.assert_array_equal(s.transfer(oracle.reflect_pad(frames[tid], 192, 192)), ref)
    g = load_golden("real_default")                               # the reference's default invocation: 436 x 1024 in 576 x 1152
    s.set_state(g["state"])
    frame = decode_png(g["frame%d_png" % int(g["transfer_id"])])
    padded = oracle.reflect_pad(frame, 576, 1152)
    out, pre = _batched(s, padded)                                # mode 1, four frames per launch
    s.set_f43(2)
    np.testing.assert_array_equal(s.transfer(padded), out)        # = conv_f43_k on the shipped layer set in every launch
    s.set_f43(1)
    out, pre = out[64:500, 64:1088], pre[64:500, 64:1088]
    assert_pre_close(pre[::4, ::4], g["pre_grid"])
    assert_pre_close(pre[186:250, 480:544], g["pre_patch"])
    assert np.abs(out[::4, ::4] - g["out_grid"]).max() <= IMG_ATOL
    assert np.abs(out[186:250, 480:544] - g["out_patch"]).max() <= IMG_ATOL
    # the on-device pad / crop entry (32-pixel-aligned windows for conv_f43_k) delivers the same pixels
    crop = s.transfer_frames([frame] * 4)
    for k in range(4):
        np.testing.assert_array_equal(crop[k], out)
    s.set_f43(2)
    g = load_golden("img1_256")                                   # BASELINE config 1
    s.set_state(g["state"])
    frame = decode_png(g["frame_png"])
    out, pre = _batched(s, oracle.reflect_pad(frame, 384, 384))
    assert_pre_close(pre[64:320, 64:320][::2, ::2], g["pre_grid"])
    assert np.abs(out[64:320, 64:320][::2, ::2] - g["out_grid"]).max() <= IMG_ATOL
    s.close()


def test_preparation_pass_never_uses_f43(pkg, weights, oracle):
    """prepare_style / add / compute run F(2x2,3x3) whatever the mode: the saved state is bit-identical in modes 0 and 2."""
    g = load_golden("global_a")
    style, frames, ids, tid = golden_inputs(pkg, g)
    states = []
    for mode in (0, 2):
        s = pkg.Stylization(weights, cuda=True)
        s.set_f43(mode)
        s.prepare_style(style); s.clean()
        for i in list(ids) * 3:                                  # nine sampled frames: the deferred encoder launches carry eight
            s.add(frames[i])
        s.compute()
        states.append(s.get_state())
        s.close()
    np.testing.assert_array_equal(states[0], states[1])


def test_fixed_choice_is_bit_identical_across_entries(pkg, weights, oracle, all_f43_layers):
    """Mode 2 (conv_f43_k in every launch, all ten packed layers): one frame per call == batched == tickets == pad/crop
    entry, bit for bit."""
    hip = pkg.Stylization(weights, cuda=True)
    hip.set_state(load_golden("global_a")["state"])
    hip.set_f43(2)
    try:
        PH, PW = oracle.padded_size(72), oracle.padded_size(100)    # ReshapeTool: 256 x 256 — what transfer_frames pads to on the device
        frames = [oracle.reflect_pad(pkg.synth_frame(300 + i, 72, 100, kind="noise"), PH, PW) for i in range(9)]
        one = [hip.transfer(f) for f in frames]
        batch = hip.transfer_batch(frames)
        for k in range(9):
            np.testing.assert_array_equal(batch[k], one[k])
        tickets = [hip.transfer_async(f) for f in frames[:4]]
        for k in (2, 0, 3, 1):
            np.testing.assert_array_equal(hip.result(tickets[k]), one[k])
        raw = [pkg.synth_frame(300 + i, 72, 100, kind="noise") for i in range(5)]
        crop = hip.transfer_frames(raw)
        for k in range(5):
            np.testing.assert_array_equal(crop[k], one[k][64:136, 64:164])
        raw2 = [pkg.synth_frame(320 + i, 200, 168, kind="smooth") for i in range(4)]       # padded 384 x 320: crop windows cut 32 x 32 items
        P2 = (oracle.padded_size(200), oracle.padded_size(168))
        crop2 = hip.transfer_frames(raw2)
        for k in range(4):
            np.testing.assert_array_equal(crop2[k], hip.transfer(oracle.reflect_pad(raw2[k], *P2))[64:264, 64:232])
        for rep in range(5):                                       # run-to-run determinism over both streams
            again = hip.transfer_batch(frames)
            np.testing.assert_array_equal(again, batch)
    finally:
        hip.close()


@pytest.fixture
def all_f43_layers():
    """Handles created inside run conv_f43_k on all ten packed layers (the encoder's ReLU and ReLU + pool epilogues too)."""
    old = os.environ.get("RRV_F43_LAYERS")
    os.environ["RRV_F43_LAYERS"] = "0x3ff"
    yield
    if old is None:
        del os.environ["RRV_F43_LAYERS"]
    else:
        os.environ["RRV_F43_LAYERS"] = old


@pytest.mark.parametrize("hw", [(200, 136), (77, 90), (40, 56), (33, 31), (8, 8), (264, 40)])
def test_f43_partial_tiles_vs_oracle(hw, pkg, oracle, weights, all_f43_layers):
    """Frame sizes that leave partial 32 x 32 work items, partial 4 x 4 tiles and odd pooling sizes at every level, every
    conv_f43_k epilogue (ReLU, ReLU + pool, LeakyReLU + norm + half-resolution residual + AdaIN): against the oracle.
    Regular bounds."""
    H, W = hw
    style = pkg.synth_style(48, 40, kind="smooth", seed=11)
    frames = [pkg.synth_frame(40 + i, H, W, kind="smooth") for i in range(3)]
    o = oracle.Stylization(weights)
    o.prepare_style(style); o.clean()
    for f in frames[:2]:
        o.add(f)
    o.compute()
    s = pkg.Stylization(weights, cuda=True)
    s.set_f43(2)
    s.set_state(o.get_state())
    got = s.transfer(frames[2])
    y = o.transfer(frames[2], return_preclamp=True)
    ref = oracle.tensor_to_image(y)
    assert got.shape == ref.shape == (H // 8 * 8, W // 8 * 8, 3)
    assert np.abs(got - ref).max() <= IMG_ATOL
    worst, _ = pre_worst(s.preclamp(*got.shape[:2]), y[0])
    assert worst <= 1.0
    b = s.transfer_batch([frames[2], frames[0], frames[2], frames[1], frames[2]])
    np.testing.assert_array_equal(b[0], got); np.testing.assert_array_equal(b[2], got); np.testing.assert_array_equal(b[4], got)
    s.close()


def test_f43_under_the_bounds_checked_debug_mode(pkg, weights, oracle, all_f43_layers):
    """Every conv_f43_k launch verified (guard bands, zero ring, slack rows): same bits as the unchecked run."""
    s = pkg.Stylization(weights, cuda=True)
    s.set_state(load_golden("global_a")["state"])
    s.set_f43(2)
    frames = [oracle.reflect_pad(pkg.synth_frame(700 + i, 50, 75, kind="noise"), 192, 208) for i in range(4)]
    raw = [pkg.synth_frame(700 + i, 50, 75, kind="noise") for i in range(4)]
    ref, ref_crop = np.array(s.transfer_batch(frames)), np.array(s.transfer_frames(raw))
    s.set_debug(2)
    np.testing.assert_array_equal(s.transfer_batch(frames), ref)
    np.testing.assert_array_equal(s.transfer_frames(raw), ref_crop)
    np.testing.assert_array_equal(s.transfer(frames[1]), ref[1])
    s.set_debug(0)
    s.close()


def test_channel_chunk_major_tensors_change_no_bit(pkg, weights, oracle, monkeypatch):
    """Round 6: between the seven packed encoder layers — and from ResidualBlock.conv1 to conv2 — the activations travel
    channel-chunk-major ([B][C/8][H+2][W+2][8]: a chunk's halo is contiguous rows instead of 2 312 scattered 32-byte pieces,
    conv_f43.h LAY) whenever the consumer runs conv_f43_k in a launch.  conv_f43_k stages the same bytes from either layout wherever every
    level is a multiple of 32 pixels wide, so there the results must be BIT-identical to the NHWC chain (RRV_P8=0): the headline's
    launch shape in the default mode.  A size with partial 32 x 32 items at every level (past the right edge of an image a tile reads
    discarded columns whose VALUES differ between the layouts: rounding noise in the edge tiles), the pad / crop entry and the
    bounds-checked debug mode: the same picture, and the oracle's."""
    video = __import__("importlib").import_module("rerevst-code_amd.video")
    g = load_golden("global_a")
    small = [oracle.reflect_pad(pkg.synth_frame(720 + i, 200, 264, kind="noise"), 392, 456) for i in range(5)]      # 392 x 456: 12.25 x 14.25 items at full resolution
    raw = [pkg.synth_frame(720 + i, 200, 264, kind="noise") for i in range(5)]
    big = np.stack([video.reflect_pad(pkg.synth_frame(1 + i, 512, 512, kind="noise"), 640, 640) for i in range(16)])
    res = {}
    for p8 in ("3", "1", "0"):      # encoder chain + ResidualBlock.conv2's input (the default) / the encoder chain only / NHWC everywhere
        monkeypatch.setenv("RRV_P8", p8)
        s = pkg.Stylization(weights, cuda=True)
        s.set_state(g["state"])
        s.set_f43(2)
        a = np.array(s.transfer_batch(small))
        b = np.array(s.transfer_frames(raw))
        s.set_debug(2)
        np.testing.assert_array_equal(s.transfer_batch(small), a)            # guard bands, zero rings (the twins are ring-layout tensors of 8-channel images) and slack rows intact after every kernel
        s.set_debug(0)
        s.set_f43(1)
        c = np.array(s.transfer_batch(big))
        res[p8] = (a, b, c)
        s.close()
    for other in ("3", "1"):
        np.testing.assert_array_equal(res[other][2], res["0"][2])      # 640 -> 320 -> 160: every level a multiple of 32 wide
        for x, y in zip(res[other][:2], res["0"][:2]):                  # partial items: the columns past the right edge hold other (discarded) values in a P8 plane
            assert np.abs(x - y).max() <= 0.2 * IMG_ATOL
    ref = oracle.Stylization(weights)
    ref.set_state(g["state"])
    assert np.abs(res["3"][0][2] - ref.transfer(small[2])).max() <= IMG_ATOL


def test_f43_mode_argument_is_checked(hip, pkg):
    with pytest.raises(pkg.RRVError):
        hip.set_f43(3)


def test_f43_large_frame_rows_beyond_the_2gib_mark(pkg, weights):
    """conv_f43_k on a 9216 x 1024 frame (64-channel full-resolution tensors of more than 2 GiB: 64-bit item origins, 32-bit
    tile-relative offsets, buffer limits clamped at 2 GiB): content that repeats every 64 rows gives bit-identical output
    rows 64 k apart, also beyond the 2 GiB mark (row 8176 on)."""
    H, W = 9216, 1024
    s = pkg.Stylization(weights, cuda=True)
    s.set_state(load_golden("global_a")["state"])
    s.set_f43(2)
    frame = np.tile(pkg.synth_frame(900, 64, W, kind="smooth"), (H // 64, 1, 1))
    out = s.transfer(frame)
    assert out.shape == (H, W, 3) and np.isfinite(out).all()
    top = out[512:576]
    assert float(top.std()) > 1.0
    for y0 in (4096, 8192, 8448, 8640):
        np.testing.assert_array_equal(out[y0:y0 + 64], top)
    s.set_f43(0)
    ref = s.transfer(frame)
    assert not np.array_equal(ref, out) and np.abs(ref - out).max() <= 2 * IMG_ATOL      # the other kernels, the same picture
    s.close()


def _choice(args, env_extra=None):
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    env.pop("RRV_F43", None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "f43_choice_check.py")] + [str(a) for a in args], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])["ms"]


@pytest.mark.parametrize("case", ["full chip, 16 frames", "tickets: a quarter of the CUs, one frame", "HSA_CU_MASK: half the CUs, 8 frames"])
def test_default_choice_is_the_faster_one_for_the_cus_a_launch_gets(case):
    """use_f43 counts rounds of the persistent workgroups a launch REALLY has (the device's CUs under HSA_CU_MASK / RRV_CUS,
    divided by the grid share of the look-ahead tickets): on the full chip, with a quarter of the CUs per launch and in a
    CU-masked process the default mode must be within 5 % of the faster of "F(2x2,3x3) everywhere" and "conv_f43_k
    everywhere" (it chooses per layer, so it is usually faster than both; measured numbers: profiles/r05_f43_choice.txt)."""
    for attempt in range(3):             # a timing comparison (medians of 12 calls per mode in a child process): a disturbed measurement gets two repeats
        if case.startswith("full"):
            ms = _choice([16, 640, 640])
        elif case.startswith("tickets"):
            ms = _choice([1, 640, 640, 4])
        else:
            full = _choice([8, 640, 640])
            ms = _choice([8, 640, 640], {"HSA_CU_MASK": "0:0-127"})
            if ms["0"] < 1.4 * full["0"]:
                pytest.skip("HSA_CU_MASK is not honoured on this box (masked %.2f ms, unmasked %.2f ms)" % (ms["0"], full["0"]))
        print("%s: F(2x2,3x3) everywhere %.3f ms, default rule %.3f ms, conv_f43_k everywhere %.3f ms" % (case, ms["0"], ms["1"], ms["2"]))
        if ms["1"] <= 1.05 * min(ms["0"], ms["2"]):
            return
    assert ms["1"] <= 1.05 * min(ms["0"], ms["2"]), (case, ms)


def test_ill_conditioned_state_keeps_f43_off_the_encoder(pkg, oracle):
    """use_f43's conditioning guard (rerevst_hip.hip: filter_conditioning): a state whose dynamic filters are far from the
    O(1) scale — the x4-decoder weight set, whose saved state the reference's own float32 run misses by 30x — runs the seven
    ENCODER layers in F(2x2,3x3) in the default mode (the three decoder layers keep F(4x4,3x3): profiles/r05_parity_margin.txt);
    a well-conditioned state on the same handle gets all ten layers back."""
    g4, g0 = load_golden("global_a_dec4"), load_golden("global_a")
    frames = np.stack([oracle.reflect_pad(pkg.synth_frame(900 + i, 128, 128, kind="noise"), 256, 256) for i in range(16)])
    w = pkg.weight_variant("dec4")
    a = pkg.Stylization(w, cuda=True)                         # default mode: the rule picks conv_f43_k on every packed layer at 16 x 256 x 256
    a.set_state(g4["state"])
    got_ill = np.array(a.transfer_batch(frames))
    a.set_state(g0["state"])
    got_ok = np.array(a.transfer_batch(frames))
    a.close()
    old = os.environ.get("RRV_F43_LAYERS")
    try:
        os.environ["RRV_F43_LAYERS"] = "0x380"                # decoder layers only, forced
        b = pkg.Stylization(w, cuda=True)
        b.set_f43(2); b.set_state(g4["state"])
        np.testing.assert_array_equal(got_ill, b.transfer_batch(frames))
        b.close()
        os.environ["RRV_F43_LAYERS"] = "0x3ff"                # all ten, forced
        c = pkg.Stylization(w, cuda=True)
        c.set_f43(2); c.set_state(g0["state"])
        np.testing.assert_array_equal(got_ok, c.transfer_batch(frames))
        c.close()
    finally:
        if old is None:
            os.environ.pop("RRV_F43_LAYERS", None)
        else:
            os.environ["RRV_F43_LAYERS"] = old
