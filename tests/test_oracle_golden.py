"""Pins the CPU oracle (oracle/rerevst_oracle.py) against the committed outputs of the
unmodified reference network (tests/golden/*.npz, made by tests/golden/make_goldens.py)."""
import numpy as np
import pytest

from conftest import (load_golden, golden_inputs, assert_state_close, assert_pre_close, IMG_ATOL)


def _run(oracle, pkg, weights, g):
    style, frames, ids, tid = golden_inputs(pkg, g)
    o = oracle.Stylization(weights)
    o.prepare_style(style)
    o.clean()
    for i in ids:
        o.add(frames[i])          # unpadded (quirk Q6)
    o.compute()
    H, W = frames[0].shape[:2]
    padded = oracle.reflect_pad(frames[tid], oracle.padded_size(H), oracle.padded_size(W))
    return o, padded


@pytest.mark.parametrize("case", ["global_a", "global_b"])
def test_state_blob_matches_reference(case, oracle, pkg, weights):
    g = load_golden(case)
    o, _ = _run(oracle, pkg, weights, g)
    assert_state_close(o.get_state(), g["state"])
    s = o.F_style["map"].sum(axis=(0, 1, 2))
    np.testing.assert_allclose(s, g["style_map_chansum"], rtol=1e-4, atol=1e-3)


def test_transfer_matches_reference_full(oracle, pkg, weights):
    g = load_golden("global_a")
    o, padded = _run(oracle, pkg, weights, g)
    assert_pre_close(o.transfer(padded, return_preclamp=True)[0], g["pre"])
    out = o.transfer(padded)
    assert out.dtype == np.float32 and out.shape == padded.shape
    assert np.abs(out - g["out"]).max() <= IMG_ATOL


def test_transfer_matches_reference_cropped_odd_sizes(oracle, pkg, weights):
    g = load_golden("global_b")
    o, padded = _run(oracle, pkg, weights, g)
    H, W = (int(v) for v in g["frame_hw"])
    pre = o.transfer(padded, return_preclamp=True)[0][64:64 + H, 64:64 + W]
    assert_pre_close(pre, g["pre_crop"])
    out = o.transfer(padded)[64:64 + H, 64:64 + W]
    assert np.abs(out - g["out_crop"]).max() <= IMG_ATOL


def test_set_state_roundtrip_and_uncomputed_error(oracle, pkg, weights):
    g = load_golden("global_a")
    o = oracle.Stylization(weights)
    with pytest.raises(RuntimeError):
        o.clean()
        o.F_style = {"relu4_1": (0, 1)}
        o.transfer(np.zeros((64, 64, 3), np.uint8))
    o2 = oracle.Stylization(weights)
    o2.set_state(g["state"])
    np.testing.assert_array_equal(o2.get_state(), g["state"])
    style, frames, ids, tid = golden_inputs(pkg, g)
    padded = oracle.reflect_pad(frames[tid], 192, 192)
    assert np.abs(o2.transfer(padded) - g["out"]).max() <= IMG_ATOL


def test_driver_helpers(oracle):
    assert [oracle.padded_size(n) for n in (256, 512, 1024, 436, 64)] == [384, 640, 1152, 576, 192]
    assert oracle.sample_indices(300) == [8 * s for s in range(37)] + [299]
    assert oracle.sample_indices(1) == [0]
    img = np.arange(5 * 4 * 3, dtype=np.uint8).reshape(5, 4, 3)
    p = oracle.reflect_pad(img, 192, 192)
    assert p.shape == (192, 192, 3)
    assert (p[63] == p[64]).all() and (p[:, 63] == p[:, 64]).all()   # edge-inclusive reflect


def test_multistyle_blend_matches_reference(oracle, pkg, weights):
    """S=2 "Multi-style Interpolation" path: per-style blobs + one blended transfer."""
    g = load_golden("multistyle_s2")
    styles = [pkg.synth_style(64, 64, kind="smooth", seed=7), pkg.synth_style(64, 64, kind="smooth", seed=8)]
    frames = [pkg.synth_frame(i, 64, 48, kind="smooth") for i in range(3)]
    padded = [oracle.reflect_pad(f, 192, 192) for f in frames]
    o = oracle.MultiStylization(weights, 2)
    o.prepare_style(styles)
    feats = [o.generate_content_features(p) for p in padded]
    o.clean()
    for i in (0, 2):
        o.add_patch(feats[i])
    o.compute_norm()
    assert_state_close(o.get_state(0), g["state0"], "style 0")
    assert_state_close(o.get_state(1), g["state1"], "style 1")
    wts = [float(v) for v in g["weights"]]
    pre = o.transfer(feats[1], wts, return_preclamp=True)[0][64:128, 64:112]
    assert_pre_close(pre, g["pre_crop"])
    assert np.abs(o.transfer(feats[1], wts)[64:128, 64:112] - g["out_crop"]).max() <= IMG_ATOL
    assert oracle.sample_indices_multistyle(33) == [0, 16, 32, 32]


def test_multistyle_s4_blend_matches_reference(oracle, pkg, weights):
    """BASELINE config 5 has FOUR styles: per-style blobs + one blended transfer with weights (.1,.2,.3,.4)."""
    g = load_golden("multistyle_s4")
    styles = [pkg.synth_style(64, 64, kind="smooth", seed=7 + k) for k in range(4)]
    frames = [pkg.synth_frame(i, 64, 48, kind="smooth") for i in range(3)]
    padded = [oracle.reflect_pad(f, 192, 192) for f in frames]
    o = oracle.MultiStylization(weights, 4)
    o.prepare_style(styles)
    feats = [o.generate_content_features(p) for p in padded]
    o.clean()
    for i in (0, 2):
        o.add_patch(feats[i])
    o.compute_norm()
    for k in range(4):
        assert_state_close(o.get_state(k), g["state%d" % k], "style %d" % k)
    wts = [float(v) for v in g["weights"]]
    assert_pre_close(o.transfer(feats[1], wts, return_preclamp=True)[0][64:128, 64:112], g["pre_crop"])
    assert np.abs(o.transfer(feats[1], wts)[64:128, 64:112] - g["out_crop"]).max() <= IMG_ATOL


def test_config2_geometry_matches_reference(oracle, pkg, weights):
    """BASELINE config 2 geometry (256x256 frame padded to 384x384, 512x512 style): the oracle's per-frame path with
    the REFERENCE's state (B = 13 sampled frames) against the reference's own output."""
    g = load_golden("config2_256")
    o = oracle.Stylization(weights)
    o.set_state(g["state"])
    padded = oracle.reflect_pad(pkg.synth_frame(int(g["transfer_id"]), 256, 256, kind="smooth"), 384, 384)
    pre = o.transfer(padded, return_preclamp=True)[0][64:320, 64:320]
    assert_pre_close(pre[::4, ::4], g["pre_grid"])
    np.testing.assert_allclose(pre.mean(axis=(0, 1)), g["pre_chanmean"], atol=1e-5)
    out = oracle.tensor_to_image(o.transfer(padded, return_preclamp=True))[64:320, 64:320]
    assert np.abs(out[::4, ::4] - g["out_grid"]).max() <= IMG_ATOL
    assert [int(i) for i in g["sample_ids"]] == oracle.sample_indices(100)


def test_frame_mode_matches_reference(oracle, pkg, weights):
    """use_Global=False model (test/style_network_frame.py): per-frame statistics, no saved state."""
    g = load_golden("frame_mode")
    o = oracle.Stylization(weights, use_Global=False)
    o.prepare_style(pkg.synth_style(64, 64, kind="smooth", seed=7))
    frame = oracle.reflect_pad(pkg.synth_frame(2, 64, 48, kind="smooth"), 192, 192)
    assert_pre_close(o.transfer(frame, return_preclamp=True)[0][64:128, 64:112], g["pre_crop"])
    assert np.abs(o.transfer(frame)[64:128, 64:112] - g["out_crop"]).max() <= IMG_ATOL


# ---- round 3: a second weight draw, degenerate channels, an ill-conditioned weight set, the reference's own inputs ----
from conftest import decode_png, assert_state_close_conditioned  # noqa: E402


@pytest.mark.parametrize("variant", ["seed1", "dead"])
def test_weight_variants_match_reference(variant, oracle, pkg):
    """global_a's flow with another weight draw (seed 1) and with dead / constant channels (exact zeros out of two VGG
    blocks, a constant channel out of Decoder.slice3.conv1: variance 0, rstd 1e4), against the unmodified reference."""
    g = load_golden("global_a_" + variant)
    w = pkg.weight_variant(variant)
    o, padded = _run(oracle, pkg, w, g)
    assert_state_close(o.get_state(), g["state"])
    assert_pre_close(o.transfer(padded, return_preclamp=True)[0], g["pre"])
    assert np.abs(o.transfer(padded) - g["out"]).max() <= IMG_ATOL


def test_ill_conditioned_weights_dec4(oracle, pkg):
    """Every Decoder weight x 4 (dynamic-filter entries up to 24): the per-frame path with the reference's state stays
    inside the regular bound; the saved state itself is ill-conditioned in float32 (the reference misses its own
    float64 run by 30x the bound), so it is held to the float64 reference relative to the reference's own miss."""
    g = load_golden("global_a_dec4")
    w = pkg.weight_variant("dec4")
    o, padded = _run(oracle, pkg, w, g)
    assert_state_close_conditioned(o.get_state(), g["state"], g["state_fp64"])
    o2 = oracle.Stylization(w)
    o2.set_state(g["state"])
    assert_pre_close(o2.transfer(padded, return_preclamp=True)[0], g["pre"])


def test_img1_256_matches_reference(oracle, pkg, weights):
    """BASELINE config 1: data/img_1.jpg (512x512) on ONE 256x256 natural frame; N = 1, so B = 1 and the sampled frame
    is the stylized one (generate_real_video.py:129-146 adds the last frame always)."""
    g = load_golden("img1_256")
    style, frame = decode_png(g["style_png"]), decode_png(g["frame_png"])
    assert style.shape == (512, 512, 3) and frame.shape == (256, 256, 3)
    oracle.set_conv_backend("torch")
    try:
        o = oracle.Stylization(weights)
        o.prepare_style(style)
        o.clean()
        o.add(frame)
        o.compute()
        assert_state_close(o.get_state(), g["state"])
        padded = oracle.reflect_pad(frame, 384, 384)
        pre = o.transfer(padded, return_preclamp=True)[0][64:320, 64:320]
        out = o.transfer(padded)[64:320, 64:320]
    finally:
        oracle.set_conv_backend("numpy")
    assert_pre_close(pre[::2, ::2], g["pre_grid"])
    np.testing.assert_allclose(pre.mean(axis=(0, 1)), g["pre_chanmean"], atol=2e-5)
    assert np.abs(out[::2, ::2] - g["out_grid"]).max() <= IMG_ATOL


def test_real_default_matches_reference(oracle, pkg, weights):
    """The reference's default invocation (generate_real_video.py:19,24): plum_flower.jpg at its native 400x564 on the
    ambush_4 frames at 436x1024 (neither a multiple of 8), sampled 0, 8, 16, 24 + last, frame 12 padded to 576x1152."""
    g = load_golden("real_default")
    style = decode_png(g["style_png"])
    ids, tid = [int(i) for i in g["sample_ids"]], int(g["transfer_id"])
    assert style.shape == (400, 564, 3) and ids == oracle.sample_indices(33) == [0, 8, 16, 24, 32]
    oracle.set_conv_backend("torch")
    try:
        o = oracle.Stylization(weights)
        o.prepare_style(style)
        o.clean()
        for i in ids:
            f = decode_png(g["frame%d_png" % i])
            assert f.shape == (436, 1024, 3)
            o.add(f)                                   # unpadded (Q6)
        o.compute()
        assert_state_close(o.get_state(), g["state"])
        np.testing.assert_allclose(o.F_style["map"].sum(axis=(0, 1, 2)), g["style_map_chansum"], rtol=1e-4, atol=1e-3)
        padded = oracle.reflect_pad(decode_png(g["frame%d_png" % tid]), 576, 1152)
        y = o.transfer(padded, return_preclamp=True)
        pre, out = y[0][64:500, 64:1088], oracle.tensor_to_image(y)[64:500, 64:1088]
    finally:
        oracle.set_conv_backend("numpy")
    assert_pre_close(pre[::4, ::4], g["pre_grid"])
    assert_pre_close(pre[186:250, 480:544], g["pre_patch"])
    np.testing.assert_allclose(pre.mean(axis=(0, 1)), g["pre_chanmean"], atol=2e-5)
    assert np.abs(out[::4, ::4] - g["out_grid"]).max() <= IMG_ATOL
    assert np.abs(out[186:250, 480:544] - g["out_patch"]).max() <= IMG_ATOL


def test_real_multistyle_matches_reference(oracle, pkg, weights):
    """"Multi-style Interpolation/test.py" on real images: two styles from the reference's data/ at 384x384, the ambush_4
    frames padded to 576x1152, features 0, 16, 32 + the last again for the statistics, frame 7 with the script's ramp."""
    g = load_golden("real_multistyle")
    styles = [decode_png(g["style%d_png" % k]) for k in range(2)]
    ids, tid = [int(i) for i in g["sample_ids"]], int(g["transfer_id"])
    assert styles[0].shape == (384, 384, 3) and ids == oracle.sample_indices_multistyle(33) == [0, 16, 32, 32]
    wts = [float(v) for v in g["weights"]]
    np.testing.assert_allclose(wts, [tid / 32.0, 1 - tid / 32.0], atol=1e-7)        # test.py:127-131
    oracle.set_conv_backend("torch")
    try:
        o = oracle.MultiStylization(weights, 2)
        o.prepare_style(styles)
        feats = {i: o.generate_content_features(oracle.reflect_pad(decode_png(g["frame%d_png" % i]), 576, 1152)) for i in sorted(set(ids + [tid]))}
        o.clean()
        for i in ids:
            o.add_patch(feats[i])
        o.compute_norm()
        for k in range(2):
            assert_state_close(o.get_state(k), g["state%d" % k], "style %d" % k)
        y = o.transfer(feats[tid], wts, return_preclamp=True)
        pre, out = y[0][64:500, 64:1088], oracle.tensor_to_image(y)[64:500, 64:1088]
    finally:
        oracle.set_conv_backend("numpy")
    assert_pre_close(pre[::4, ::4], g["pre_grid"])
    assert_pre_close(pre[186:250, 480:544], g["pre_patch"])
    np.testing.assert_allclose(pre.mean(axis=(0, 1)), g["pre_chanmean"], atol=2e-5)
    assert np.abs(out[::4, ::4] - g["out_grid"]).max() <= IMG_ATOL
    assert np.abs(out[186:250, 480:544] - g["out_patch"]).max() <= IMG_ATOL


def test_real_frame_mode_matches_reference(oracle, pkg, weights):
    """use_Global=False on the reference's default inputs (plum_flower 400x564, ambush_4 frame 12 padded to 576x1152; the
    inputs are the ones stored in real_default.npz)."""
    g, gin = load_golden("real_frame_mode"), load_golden("real_default")
    tid = int(g["transfer_id"])
    oracle.set_conv_backend("torch")
    try:
        o = oracle.Stylization(weights, use_Global=False)
        o.prepare_style(decode_png(gin["style_png"]))
        padded = oracle.reflect_pad(decode_png(gin["frame%d_png" % tid]), 576, 1152)
        y = o.transfer(padded, return_preclamp=True)
        pre, out = y[0][64:500, 64:1088], oracle.tensor_to_image(y)[64:500, 64:1088]
    finally:
        oracle.set_conv_backend("numpy")
    assert_pre_close(pre[::4, ::4], g["pre_grid"])
    assert_pre_close(pre[186:250, 480:544], g["pre_patch"])
    np.testing.assert_allclose(pre.mean(axis=(0, 1)), g["pre_chanmean"], atol=2e-5)
    assert np.abs(out[::4, ::4] - g["out_grid"]).max() <= IMG_ATOL
    assert np.abs(out[186:250, 480:544] - g["out_patch"]).max() <= IMG_ATOL
