"""GPU tests of the C-ABI boundary's round-2 additions (run with -m gpu): page-locked caller buffers, stream-ordered
device entries, the active style after a blend, large frames (32-bit offset regression), the bounds-checked debug
mode, and the bench's N = 2 control flow with the HIP model."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden, golden_inputs, IMG_ATOL, fixed_kernels

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True, scope="module")
def _one_kernel_family():
    """This module asserts cross-entry invariants bit for bit (page-locked == pageable == tickets == zero copy == debug mode
    == sharded ...): they hold for a FIXED kernel choice, so every handle created here — in this process and in the child
    processes, through RRV_F43 — runs F(2x2,3x3).  The same entries with the DEFAULT choice meet the oracle in
    tests/test_gpu_default_choice.py::test_every_entry_in_the_default_mode_vs_oracle."""
    with fixed_kernels():
        yield


@pytest.fixture(scope="module")
def hip(pkg, weights):
    s = pkg.Stylization(weights, cuda=True)
    s.set_state(load_golden("global_a")["state"])
    yield s
    s.close()


def test_page_locked_buffers_equal_pageable(hip, pkg, oracle):
    """rrv_transfer_batch / rrv_transfer_frames with rrv_host_alloc'ed caller buffers (direct DMA, no staging) ==
    the staged path with pageable arrays; mixed (pinned in / pageable out) too."""
    frames = np.stack([oracle.reflect_pad(pkg.synth_frame(600 + i, 40, 56, kind="noise"), 128, 128) for i in range(19)])
    ref = hip.transfer_batch(frames)
    pin_in = pkg.pinned_empty(frames.shape, np.uint8)
    pin_in[...] = frames
    pin_out = pkg.pinned_empty(ref.shape, np.float32)
    pin_out[...] = -1.0
    assert hip.transfer_batch(pin_in, out=pin_out) is pin_out
    np.testing.assert_array_equal(pin_out, ref)
    np.testing.assert_array_equal(hip.transfer_batch(pin_in), ref)
    out2 = pkg.pinned_empty(ref.shape, np.float32)
    hip.transfer_batch(frames, out=out2)
    np.testing.assert_array_equal(out2, ref)
    raw = np.stack([pkg.synth_frame(620 + i, 40, 56, kind="noise") for i in range(11)])
    ref2 = hip.transfer_frames(raw)
    pr = pkg.pinned_empty(raw.shape, np.uint8)
    pr[...] = raw
    po = pkg.pinned_empty(ref2.shape, np.float32)
    hip.transfer_frames(pr, out=po)
    np.testing.assert_array_equal(po, ref2)
    np.testing.assert_array_equal(hip.transfer(pin_in[3]), ref[3])


def test_device_entry_ordered_against_caller_stream(pkg, weights, oracle):
    """rrv_set_caller_stream: the input is produced and the output consumed on a torch stream with NO host
    synchronisation in between; results must equal the synchronous path."""
    import torch
    g = load_golden("global_a")
    s = pkg.Stylization(weights, cuda=True)
    s.set_state(g["state"])
    frames = np.stack([oracle.reflect_pad(pkg.synth_frame(700 + i, 64, 64, kind="noise"), 192, 192) for i in range(4)])
    ref = s.transfer_batch(frames)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    s.set_caller_stream(st.cuda_stream, True)
    with torch.cuda.stream(st):
        for rep in range(6):
            h_in = torch.from_numpy(np.roll(frames, rep, axis=0)).pin_memory()     # a different batch every time
            want = np.roll(ref, rep, axis=0)
            big = torch.randn(4096, 4096, device=dev)
            for _ in range(3):
                big = big @ big * 1e-3                               # keeps the stream busy ahead of the copy
            d_in = h_in.to(dev, non_blocking=True)                   # produced on st, not yet complete on return
            d_out = torch.zeros((4, 192, 192, 3), dtype=torch.float32, device=dev)
            s.transfer_batch_device(d_in.data_ptr(), 4, 192, 192, d_out.data_ptr())
            total = d_out.sum(dtype=torch.float64)                   # consumed on st without rrv_sync
            got = d_out.cpu()
            np.testing.assert_array_equal(got.numpy(), want)
            assert abs(float(total) - float(want.astype(np.float64).sum())) <= 1e-6 * abs(float(want.astype(np.float64).sum()))
    s.set_caller_stream(0, False)
    s.close()


def test_plain_transfer_after_blend_restores_selected_style(pkg, weights, oracle):
    """after a blended transfer the plain entries must go back to the style that was active before it
    (here style 1, selected with set_state), not to the first computed one."""
    g = load_golden("multistyle_s2")
    s = pkg.Stylization(weights, cuda=True, style_num=2)
    s.set_state(g["state0"], 0)
    frame = oracle.reflect_pad(pkg.synth_frame(1, 64, 48, kind="smooth"), 192, 192)
    ref0 = s.transfer(frame)                       # style 0 is the active one
    s.clean()                                      # nothing active any more
    s.set_state(g["state1"], 1)                    # selects style 1
    s.set_state(g["state0"], 0)                    # loading another style's state does not steal the selection
    ref1 = s.transfer(frame)
    assert np.abs(ref1 - ref0).max() > 1.0         # the two styles really differ
    blended = s.transfer(frame, style_weight=[0.5, 0.5])
    assert np.abs(blended - ref1).max() > 0.5
    np.testing.assert_array_equal(s.transfer(frame), ref1)
    np.testing.assert_array_equal(s.transfer_batch([frame, frame])[1], ref1)
    s.close()


def test_large_frame_rows_beyond_the_2gib_mark(pkg, weights):
    """with (H+2)*(W+2)*256 B >= 2^31 the last kernel's 32-bit byte offsets used to wrap and the
    lower rows came out wrong.  A tall 9216 x 1024 frame whose content repeats every 64 rows must give bit-identical
    output rows 64*k apart wherever the receptive field sees only the periodic interior — in particular for rows
    BEYOND the 2 GiB mark of the 64-channel full-resolution tensors (row 8176 on)."""
    H, W = 9216, 1024
    mark = 2 ** 31 // ((W + 2) * 256)
    assert mark < H - 512
    g = load_golden("global_a")
    s = pkg.Stylization(weights, cuda=True)
    s.set_state(g["state"])
    strip = pkg.synth_frame(900, 64, W, kind="smooth")
    frame = np.tile(strip, (H // 64, 1, 1))
    out = s.transfer(frame)
    assert out.shape == (H, W, 3) and np.isfinite(out).all()
    top = out[512:576]                                   # rows far from both borders
    assert float(top.std()) > 1.0
    for y0 in (4096, 8192, 8448, 8640):                  # the last three lie beyond the mark, >= 512 rows above the bottom border
        assert y0 % 64 == 0 and y0 + 64 <= H - 512
        np.testing.assert_array_equal(out[y0:y0 + 64], top)
    assert 8192 > mark
    with pytest.raises(pkg.RRVError, match="too large"):
        s.transfer(np.zeros((8, 2 ** 22, 3), np.uint8))           # (H+2)(W+2)*64 >= 2^31: refused up front, not wrapped
    s.close()


def test_bench_two_rank_control_flow_with_hip_model(tmp_path):
    """`python bench.py --gpus 2` started plainly launches its own two ranks (both on GPU 0 here, gloo for the
    broadcast): rank 0 prepares and broadcasts the state, rank 1 set_state()s it; the JSON line reports n_gpus = 2."""
    env = dict(os.environ, RRV_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "256",
           "--batch", "8", "--frames", "20", "--no-cpu-baseline", "--profile-steps", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["roofline"]["frac"] < 1.0 and j["roofline"]["algorithmic_speedup"] > 1.0


def test_sharded_video_two_ranks_equals_single_process(tmp_path, pkg, weights):
    """video.stylize_video with the HIP model on two ranks (gloo broadcast of the blob, both ranks on GPU 0):
    rank 1's frames, stylized with the state it RECEIVED, equal a single-process run bit for bit."""
    code = r'''
import os, sys, importlib, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")
D = importlib.import_module("rerevst-code_amd.dist")
r, w, _ = D.init_from_env("gloo")
m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True, device=0)
frames = [pkg.synth_frame(i, 40, 56, kind="smooth") for i in range(9)]
out = V.stylize_video(m, frames, pkg.synth_style(48, 48, kind="smooth"), rank=r, world=w, broadcast=D.broadcast_state)
np.savez(os.path.join(%r, "rank%%d.npz" %% r), state=m.get_state(), **{"f%%d" %% k: v for k, v in out.items()})
m.close()
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
''' % (ROOT, str(tmp_path))
    port = 29700 + os.getpid() % 200
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    V = importlib.import_module("rerevst-code_amd.video")
    s = pkg.Stylization(weights, cuda=True)
    frames = [pkg.synth_frame(i, 40, 56, kind="smooth") for i in range(9)]
    ref = V.stylize_video(s, frames, pkg.synth_style(48, 48, kind="smooth"))
    r0, r1 = (np.load(tmp_path / ("rank%d.npz" % r)) for r in (0, 1))
    np.testing.assert_array_equal(r0["state"], r1["state"])
    np.testing.assert_array_equal(r0["state"], s.get_state())
    got = {int(k[1:]): r[k] for r in (r0, r1) for k in r.files if k.startswith("f")}
    assert sorted(got) == list(range(9))
    assert sorted(int(k[1:]) for k in r1.files if k.startswith("f")) == [4, 5, 6, 7, 8]
    for i in range(9):
        np.testing.assert_array_equal(got[i], ref[i])
    s.close()


def test_bounds_checked_debug_mode(pkg, weights, oracle):
    """rrv_set_debug(2): guard bands around every activation tensor, zero-ring / slack verification and a stream
    sync after EVERY kernel launch, over the whole flow on sizes with partial tiles everywhere; same bits as the
    unchecked run, and the checker's self-test (a planted ring store and a planted guard-band store) must fire."""
    style = pkg.synth_style(40, 56, kind="smooth", seed=11)
    sampled = [pkg.synth_frame(i, 37, 53, kind="smooth", seed=50) for i in range(3)]
    frame = pkg.synth_frame(9, 200, 136, kind="smooth", seed=50)
    raw = [pkg.synth_frame(30 + i, 67, 33, kind="noise", seed=50) for i in range(3)]
    def flow(level):
        s = pkg.Stylization(weights, cuda=True, style_num=2)
        s.set_debug(level)
        if level:
            s.debug_selftest()
        s.prepare_style([style, style[::-1].copy()])
        s.clean()
        for f in sampled:
            s.add(f)
        s.compute()
        outs = [s.get_state(0), s.get_state(1), s.transfer(frame), s.transfer_batch([frame] * 3), s.transfer_frames(raw),
                s.transfer(frame, style_weight=[0.25, 0.75])]
        s.set_workspace_cap(1)                      # the streaming preparation pass under the checker as well
        s.clean()
        for f in sampled:
            s.add(f)
        s.compute()
        outs.append(s.get_state(0))
        s.close()
        fm = pkg.Stylization(weights, cuda=True, use_Global=False)
        fm.set_debug(level)
        fm.prepare_style(style)
        outs.append(fm.transfer(frame))
        fm.close()
        return outs
    plain, checked = flow(0), flow(2)
    for a, b in zip(plain, checked):
        np.testing.assert_array_equal(a, b)


# ---- round 3 -----------------------------------------------------------------------------------------------------
def test_blend_transfer_of_a_frame_whose_size_is_not_a_multiple_of_8(pkg, weights, oracle):
    """transfer(frame, style_weight) at 67x93: the stylized frame is 64x88 and the host copy must be sized for THAT
    (the D2H copy was sized H*W*3 floats — a heap overflow past the caller's array)."""
    style = pkg.synth_style(40, 56, kind="smooth", seed=11)
    sampled = [pkg.synth_frame(i, 37, 53, kind="smooth", seed=50) for i in range(2)]
    s = pkg.Stylization(weights, cuda=True, style_num=2)
    s.prepare_style([style, style[::-1].copy()])
    s.clean()
    for f in sampled:
        s.add(f)
    s.compute()
    frame = pkg.synth_frame(5, 67, 93, kind="smooth", seed=51)
    guard = np.full((64 * 88 * 3 + 4096,), -7.0, np.float32)        # the result lands in the middle of a canary buffer
    out = guard[1024:1024 + 64 * 88 * 3].reshape(64, 88, 3)
    a = np.ascontiguousarray(frame)
    import ctypes as C
    w = (C.c_float * 2)(0.25, 0.75)
    s._chk(s._lib.rrv_transfer_blend(s._h, a.ctypes.data_as(C.c_void_p), 67, 93, w, 2, out.ctypes.data_as(C.c_void_p)))
    assert (guard[:1024] == -7.0).all() and (guard[1024 + 64 * 88 * 3:] == -7.0).all()
    got = s.transfer(frame, style_weight=[0.25, 0.75])
    assert got.shape == (64, 88, 3)
    np.testing.assert_array_equal(got, out)
    o = oracle.MultiStylization(weights, 2)
    o.prepare_style([style, style[::-1].copy()])
    o.clean()
    for f in sampled:
        o.add_patch(o.generate_content_features(f))
    o.compute_norm()
    ref = o.transfer(o.generate_content_features(frame), [0.25, 0.75])
    assert ref.shape == (64, 88, 3) and np.abs(got - ref).max() <= IMG_ATOL
    with pytest.raises(pkg.RRVError):
        s.transfer(np.zeros((7, 40, 3), np.uint8), style_weight=[0.5, 0.5])
    s.close()


def test_allocation_failure_leaves_no_half_built_workspace(hip, pkg, oracle):
    """rrv_debug_fail_alloc: an out-of-memory in the middle of building a workspace is RRV_E_NOMEM, and the next call
    with the SAME geometry rebuilds it (the plan recorded its geometry before the allocations, so the
    retry passed the cache test and launched kernels on null tensors)."""
    frame = oracle.reflect_pad(pkg.synth_frame(900, 40, 56, kind="smooth"), 104, 120)      # a geometry no other test of this handle uses
    fails = 0
    for nth in range(1, 64):             # every allocation of the chain fails once: staging, 9 encoder tensors, 14 decoder ones, the split-K partials
        hip.debug_fail_alloc(nth)
        try:
            hip.transfer_batch([frame, frame, frame])
        except pkg.RRVError as e:
            assert e.code == -5 and "out of device memory" in str(e), str(e)
            fails += 1
            continue
        finally:
            hip.debug_fail_alloc(0)
        break
    assert fails >= 12, fails
    ref = hip.transfer(frame)
    got = hip.transfer_batch([frame, frame, frame])
    for k in range(3):
        np.testing.assert_array_equal(got[k], ref)


def test_two_frame_sizes_alternate_without_reallocating(hip, pkg, oracle):
    """Two workspace geometries stay resident per slot: alternating between two frame sizes gives the same bits as
    running each size alone, and a third size evicts only the least recently used one."""
    a = oracle.reflect_pad(pkg.synth_frame(910, 40, 56, kind="smooth"), 128, 128)
    b = oracle.reflect_pad(pkg.synth_frame(911, 60, 50, kind="smooth"), 192, 128)
    c = oracle.reflect_pad(pkg.synth_frame(912, 24, 24, kind="smooth"), 192, 192)
    ra, rb, rc = hip.transfer(a), hip.transfer(b), hip.transfer(c)
    hip.debug_fail_alloc(1)              # from here on ANY device allocation would fail ...
    try:
        for _ in range(3):               # ... but c (just used) and b alternate inside the two resident plans
            np.testing.assert_array_equal(hip.transfer(c), rc)
            np.testing.assert_array_equal(hip.transfer(b), rb)
    finally:
        hip.debug_fail_alloc(0)
    np.testing.assert_array_equal(hip.transfer(a), ra)


def test_async_tickets_equal_plain_transfer(hip, pkg, oracle):
    """rrv_transfer_async / rrv_transfer_wait (the look-ahead loop of a one-frame-per-call driver): same bits as
    transfer(), in any await order, with more submissions than staging sets, pageable and page-locked outputs."""
    frames = [oracle.reflect_pad(pkg.synth_frame(920 + i, 40, 56, kind="noise"), 128, 128) for i in range(11)]
    ref = [hip.transfer(f) for f in frames]
    prev, got = None, []
    for f in frames:                                   # the driver loop: submit i+1, then collect i
        t = hip.transfer_async(f)
        if prev is not None:
            got.append(hip.result(prev))
        prev = t
    got.append(hip.result(prev))
    for k in range(11):
        np.testing.assert_array_equal(got[k], ref[k])
    tickets = [hip.transfer_async(f) for f in frames[:4]]          # four open tickets, collected backwards
    for k in (3, 1, 2, 0):
        np.testing.assert_array_equal(hip.result(tickets[k]), ref[k])
    tickets = [hip.transfer_async(f) for f in frames[:7]]          # more than four: the oldest are retired by later submissions
    for k in range(7):
        np.testing.assert_array_equal(hip.result(tickets[k]), ref[k])
    pin = pkg.pinned_empty((2,) + ref[0].shape, np.float32)
    t0 = hip.transfer_async(frames[0], out=pin[0])
    t1 = hip.transfer_async(frames[1], out=pin[1])
    batch = hip.transfer_batch(frames[2:5])                        # another entry in between retires the open tickets first
    np.testing.assert_array_equal(hip.result(t1), ref[1])
    np.testing.assert_array_equal(hip.result(t0), ref[0])
    for k in range(3):
        np.testing.assert_array_equal(batch[k], ref[2 + k])
    odd = pkg.synth_frame(940, 67, 93, kind="noise")               # 8*(H/8) x 8*(W/8) output
    np.testing.assert_array_equal(hip.result(hip.transfer_async(odd)), hip.transfer(odd))
    hip.set_pipeline(1)                                            # tickets keep their own streams whatever the pipeline depth is
    try:                                                           # (found by tools/soak_host_entries.py: the completion event sat on an idle stream)
        tickets = [hip.transfer_async(f) for f in frames[:4]]
        for k in range(4):
            np.testing.assert_array_equal(hip.result(tickets[k]), ref[k])
    finally:
        hip.set_pipeline(2)
    t = hip.transfer_async(frames[0])
    np.testing.assert_array_equal(hip.result(t), ref[0])
    np.testing.assert_array_equal(hip.result(t), ref[0])           # waiting twice is harmless
    with pytest.raises(pkg.RRVError) as e:                         # a ticket that was never issued
        hip.result((t[0] + 1000, t[1]))
    assert e.value.code == -1 and "no such ticket" in str(e.value)   # its own message, not a stale rrv_last_error


def test_async_ticket_input_may_be_overwritten_and_ticket_dropped(hip, pkg, oracle):
    """include/rerevst_hip.h: `frame_bgr` may be reused as soon as rrv_transfer_async returns — also a PAGE-LOCKED one,
    which is the direct source of the asynchronous H2D copy (the next frame decoded into the same pinned
    buffer corrupted the input).  And a ticket the caller drops keeps its output block until it is retired."""
    import gc
    frames = [oracle.reflect_pad(pkg.synth_frame(960 + i, 200, 264, kind="noise"), 384, 384) for i in range(6)]
    ref = [hip.transfer(f) for f in frames]
    buf = pkg.pinned_empty(frames[0].shape, np.uint8)              # ONE page-locked input buffer, refilled for every frame
    tickets = []
    for f in frames:
        buf[...] = f
        tickets.append(hip.transfer_async(buf))
        buf[...] = 255 - f                                         # overwritten the moment the call returns
    for k in range(6):
        np.testing.assert_array_equal(hip.result(tickets[k]), ref[k])
    # dropped tickets: the outputs stay referenced by the Stylization object while the GPU writes them
    for rep in range(3):
        for f in frames[:4]:
            hip.transfer_async(f)                                   # ticket discarded at once
        gc.collect()
        junk = [pkg.pinned_empty(ref[0].shape, np.float32) for _ in range(2)]      # pool / allocator churn next to the open tickets
        del junk
    last = hip.transfer_async(frames[5])
    np.testing.assert_array_equal(hip.result(last), ref[5])
    hip.sync()


def test_feature_cache_cap_falls_back_to_reencoding(pkg, weights, oracle):
    """rrv_set_feature_cache_cap with a deliberately tiny cap: features beyond it are kept as pixels and re-encoded at
    every use (the reference spills to disk and is unbounded, test.py:87-101) — same images to rounding, never an error."""
    styles = [pkg.synth_style(64, 64, kind="smooth", seed=7 + k) for k in range(2)]
    frames = [oracle.reflect_pad(pkg.synth_frame(i, 64, 48, kind="smooth"), 192, 192) for i in range(5)]
    def flow(cap):
        s = pkg.MultiStyleStylization(weights, cuda=True, style_num=2)
        if cap is not None:
            s.set_feature_cache_cap(cap)
        s.prepare_style(styles)
        feats = [s.generate_content_features(f) for f in frames]
        info = s.feature_cache_info()
        s.clean()
        for i in (0, 2, 4):
            s.add_patch(feats[i])
        s.compute_norm()
        st = [s.get_state(k) for k in range(2)]
        one = s.transfer(feats[1], [0.3, 0.7])
        many = s.transfer_many(feats, [[k / 4.0, 1 - k / 4.0] for k in range(5)])
        s.release_features()
        assert s.feature_cache_info() == (0, 0, 0)
        s.close()
        return info, st, one, many
    info_all, st_all, one_all, many_all = flow(None)
    per = (24 + 2) * (24 + 2) * 512 * 4
    info_cap, st_cap, one_cap, many_cap = flow(2 * per + 3 * 20 * 46 * 512 * 4)        # room for two features
    assert info_all[:2] == (5, 0) and info_cap[:2] == (2, 3)
    for a, b in zip(st_all, st_cap):                                    # spilled features were re-encoded for add_patch: same encoder, same bits
        np.testing.assert_array_equal(a, b)
    # a re-encoded frame normalises inside the encoder's epilogue instead of a pointwise pass over the cached feature
    assert np.abs(one_cap - one_all).max() <= IMG_ATOL and np.abs(many_cap - many_all).max() <= IMG_ATOL
    np.testing.assert_array_equal(many_cap[:2], many_all[:2])           # the two cached ones are untouched


def test_batched_feature_caching_equals_per_frame(pkg, weights, oracle):
    """rrv_generate_content_features_batch (sub-batches over two streams, the encoder's last layer storing straight into the
    cache arena) == rrv_generate_content_features frame by frame: the decoder on either feature gives the same bits; page-locked
    and pageable input, a call that crosses the cache cap (the tail kept as pixels), features of an earlier call still valid,
    release and re-use."""
    g = load_golden("multistyle_s2")
    styles = [pkg.synth_style(64, 64, kind="smooth", seed=7), pkg.synth_style(64, 64, kind="smooth", seed=8)]
    frames = np.stack([oracle.reflect_pad(pkg.synth_frame(40 + i, 100, 150, kind="noise"), 256, 320) for i in range(21)])
    s = pkg.MultiStyleStylization(weights, cuda=True, style_num=2)
    s.set_state(g["state0"], 0)
    s.set_state(g["state1"], 1)
    w = [0.3, 0.7]
    one = [s.generate_content_features(f) for f in frames]
    ref = np.stack([s.transfer(f, w) for f in one])
    s.release_features()
    first = s.generate_content_features_batch(frames[:5])                       # pageable, one sub-batch
    pin = pkg.pinned_empty(frames.shape, np.uint8)
    pin[:] = frames
    rest = s.generate_content_features_batch(pin)                                # page-locked, several sub-batches (ragged last one)
    assert [f.id for f in first] == list(range(5)) and [f.id for f in rest] == list(range(5, 26))
    np.testing.assert_array_equal(s.transfer_many(rest, [w] * 21), ref)
    np.testing.assert_array_equal(s.transfer_many(first, [w] * 5), ref[:5])     # the earlier arena is untouched
    res, sp, nbytes = s.feature_cache_info()
    assert (res, sp) == (26, 0)
    s.release_features()
    assert s.feature_cache_info() == (0, 0, 0)
    s.set_feature_cache_cap(nbytes // 26 * 9)                                    # room for about eight features
    capped = s.generate_content_features_batch(frames)
    res, sp, _ = s.feature_cache_info()
    assert 6 <= res <= 9 and res + sp == 21
    np.testing.assert_array_equal(s.transfer_many(capped[:res], [w] * res), ref[:res])
    assert np.abs(s.transfer_many(capped[res:], [w] * sp) - ref[res:]).max() <= IMG_ATOL      # spilled: re-encoded on use (the per-frame path's encoder)
    s.release_features()
    s.close()


def test_c_abi_rccl_broadcast_single_rank(hip, pkg):
    """rrv_comm_unique_id -> rrv_comm_init_rank -> rrv_broadcast_state (ncclBroadcast through the dlopen'ed librccl) at
    world size 1: the north star's RCCL broadcast as a C symbol; the state is unchanged and transfers still work."""
    before = hip.get_state()
    comm = hip.comm_init_rank(hip.comm_unique_id(), 1, 0)
    hip.broadcast_state(comm, 0, 0, 0)
    hip.comm_destroy(comm)
    np.testing.assert_array_equal(hip.get_state(), before)
    with pytest.raises(pkg.RRVError):
        hip.broadcast_state(None, 0, 0, 0)


def test_bench_nccl_process_group_at_world_size_one():
    """bench.py forced through init_process_group("nccl", device_id=...), barrier, broadcast, all_gather, all_reduce on
    CUDA tensors at world size 1 (RCCL runs fine with one rank): the branch the 8-GPU run takes, executed here."""
    env = dict(os.environ, RRV_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29900 + os.getpid() % 90))
    env.pop("RRV_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--size", "256",
           "--batch", "8", "--frames", "20", "--no-cpu-baseline", "--no-extras", "--profile-steps", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["value"] > 0
    pr = j["per_rank"]
    assert pr["process_group"] == "nccl" and len(pr["frames_per_s"]) == 1 and pr["min"] == pr["max"] > 0
    assert pr["c_abi_rccl_broadcast"].startswith("ok") and "bit-identical" in pr["c_abi_rccl_broadcast"], pr


def test_zero_copy_host_io_equals_staged(hip, pkg, oracle):
    """rrv_set_host_io(1): the kernels read the frames from / write the results to page-locked host memory directly
    (no H2D / D2H copies).  Every host entry gives the same bits as the staged mode: page-locked and pageable caller
    arrays, ragged sub-batches, the pad / crop entry, one frame per call, look-ahead tickets."""
    frames = np.stack([oracle.reflect_pad(pkg.synth_frame(950 + i, 40, 56, kind="noise"), 128, 128) for i in range(19)])
    raw = np.stack([pkg.synth_frame(970 + i, 67, 33, kind="noise") for i in range(5)])
    ref, ref_raw = hip.transfer_batch(frames), hip.transfer_frames(raw)
    hip.set_host_io(1)
    try:
        np.testing.assert_array_equal(hip.transfer_batch(frames), ref)                  # pageable in / out
        pin_in = pkg.pinned_empty(frames.shape, np.uint8)
        pin_in[...] = frames
        pin_out = pkg.pinned_empty(ref.shape, np.float32)
        pin_out[...] = -1.0
        hip.transfer_batch(pin_in, out=pin_out)                                          # page-locked in / out: no staging at all
        np.testing.assert_array_equal(pin_out, ref)
        np.testing.assert_array_equal(hip.transfer_batch(pin_in), ref)                  # mixed
        np.testing.assert_array_equal(hip.transfer_frames(raw), ref_raw)
        np.testing.assert_array_equal(hip.transfer(frames[7]), ref[7])
        tickets = [hip.transfer_async(frames[k]) for k in range(6)]
        for k in (5, 0, 3, 1, 2, 4):
            np.testing.assert_array_equal(hip.result(tickets[k]), ref[k])
        t = hip.transfer_async(pin_in[9], out=pin_out[0])
        np.testing.assert_array_equal(hip.result(t), ref[9])
        for mode in (2, 3):                                                              # one direction zero copy, the other staged
            hip.set_host_io(mode)
            np.testing.assert_array_equal(hip.transfer_batch(frames), ref)
            pin_out[...] = -1.0
            hip.transfer_batch(pin_in, out=pin_out)
            np.testing.assert_array_equal(pin_out, ref)
            np.testing.assert_array_equal(hip.transfer_frames(raw), ref_raw)
            np.testing.assert_array_equal(hip.transfer(frames[7]), ref[7])
    finally:
        hip.set_host_io(0)
    np.testing.assert_array_equal(hip.transfer_batch(frames), ref)


def test_destroy_returns_the_device_memory(pkg, weights, oracle):
    """rrv_destroy frees what the handle allocated (weights and their packs, state sets, workspaces, staging): a host that
    opens and closes a model per video must not leak HBM.  Measured with hipMemGetInfo through torch."""
    import torch
    frame = oracle.reflect_pad(pkg.synth_frame(77, 64, 48, kind="smooth"), 192, 192)
    style = pkg.synth_style(64, 64, kind="smooth", seed=3)
    def cycle():
        s = pkg.Stylization(weights, cuda=True, use_Global=True)
        s.prepare_style(style)
        s.clean(); s.add(frame[64:128, 64:112]); s.compute()
        s.transfer(frame)
        s.transfer_batch([frame] * 3)
        s.result(s.transfer_async(frame))
        s.close()
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]
    cycle()                                   # the first one also pays for the runtime's own one-time allocations
    free1 = cycle()
    for _ in range(3):
        free2 = cycle()
    assert free1 - free2 < (8 << 20), (free1, free2)      # < 8 MiB drift over three more create/destroy cycles (one handle holds ~1.3 GB)


def test_random_sequence_of_host_entries_is_bit_exact():
    """tools/soak_host_entries.py, short form: a random mix of transfer / transfer_batch (pageable and page-locked) /
    transfer_frames / look-ahead tickets collected in any order / host-I/O modes / pipeline depths over three frame
    geometries — every output bit-identical to the plain one-frame transfer()."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("soak_host_entries", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak_host_entries.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(iters=250, seed=11, verbose=False)


def test_streams_in_flight_do_not_disturb_each_other():
    """tools/device_stream_stress.py, short form: 24 000 small frames, one per launch, on four streams with full-size
    grids (the most kernel-to-kernel overlap the library can produce) — every frame bit-identical to its single-stream
    result.  Regression test of the round-3 find: a missing barrier between the first work item's prologue and its first
    LDS-DMA in the transform-domain kernels made about one such frame in 10^4 slightly wrong."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("device_stream_stress", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "device_stream_stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(iters=2000, slots=4, share=1, verbose=False) == 0
    assert mod.run(iters=700, slots=2, share=1, verbose=False) == 0
