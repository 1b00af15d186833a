#!/usr/bin/env python
"""The reference's one-frame-per-call surface (generate_real_video.py:152-171) under the microscope: host-side time per
phase of Stylization.transfer / transfer_async + result, and (run under `rocprofv3 --kernel-trace --stats`) the GPU kernel
time per frame at one frame per launch.
    python tools/one_frame_profile.py [--size 512] [--frames 64] [--mode sync|async|both]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--mode", default="both")
    a = ap.parse_args()
    pkg = importlib.import_module("rerevst-code_amd")
    V = importlib.import_module("rerevst-code_amd.video")
    S, P = a.size, V.padded_size(a.size)
    m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    m.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7))
    m.clean()
    for i in (0, 8, 16):
        m.add(pkg.synth_frame(i, S, S, kind="noise"))
    m.compute()
    frames = [V.reflect_pad(pkg.synth_frame(i % 16, S, S, kind="noise"), P, P) for i in range(a.frames)]
    res = {"size": S, "padded": P, "frames": a.frames}
    for f in frames[:3]:
        m.transfer(f)
    if a.mode in ("sync", "both"):
        t0 = time.perf_counter()
        for f in frames:
            m.transfer(f)
        res["sync_frames_per_s"] = round(a.frames / (time.perf_counter() - t0), 1)
        out = np.empty((P, P, 3), np.float32)
        pin_in, pin_out = pkg.pinned_empty((P, P, 3), np.uint8), pkg.pinned_empty((P, P, 3), np.float32)
        pin_in[...] = frames[0]
        import ctypes as C
        t0 = time.perf_counter()
        for _ in range(a.frames):       # the C entry alone, page-locked buffers: no staging copies, no numpy allocation
            m._chk(m._lib.rrv_transfer(m._h, pin_in.ctypes.data_as(C.c_void_p), P, P, pin_out.ctypes.data_as(C.c_void_p)))
        res["sync_c_entry_pinned_frames_per_s"] = round(a.frames / (time.perf_counter() - t0), 1)
    if a.mode in ("async", "both"):
        prev = None
        t_sub = t_res = 0.0
        t0 = time.perf_counter()
        for f in frames:
            t1 = time.perf_counter()
            tk = m.transfer_async(f)
            t2 = time.perf_counter()
            if prev is not None:
                m.result(prev)
            t3 = time.perf_counter()
            t_sub += t2 - t1
            t_res += t3 - t2
            prev = tk
        m.result(prev)
        dt = time.perf_counter() - t0
        res["lookahead_frames_per_s"] = round(a.frames / dt, 1)
        res["lookahead_ms_per_frame_in_submit"] = round(1e3 * t_sub / a.frames, 3)
        res["lookahead_ms_per_frame_in_result"] = round(1e3 * t_res / a.frames, 3)
        outs = [pkg.pinned_empty((P, P, 3), np.float32) for _ in range(4)]
        pins = [pkg.pinned_empty((P, P, 3), np.uint8) for _ in range(4)]
        for k in range(4):
            pins[k][...] = frames[k]
        prev = None
        t0 = time.perf_counter()
        for k in range(a.frames):
            tk = m.transfer_async(pins[k & 3], out=outs[k & 3])
            if prev is not None:
                m.result(prev)
            prev = tk
        m.result(prev)
        res["lookahead_pinned_frames_per_s"] = round(a.frames / (time.perf_counter() - t0), 1)
        for depth in (2, 3):
            q = []
            t0 = time.perf_counter()
            for k in range(a.frames):
                q.append(m.transfer_async(pins[k & 3], out=outs[k & 3]))
                if len(q) > depth:
                    m.result(q.pop(0))
            while q:
                m.result(q.pop(0))
            res["lookahead_pinned_depth%d_frames_per_s" % depth] = round(a.frames / (time.perf_counter() - t0), 1)
    print(json.dumps(res))
    m.close()


if __name__ == "__main__":
    main()
