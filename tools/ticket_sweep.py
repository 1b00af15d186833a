#!/usr/bin/env python
"""Look-ahead tickets (rrv_transfer_async: the one-frame-per-call loop with frames in flight) under other work decompositions
(VERDICT r5 #1): tickets that may be open (RRV_TICKETS), persistent workgroups per ticket launch (RRV_TICKET_GRID), trimmed
grids (RRV_TRIM: ceil(items / rounds) workgroups).  At 512 x 512 (640 x 640 padded) every layer has 25 * 2^k work items, so a
grid of 64 workgroups (a quarter of the chip) runs 3.125 -> 4, 1.56 -> 2, 6.25 -> 7 rounds while a grid of 50 runs whole rounds.
    python tools/ticket_sweep.py [--size 512] [--frames 96]          (spawns one process per configuration)"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    pkg = importlib.import_module("rerevst-code_amd")
    V = importlib.import_module("rerevst-code_amd.video")
    S, P = a.size, V.padded_size(a.size)
    m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    m.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7))
    m.clean()
    for i in (0, 8, 16):
        m.add(pkg.synth_frame(i, S, S, kind="noise"))
    m.compute()
    n = a.depth + 1
    pins = [pkg.pinned_empty((P, P, 3), np.uint8) for _ in range(8)]
    outs = [pkg.pinned_empty((P, P, 3), np.float32) for _ in range(8)]
    for k in range(8):
        pins[k][...] = V.reflect_pad(pkg.synth_frame(k, S, S, kind="noise"), P, P)
    best = 0.0
    for rep in range(3):
        q = []
        t0 = time.perf_counter()
        for k in range(a.frames):
            q.append(m.transfer_async(pins[k & 7], out=outs[k & 7]))
            if len(q) > a.depth:
                m.result(q.pop(0))
        while q:
            m.result(q.pop(0))
        best = max(best, a.frames / (time.perf_counter() - t0))
    ref = np.array(m.transfer(pins[3]))
    t = m.transfer_async(pins[3], out=outs[0])
    same = bool(np.array_equal(np.array(m.result(t)), ref)) if os.environ.get("RRV_F43") in ("0", "2") else None
    print(json.dumps({"frames_per_s": round(best, 1), "in_flight": n, "same_bits_as_transfer": same}))
    m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=96)
    ap.add_argument("--depth", type=int, default=-1)
    a = ap.parse_args()
    if a.depth >= 0:
        return child(a)
    print("# tools/ticket_sweep.py: %d x %d, look-ahead tickets with page-locked buffers, best of 3 passes over %d frames; in flight = tickets open while one is collected" % (a.size, a.size, a.frames))
    configs = [(4, 0, 0), (4, 0, 1), (5, 0, 1), (5, 50, 0), (5, 50, 1), (5, 56, 1), (6, 0, 1), (6, 48, 1), (4, 50, 1), (4, 64, 1), (8, 0, 1), (3, 0, 1)]
    for tickets, grid, trim in configs:
        env = dict(os.environ, RRV_TICKETS=str(tickets), RRV_TRIM=str(trim))
        env.pop("RRV_TICKET_GRID", None)
        if grid:
            env["RRV_TICKET_GRID"] = str(grid)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--size", str(a.size), "--frames", str(a.frames), "--depth", str(tickets - 1)],
                           env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("RRV_TICKETS=%d RRV_TICKET_GRID=%-3s RRV_TRIM=%d  %s" % (tickets, grid or "-", trim, line[-1] if line else "FAILED: " + r.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
