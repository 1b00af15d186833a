#!/usr/bin/env python
"""Host -> host rates of the boundary's entries (PCIe-inclusive) at one frame size: page-locked vs pageable caller
arrays, several frames-per-call, padded-frame and unpadded/cropped entries, one frame per call.
    python tools/host_entry_rate.py [--size 512] [--reps 6]
Prints one JSON object (commit it under profiles/)."""
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
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--batches", type=str, default="8,16,64,128")
    a = ap.parse_args()
    import torch
    pkg = importlib.import_module("rerevst-code_amd")
    V = importlib.import_module("rerevst-code_amd.video")
    S, P = a.size, V.padded_size(a.size)
    m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    m.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7))
    m.clean()
    for i in (0, 8, 16):
        m.add(pkg.synth_frame(i, S, S, kind="noise"))
    m.compute()
    res = {"size": S, "padded": P, "rows": []}
    maxb = max(int(b) for b in a.batches.split(","))
    raw = np.stack([pkg.synth_frame(i % 16, S, S, kind="noise") for i in range(maxb)])
    padded = np.stack([V.reflect_pad(f, P, P) for f in raw[:16]])
    padded = np.concatenate([padded] * ((maxb + 15) // 16))[:maxb]
    for B in [int(b) for b in a.batches.split(",")]:
        for kind in ("page_locked", "pageable"):
            alloc = pkg.pinned_empty if kind == "page_locked" else (lambda shp, dt: np.empty(shp, dt))
            for entry in ("transfer_batch", "transfer_frames"):
                src = padded if entry == "transfer_batch" else raw
                h_in = alloc(src[:B].shape, np.uint8)
                h_in[...] = src[:B]
                h_out = alloc(src[:B].shape, np.float32)
                fn = getattr(m, entry)
                fn(h_in, out=h_out)
                ts = []
                for _ in range(a.reps):
                    t0 = time.perf_counter()
                    fn(h_in, out=h_out)
                    ts.append(time.perf_counter() - t0)
                res["rows"].append({"entry": entry, "memory": kind, "frames_per_call": B,
                                    "frames_per_s_median": round(B / sorted(ts)[len(ts) // 2], 1), "frames_per_s_best": round(B / min(ts), 1)})
    # the reference's surface: one padded frame per call, fresh output array each time
    for kind in ("page_locked", "pageable"):
        f = pkg.pinned_empty((P, P, 3), np.uint8) if kind == "page_locked" else np.empty((P, P, 3), np.uint8)
        f[...] = padded[0]
        m.transfer(f)
        t0 = time.perf_counter()
        for _ in range(32):
            m.transfer(f)
        res["rows"].append({"entry": "transfer (one frame per call, fresh output array)", "memory": kind + " input",
                            "frames_per_s_median": round(32 / (time.perf_counter() - t0), 1)})
    # device-resident reference point
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(padded[:16]).to(dev).view(2, 8, P, P, 3)
    d_out = torch.empty((4, 8, P, P, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for i in range(2):
        m.transfer_batch_device(d_in[i % 2].data_ptr(), 8, P, P, d_out[i & 3].data_ptr())
    m.sync()
    t0 = time.perf_counter()
    for i in range(40):
        m.transfer_batch_device(d_in[i % 2].data_ptr(), 8, P, P, d_out[i & 3].data_ptr())
    m.sync()
    res["device_resident_frames_per_s"] = round(320 / (time.perf_counter() - t0), 1)
    m.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
