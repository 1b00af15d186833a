#!/usr/bin/env python
"""The TAIL of the full-size parity error, measured (VERDICT r5 #2 iv): every frame of bench.py's first step — the 128
white-noise 512 x 512 frames of the 300-frame video, padded to 640 x 640, sixteen per launch, the bench's B = 38 state,
the library's default kernel choice — against the oracle with every convolution accumulated in float64 ("torch64": the
implementation's own error alone) and, for the reference arithmetic's own tail, the oracle on torch's float32 conv2d.
Per frame: worst pre-clamp error / bound, values outside the bound, 99.99th percentile, mean, image max |d| and values
beyond 0.05 grey levels; then the distribution over the frames.  (run on the GPU box; the oracle runs in a process pool)
    python tools/fullsize_tail.py [--frames 128] [--workers 8] [--threads 32] [--modes 1,0] [--size 512] [--per-launch 16]"""
import argparse
import importlib
import os
import sys
import time
import multiprocessing
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

_W = {}


def _oracle_init(threads, state):
    import torch
    torch.set_num_threads(threads)
    import rerevst_oracle as O
    pkg = importlib.import_module("rerevst-code_amd")
    o = O.Stylization(pkg.synthetic_weights(0))
    o.set_state(state)
    _W["o"], _W["O"], _W["pkg"] = o, O, pkg
    _W["V"] = importlib.import_module("rerevst-code_amd.video")


def _oracle_frame(job):
    i, S, P, backend = job
    O, o, pkg, V = _W["O"], _W["o"], _W["pkg"], _W["V"]
    frame = V.reflect_pad(pkg.synth_frame(i, S, S, kind="noise"), P, P)
    O.set_conv_backend(backend)
    try:
        pre = o.transfer(frame, return_preclamp=True)[0]
    finally:
        O.set_conv_backend("numpy")
    return i, backend, pre


def row(pre, ref64, img, img64, T):
    bound = T.PRE_ATOL + T.PRE_RTOL * np.abs(ref64.astype(np.float64))
    r = np.abs(pre.astype(np.float64) - ref64) / bound
    d = np.abs(img.astype(np.float64) - img64)
    return (float(r.max()), int((r > 1).sum()), float(np.percentile(r, 99.99)), float(r.mean()), float(d.max()), int((d > T.IMG_ATOL).sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--modes", default="1,0")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--per-launch", type=int, default=16)
    ap.add_argument("--no-f32-oracle", action="store_true")
    a = ap.parse_args()
    import state_bounds as T
    import rerevst_oracle as O
    pkg = importlib.import_module("rerevst-code_amd")
    V = importlib.import_module("rerevst-code_amd.video")
    S, P, G = a.size, V.padded_size(a.size), a.per_launch
    s = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    s.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7))
    s.clean()
    for i in V.sample_indices(300):
        s.add(pkg.synth_frame(i, S, S, kind="noise"))
    s.compute()
    state = s.get_state()
    ids = list(range(a.frames))
    # ---- the HIP path: launches of G frames, exactly the sub-batches of rrv_transfer_batch (a sub-batch's bits do not depend on the rest of a call)
    modes = [int(m) for m in a.modes.split(",")]
    got = {}
    for mode in modes:
        s.set_f43(mode)
        for g0 in range(0, a.frames, G):
            grp = ids[g0:g0 + G]
            frames = np.stack([V.reflect_pad(pkg.synth_frame(i, S, S, kind="noise"), P, P) for i in grp])
            out = np.array(s.transfer_batch(frames))
            for k, i in enumerate(grp):
                got[(mode, i)] = (np.array(s.preclamp(P, P, image=k)), out[k])
    s.close()
    # ---- the oracle, in a process pool
    t0 = time.time()
    backends = ["torch64"] + ([] if a.no_f32_oracle else ["torch"])
    refs = {}
    with ProcessPoolExecutor(a.workers, mp_context=multiprocessing.get_context("spawn"), initializer=_oracle_init, initargs=(a.threads, state)) as ex:      # spawn: no fork behind an initialised HIP runtime
        for i, be, pre in ex.map(_oracle_frame, [(i, S, P, be) for be in backends for i in ids]):
            refs[(be, i)] = pre
    print("# tools/fullsize_tail.py: %d white-noise %d x %d frames (padded %d x %d), %d per launch, the bench's B = 38 state; oracle: %d workers x %d threads, %.0f s"
          % (a.frames, S, S, P, P, G, a.workers, a.threads, time.time() - t0))
    print("# per frame: worst pre-clamp error / bound, values outside the bound, 99.99th percentile, mean | image max |d| (grey levels), values beyond %.2f" % T.IMG_ATOL)
    names = {1: "default kernel choice", 0: "F(2x2,3x3) everywhere", 2: "conv_f43_k everywhere"}
    table = {}
    for i in ids:
        r64 = refs[("torch64", i)]
        img64 = O.tensor_to_image(r64[None])
        cols = []
        for mode in modes:
            pre, img = got[(mode, i)]
            table.setdefault(names[mode], []).append(row(pre, r64, img, img64, T))
        if not a.no_f32_oracle:
            r32 = refs[("torch", i)]
            table.setdefault("the float32 oracle itself", []).append(row(r32, r64, O.tensor_to_image(r32[None]), img64, T))
        for name, rows in table.items():
            w, n, p, m, d, dn = rows[-1]
            cols.append("%s: %.3f / %d / %.3f / %.4f | %.4f / %d" % (name, w, n, p, m, d, dn))
        print("frame %3d  %s" % (i, "  ||  ".join(cols)), flush=True)
    print("\n## distribution over the %d frames" % a.frames)
    for name, rows in table.items():
        R = np.array(rows, dtype=np.float64)
        def q(col, pct):
            return float(np.percentile(R[:, col], pct))
        print("%-28s worst / bound: max %.3f, 99th pct %.3f, 90th %.3f, median %.3f; frames with a value outside the bound: %d, most values outside in one frame: %d (%.2e of its values), 99th pct %d;"
              " 99.99th-percentile max %.3f; mean max %.4f | image max |d|: max %.4f, 99th pct %.4f, median %.4f; frames with a value beyond %.2f: %d (most in one frame: %d)"
              % (name, R[:, 0].max(), q(0, 99), q(0, 90), q(0, 50), int((R[:, 1] > 0).sum()), int(R[:, 1].max()), R[:, 1].max() / (P * P * 3.0), int(q(1, 99)),
                 R[:, 2].max(), R[:, 3].max(), R[:, 4].max(), q(4, 99), q(4, 50), T.IMG_ATOL, int((R[:, 5] > 0).sum()), int(R[:, 5].max())))


if __name__ == "__main__":
    main()
