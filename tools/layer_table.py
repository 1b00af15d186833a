#!/usr/bin/env python
"""Per-layer cost of one frame per launch against sixteen: for every launch of a frame the kernel, the
layer ("Cin x Cout @ H x W"), the library's HIP-event time at B = 1, at B = 16 divided by 16, and their ratio.
    python tools/layer_table.py [--size 512] [--reps 5] [--big 16]
Device-resident frames (rrv_transfer_batch_device), one stream, median over --reps profiled calls."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def profile(m, torch, P, B, reps):
    x = torch.randint(0, 256, (B, P, P, 3), dtype=torch.uint8, device="cuda")
    y = torch.empty((B, P, P, 3), dtype=torch.float32, device="cuda")
    for _ in range(2):
        m.transfer_batch_device(x.data_ptr(), B, P, P, y.data_ptr())
    m.sync()
    runs = []
    for _ in range(reps):
        m.profile_begin()
        m.transfer_batch_device(x.data_ptr(), B, P, P, y.data_ptr())
        runs.append(m.profile_end())
    names = [r[0] for r in runs[0]]
    ms = np.median(np.array([[r[1] for r in run] for run in runs]), axis=0)
    fx = [r[4] for r in runs[0]]
    return names, ms, fx


def items_rounds(name, B, n_cus=256):
    """(work items, rounds of the persistent grid) of a transform-domain launch, from its profile name "<kernel>@CinxCout@HxW"."""
    import math
    if "@" not in name:
        return None
    kern, cc, hw = name.split("@")
    cin, cout = (int(v) for v in cc.split("x"))
    H, W = (int(v) for v in hw.split("x"))
    if kern.startswith("conv_f43"):
        items, R = math.ceil(H / 32) * math.ceil(W / 32) * B * (cout // 32), n_cus
    elif kern.startswith("conv_upw"):
        items, R = math.ceil(H / 16) * math.ceil(W / 16) * B * (cout // 32), 2 * n_cus      # two workgroups per CU
    elif kern.startswith("conv_wino"):
        tiles = math.ceil(H / 16) * math.ceil(W / 16)
        split = (8 if tiles * 8 <= 320 else 4 if tiles * 4 <= 512 else 2 if tiles * 2 <= 256 else 1) if cout == 32 else 1      # filter_down's split K
        items, R = tiles * B * max(1, cout // 32) * split, n_cus
    else:
        return None
    return items, items / R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--big", type=int, default=16)
    a = ap.parse_args()
    import torch
    pkg = importlib.import_module("rerevst-code_amd")
    V = importlib.import_module("rerevst-code_amd.video")
    S, P = a.size, V.padded_size(a.size)
    m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    m.set_pipeline(1)
    m.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7))
    m.clean()
    for i in (0, 8, 16):
        m.add(pkg.synth_frame(i, S, S, kind="noise"))
    m.compute()
    n1, t1, f1 = profile(m, torch, P, 1, a.reps)
    nb, tb, fb = profile(m, torch, P, a.big, a.reps)
    print("# %dx%d padded to %dx%d; ms per frame at 1 frame per launch, at %d frames per launch / %d, ratio; executed TF/s at B = 1;" % (S, S, P, P, a.big, a.big))
    print("# items / rounds of the persistent grid (256 workgroups; 512 for the upsample-fused kernel) at B = 1 and what the rounds alone predict: quant = (ceil(rounds) / rounds at B = 1) / (the same at B = %d);" % a.big)
    print("# rest = ratio / quant = launch floor (~4-7 us per launch: sum_parts is nothing else) + pipeline fill of a one- or two-round launch")
    print("%-58s %-58s %9s %9s %6s %7s %7s %7s %6s %6s" % ("launch at B = 1", "launch at B = %d" % a.big, "B=1 ms", "B=%d/%d" % (a.big, a.big), "ratio", "TF/s", "items", "rounds", "quant", "rest"))
    tot1 = totb = 0.0
    for i in range(max(len(n1), len(nb))):
        a1 = n1[i] if i < len(n1) else "-"
        ab = nb[i] if i < len(nb) else "-"
        x1 = t1[i] if i < len(n1) else 0.0
        xb = tb[i] / a.big if i < len(nb) else 0.0
        tot1 += x1
        totb += xb
        tf = f1[i] / (x1 * 1e-3) / 1e12 if i < len(n1) and x1 > 0 else 0.0
        ir1, irb = items_rounds(a1, 1), items_rounds(ab, a.big)
        extra = ""
        if ir1 and irb and xb > 0:
            import math
            quant = (math.ceil(ir1[1]) / ir1[1]) / (math.ceil(irb[1]) / irb[1])
            extra = " %7d %7.2f %6.2f %6.2f" % (ir1[0], ir1[1], quant, x1 / xb / quant)
        print("%-58s %-58s %9.4f %9.4f %6.2f %7.1f%s" % (a1, ab, x1, xb, x1 / xb if xb > 0 else 0.0, tf, extra))
    print("%-117s %9.4f %9.4f %6.2f" % ("total", tot1, totb, tot1 / totb))
    m.close()


if __name__ == "__main__":
    main()
