#!/usr/bin/env python
"""FETCH_SIZE per calibration kernel of tools/fetch_calib.hip next to the bytes it really read:
    python tools/fetch_calib_summary.py <pmc.db>
(the rocpd database of `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/bin/fetch_calib`)."""
import re
import sqlite3
import sys
from collections import defaultdict

BYTES = 2 << 30


def useful(name):
    if name.startswith("stream_k"):
        return BYTES
    if name.startswith("p8row_k"):
        return BYTES // (34 * 648 * 32) * 18 * 34 * 36 * 32      # whole 34-row bands of eighteen 36-pixel tiles
    if "chunkmajor_k" in name:
        return BYTES          # every byte once (2 GiB is a whole number of 1024-pixel blocks for 64 / 128 / 256 channels)
    m = re.search(r"seg_k<(\d+), (\d+)>", name)
    if not m:
        return None
    seg, stride = int(m.group(1)), int(m.group(2))
    return BYTES // stride * seg


def main(path):
    c = sqlite3.connect(path)
    per = defaultdict(float)
    for name, disp, val in c.execute("select name, dispatch_id, counter_value from pmc_events where counter_name='FETCH_SIZE'"):
        per[(name, disp)] += val
    print("# FETCH_SIZE (KiB x 1024) against the bytes each kernel reads exactly once from a 2 GiB buffer")
    print("%-34s %14s %14s %8s" % ("kernel", "useful bytes", "FETCH_SIZE B", "ratio"))
    for (name, _), v in sorted(per.items()):
        u = useful(name)
        if u:
            print("%-34s %14d %14d %8.3f" % (name[:34], u, v * 1024, v * 1024 / u))


if __name__ == "__main__":
    main(sys.argv[1])
