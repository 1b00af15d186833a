#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (rocpd sqlite) per kernel:
    python tools/pmc_summary.py db1 [db2 ...]
Each counter is summed over its hardware instances per dispatch, then averaged over the
dispatches of a kernel name.  FETCH_SIZE/WRITE_SIZE are KiB per the tool's definition; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 B (MI355X_MICROARCH.md §HBM): it is HALF the bytes of >= 128-byte
contiguous reads (column fetch_x2_MB) but EXACT for the 64-byte-segment LDS-DMA of the transform-domain kernels
(column fetch_MB; calibration: profiles/r03_fetch_size_calibration.txt, tools/fetch_calib.hip)."""
import sqlite3
import sys
from collections import defaultdict


def load(path):
    c = sqlite3.connect(path)
    per = defaultdict(lambda: defaultdict(float))      # (name, dispatch) -> counter -> sum
    dur = {}
    for name, disp, cn, val, d in c.execute(
            "select name, dispatch_id, counter_name, counter_value, duration from pmc_events"):
        per[(name, disp)][cn] += val
        dur[(name, disp)] = d
    agg = defaultdict(lambda: defaultdict(list))
    for (name, disp), cs in per.items():
        for cn, v in cs.items():
            agg[name][cn].append(v)
        agg[name]["_dur_us"].append(dur[(name, disp)] / 1e3)
    return agg


def main(paths):
    merged = defaultdict(dict)
    for p in paths:
        for name, cs in load(p).items():
            for cn, vals in cs.items():
                merged[name][cn] = (sum(vals) / len(vals), len(vals))
    names = sorted(merged, key=lambda n: -merged[n].get("_dur_us", (0, 0))[0] * merged[n].get("_dur_us", (0, 0))[1])
    counters = sorted({cn for n in merged for cn in merged[n] if not cn.startswith("_")})
    print("# per-dispatch averages; SQ_* cycle counters are in quad-cycles summed over all SIMDs (MFMA_BUSY in cycles)")
    for n in names:
        m = merged[n]
        if m["_dur_us"][0] * m["_dur_us"][1] < 50:
            continue
        print("%s  (n=%d, avg %.1f us under counters)" % (n[:90], m["_dur_us"][1], m["_dur_us"][0]))
        line = []
        for cn in counters:
            if cn in m:
                line.append("%s=%.4g" % (cn, m[cn][0]))
        print("    " + "  ".join(line))
        d = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            # MFMA_BUSY counts cycles per SIMD... normalise by (active cycles x 1024 SIMDs)
            gui = m["GRBM_GUI_ACTIVE"][0] / 8.0          # summed over 8 XCDs
            d.append("mfma_busy_frac=%.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (gui * 1024.0)))
        if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m:
            w = m["SQ_WAVE_CYCLES"][0]
            d.append("wait_any=%.3f wait_inst=%.3f active_inst=%.3f (of wave cycles)" % (
                m["SQ_WAIT_ANY"][0] / w, m.get("SQ_WAIT_INST_ANY", (0,))[0] / w, m.get("SQ_ACTIVE_INST_ANY", (0,))[0] / w))
        if "FETCH_SIZE" in m:
            d.append("fetch_MB=%.2f fetch_x2_MB=%.2f" % (m["FETCH_SIZE"][0] / 1024.0, 2 * m["FETCH_SIZE"][0] / 1024.0))
        if "WRITE_SIZE" in m:
            d.append("write_MB=%.2f" % (m["WRITE_SIZE"][0] / 1024.0))
        if d:
            print("    -> " + "  ".join(d))


if __name__ == "__main__":
    main(sys.argv[1:])
