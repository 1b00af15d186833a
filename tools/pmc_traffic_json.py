#!/usr/bin/env python
"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 --pmc passes into profiles/hbm_traffic_<cfg>.json:
    python tools/pmc_traffic_json.py <fetch.db> <write.db> "<bench command the passes ran>" > profiles/hbm_traffic_512.json
Per kernel name: average fabric-side bytes per launch = F * FETCH_SIZE*1024 + WRITE_SIZE*1024.  FETCH_SIZE is KiB and on
gfx950 tallies 128-byte requests at 64 B (MI355X_MICROARCH.md §HBM), so what it means depends on the access pattern —
calibrated here with tools/fetch_calib.hip (profiles/r03_fetch_size_calibration.txt: every kernel reads a known byte count
once from a 2 GiB buffer):
    contiguous >= 128 B per request (float4 streaming, 256-byte LDS-DMA segments)   FETCH_SIZE = 0.5 x bytes  ->  F = 2
    64-byte segments (the transform-domain kernels' LDS-DMA: 16 channels of a pixel)   FETCH_SIZE = 1.0 x bytes  ->  F = 1
    sparse 32-byte segments                                                            FETCH_SIZE = 2.0 x bytes  ->  F = 0.5
    conv_f43_k: every 32-byte piece of each line, chunk-major (r05_fetch_calib.txt)    FETCH_SIZE = 1.08-1.56 x   ->  F = 1 (upper bound)
    conv_f43_k<.., LAY & 1>: rows of a channel-chunk-major plane (r06_fetch_calib.txt)   FETCH_SIZE = 0.5 x bytes  ->  F = 2
(rounds 1-2 applied F = 2 to every kernel, which doubled the read side of the Winograd kernels: the "1.46x traffic" of
the dominant kernel was this artefact — with F = 1 it reads 1.27x its input (halo rows) and moves 1.09x its algorithmic bytes.)
bench.py reads this file to fill roofline.traffic for its dominant kernel (PMC counters cannot be collected inside the timed run)."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    per = defaultdict(float)
    for name, disp, val in c.execute("select name, dispatch_id, counter_value from pmc_events where counter_name=?", (counter,)):
        per[(name, disp)] += val
    agg = defaultdict(list)
    for (name, _), v in per.items():
        agg[name].append(v)
    return {n: (sum(v) / len(v), len(v)) for n, v in agg.items()}


def fetch_factor(name):
    """Bytes per FETCH_SIZE byte for the kernel's read pattern (see the module docstring)."""
    m = re.match(r"void conv_f43_k<\d+, (\d+)>", name)
    if m and int(m.group(1)) & 1:
        # channel-chunk-major input (round 6, conv_f43.h LAY & 1): a halo row is 1 152 contiguous bytes, the L2 fetches whole 128-byte lines.
        # Calibrated on that pattern (tools/fetch_calib.hip p8row_k, profiles/r06_fetch_calib.txt): FETCH_SIZE = 0.500 x the bytes.
        return 2.0
    if name.startswith("void conv_f43_k"):
        # 8-channel chunks: 32-byte pieces of every pixel, chunk after chunk over the same 128-byte lines.  Calibrated on that
        # pattern (tools/fetch_calib.hip chunkmajor_k, profiles/r05_fetch_calib.txt): FETCH_SIZE = 1.56 / 1.11 / 1.08 x the bytes
        # for 64 / 128 / 256 channels, i.e. F = 0.64 .. 0.93.  One kernel name serves layers of all three widths, so the file
        # carries F = 1: an upper bound on the read side.  (Round 4 used 0.5 — the factor of SPARSE 32-byte segments — and
        # reported less than the compulsory input.)
        return 1.0
    if name.startswith(("void conv_wino_k", "void conv_wino_split_k", "void conv_mfma_k")):
        return 1.0          # LDS-DMA of 16-channel chunks: 64-byte segments
    return 2.0              # conv_last_k (256-byte pixels), conv_first_k, streaming float4 kernels, blit copies


def main(fetch_db, write_db, cmd):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out = {"source": {"fetch": fetch_db, "write": write_db, "command": cmd,
                      "formula": "bytes = (F*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, averaged over launches; F = 1 for the 64-byte-segment "
                                 "LDS-DMA kernels and for conv_f43_k on NHWC input (upper bound: its pattern calibrates to 0.64-0.93), 2 for >= 128-byte "
                                 "contiguous reads incl. conv_f43_k on channel-chunk-major input (profiles/r05_fetch_calib.txt, r06_fetch_calib.txt)"},
           "kernels": {}}
    for n in sorted(set(f) | set(w)):
        fb = fetch_factor(n) * f.get(n, (0, 0))[0] * 1024
        wb = w.get(n, (0, 0))[0] * 1024
        out["kernels"][n] = {"bytes_per_launch": round(fb + wb), "read_bytes": round(fb), "write_bytes": round(wb), "fetch_factor": fetch_factor(n),
                             "launches": f.get(n, w.get(n))[1]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
