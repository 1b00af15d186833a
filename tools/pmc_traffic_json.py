#!/usr/bin/env python
"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 --pmc passes into profiles/hbm_traffic.json:
    python tools/pmc_traffic_json.py <fetch.db> <write.db> "<bench command the passes ran>" > profiles/hbm_traffic.json
Per kernel name: average HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (FETCH_SIZE is KiB and
under-reports wide coalesced reads by 2x on gfx950: MI355X_MICROARCH.md §HBM).  bench.py reads this file to
fill roofline.traffic for its dominant kernel (PMC counters cannot be collected inside the timed run)."""
import json
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


def main(fetch_db, write_db, cmd):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out = {"source": {"fetch": fetch_db, "write": write_db, "command": cmd,
                      "formula": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, averaged over launches"},
           "kernels": {}}
    for n in sorted(set(f) | set(w)):
        fb = 2 * f.get(n, (0, 0))[0] * 1024
        wb = w.get(n, (0, 0))[0] * 1024
        out["kernels"][n] = {"bytes_per_launch": round(fb + wb), "read_bytes": round(fb), "write_bytes": round(wb),
                             "launches": f.get(n, w.get(n))[1]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
