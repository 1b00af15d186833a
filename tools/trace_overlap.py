#!/usr/bin/env python
"""How busy the GPU is in a rocprofv3 --kernel-trace database: wall span of the steady part, union of kernel intervals
(= time at least one kernel runs), sum of kernel durations (> union when kernels of two streams overlap), per queue.
    python tools/trace_overlap.py <results.db> [skip_first_fraction=0.3]"""
import sqlite3
import sys


def main(path, skip=0.3):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(c.execute("select start, end, name%s from kernels order by start" % ((", " + qcol) if qcol else "")))
    t0, t1 = rows[0][0], rows[-1][1]
    lo = t0 + skip * (t1 - t0)
    rows = [r for r in rows if r[0] >= lo]
    span = rows[-1][1] - rows[0][0]
    ssum = sum(r[1] - r[0] for r in rows)
    union, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, *_ in rows[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    nlast = sum(1 for r in rows if r[2].startswith("conv_last_k"))
    print("steady part: %.1f ms wall, %d frames (conv_last_k launches) -> %.3f ms per frame" % (span / 1e6, nlast, span / 1e6 / max(1, nlast)))
    print("  at least one kernel running %.1f%% of the wall; sum of kernel durations = %.2fx the wall (%.3f ms per frame)"
          % (100.0 * union / span, ssum / span, ssum / 1e6 / max(1, nlast)))
    if qcol:
        per = {}
        for r in rows:
            per.setdefault(r[3], [0, 0])
            per[r[3]][0] += r[1] - r[0]
            per[r[3]][1] += 1
        for q, (d, n) in sorted(per.items()):
            print("  %s %s: %d dispatches, busy %.1f%% of the wall" % (qcol, q, n, 100.0 * d / span))
    blit = sum(r[1] - r[0] for r in rows if "rocclr" in r[2])
    print("  copy (blit) kernels: %.3f ms per frame" % (blit / 1e6 / max(1, nlast)))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
