#!/usr/bin/env python
"""Per-kernel breakdown of Stylization(use_Global=False).transfer at 512x512 (padded 640x640), one frame per call:
HIP-event time per launch group inside the library + wall time per call.  Prints one JSON object."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = V.padded_size(S)
m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True, use_Global=False)
m.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7))
frames = [V.reflect_pad(pkg.synth_frame(i, S, S, kind="noise"), P, P) for i in range(4)]
for f in frames[:2]:
    m.transfer(f)
t0 = time.perf_counter()
n = 24
for i in range(n):
    m.transfer(frames[i % 4])
wall = (time.perf_counter() - t0) / n
m.profile_begin()
m.transfer(frames[0])
rows = m.profile_end()
agg = {}
for name, ms, fl, by, fx in rows:
    k = name.split("@")[0]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += ms
out = {"size": S, "wall_ms_per_frame": round(wall * 1e3, 3), "frames_per_s": round(1 / wall, 1), "launches_profiled": len(rows),
       "event_ms_total": round(sum(a[1] for a in agg.values()), 3),
       "kernels": [{"kernel": k, "launches": a[0], "ms": round(a[1], 4)} for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])]}
print(json.dumps(out))
