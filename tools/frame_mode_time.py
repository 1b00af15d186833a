import importlib, sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")
s = pkg.Stylization(pkg.synthetic_weights(0), cuda=True, use_Global=False)
s.prepare_style(pkg.synth_style(256, 256))
f = V.reflect_pad(pkg.synth_frame(0, 512, 512), 640, 640)
for i in range(3): s.transfer(f)
t = time.perf_counter(); n = 20
for i in range(n): s.transfer(f)
dt = time.perf_counter() - t
print("frame mode: %.1f frames/s (%.2f ms/frame)" % (n / dt, 1e3 * dt / n))
s.profile_begin(); s.transfer(f); rows = s.profile_end()
agg = {}
for name, ms, fl, by, fx in rows:
    k = name.split("@")[0]; a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += ms
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]: print("  %-62s x%-3d %.3f ms" % (k, a[0], a[1]))
print("  kernels total %.2f ms" % tot)
