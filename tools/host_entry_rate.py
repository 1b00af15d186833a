"""PCIe-inclusive rate of the host-buffer entry (uint8 frame in host memory -> float32 frame in host memory),
for DESIGN.md §7; never the bench `value`."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")
s = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
s.prepare_style(pkg.synth_style(256, 256)); s.clean(); s.add(pkg.synth_frame(0, 512, 512)); s.compute()
frames = [V.reflect_pad(pkg.synth_frame(i, 512, 512), 640, 640) for i in range(8)]
for f in frames[:2]: s.transfer(f)
t = time.perf_counter(); n = 40
for i in range(n): s.transfer(frames[i % 8])
dt = time.perf_counter() - t
print("host entry, 1 frame per call: %.1f frames/s (%.3f ms/frame)" % (n / dt, 1e3 * dt / n))
t = time.perf_counter()
for i in range(5): s.transfer_batch(frames)
dt = time.perf_counter() - t
print("host entry, 8 frames per call: %.1f frames/s (%.3f ms/frame)" % (40 / dt, 1e3 * dt / 40))
