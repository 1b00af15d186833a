"""Determinism soak of the two single-stream paths: compute() (resident and streaming) repeated with the same sampled
frames, frame-mode transfer() repeated over a few frames — every state / image bit-identical to the first run."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rerevst-code_amd"); V = importlib.import_module("rerevst-code_amd.video")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
style = pkg.synth_style(200, 152, kind="smooth", seed=5)
sampled = [pkg.synth_frame(i, 136, 200, kind="noise") for i in range(11)]
def state(cap=None):
    if cap: m.set_workspace_cap(cap)
    m.prepare_style(style); m.clean()
    for f in sampled: m.add(f)
    m.compute()
    return m.get_state()
ref = state()
bad = sum(int(not np.array_equal(state(), ref)) for _ in range(n))
info = m.last_compute_info()
ref_s = state(64 << 20)
bad_s = sum(int(not np.array_equal(state(64 << 20), ref_s)) for _ in range(n // 4))
print("compute(): %d resident runs (%s), %d differ; %d streaming runs (%s), %d differ" % (n, info, bad, n // 4, m.last_compute_info(), bad_s))
m.close()
fm = pkg.Stylization(pkg.synthetic_weights(0), cuda=True, use_Global=False)
fm.prepare_style(style)
frames = [V.reflect_pad(pkg.synth_frame(50 + i, 100, 72, kind="noise"), 256, 256) for i in range(4)]
refs = [fm.transfer(f) for f in frames]
bad_f = 0
for it in range(n * 5):
    k = it % 4
    bad_f += int(not np.array_equal(fm.transfer(frames[k]), refs[k]))
print("frame mode: %d transfers, %d differ" % (n * 5, bad_f))
assert bad == 0 and bad_s == 0 and bad_f == 0
