import importlib, sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import state_bounds as T, rerevst_oracle as O
pkg = importlib.import_module("rerevst-code_amd")
g = T.load_golden("real_default")
frame = T.decode_png(g["frame%d_png" % int(g["transfer_id"])])
padded = O.reflect_pad(frame, 576, 1152)
def run(mode, B, layers=None):
    if layers is not None: os.environ["RRV_F43_LAYERS"] = hex(layers)
    s = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    s.set_state(g["state"]); s.set_f43(mode)
    out = np.array(s.transfer_batch([padded] * B)[0]) if B > 1 else np.array(s.transfer(padded))
    s.close()
    return out
ref = {}
for layers in (0x3ff, 0x3c5):
    a = run(2, 1, layers); b = run(2, 4, layers); c = run(1, 4, layers); d = run(1, 2, layers); e = run(1, 1, layers); z = run(0, 1, layers)
    print("layers %#x: mode2 B1 vs mode2 B4: %s | mode1 B4 vs mode2 B4: %s | mode1 B2 vs mode2: %s | mode1 B1 vs mode0: %s, vs mode2: %s" % (
        layers, np.array_equal(a, b), np.array_equal(c, b), np.array_equal(d, b), np.array_equal(e, z), np.array_equal(e, a)), flush=True)
    print("   max|mode2 B1 - mode2 B4| = %.2e, max|mode1 B4 - mode2 B4| = %.2e" % (np.abs(a - b).max(), np.abs(c - b).max()))
for bit in range(10):
    a = run(2, 1, 1 << bit); b = run(2, 4, 1 << bit); c = run(1, 4, 1 << bit)
    print("only layer bit %d: mode2 B1 == mode2 B4: %s (max %.2e) | mode1 B4 == mode2 B4: %s" % (bit, np.array_equal(a, b), np.abs(a - b).max(), np.array_equal(c, b)), flush=True)
