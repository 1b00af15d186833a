"""Focused stress of the look-ahead tickets (the soak's rare 1-ulp mismatch): tickets in flight, collected by another
entry or by result(), several geometries; on a mismatch print where the differences sit."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
m.prepare_style(pkg.synth_style(96, 80, kind="smooth", seed=3)); m.clean()
for i in (0, 5): m.add(pkg.synth_frame(i, 72, 88, kind="smooth"))
m.compute()
fs = [(40, 40), (100, 40), (40, 100)]
pools, refs = [], []
for h, w in fs:
    P = (V.padded_size(h), V.padded_size(w))
    pool = np.stack([V.reflect_pad(pkg.synth_frame(100 + i, h, w, kind="noise"), *P) for i in range(12)])
    pools.append(pool); refs.append(np.stack([m.transfer(f) for f in pool]))
m.set_host_io(mode)
rng = np.random.default_rng(5)
bad = 0
t0 = time.time()
for it in range(iters):
    g = int(rng.integers(3))
    ks = rng.integers(12, size=4)
    tickets = [m.transfer_async(pools[g][k]) for k in ks]
    what = int(rng.integers(3))
    if what == 0:
        g2 = int(rng.integers(3)); idx = rng.integers(12, size=int(rng.integers(1, 24)))
        out = m.transfer_batch(pools[g2][idx])
        if not np.array_equal(out, refs[g2][idx]): print("iteration %d: the batch between differs" % it); bad += 1
    elif what == 1:
        m.transfer(pools[g][0])
    for t, k in zip(tickets, ks):
        got = m.result(t)
        if not np.array_equal(got, refs[g][k]):
            d = got != refs[g][k]
            ys, xs, cs = np.nonzero(d)
            print("iteration %d (geometry %s, between: %d): ticket of frame %d differs in %d of %d values; rows %d..%d cols %d..%d; per channel %s; max|d| %.3g; 16x16 tiles touched %d of %d"
                  % (it, refs[g][k].shape, what, k, d.sum(), d.size, ys.min(), ys.max(), xs.min(), xs.max(), [int(d[..., c].sum()) for c in range(3)],
                     np.abs(got - refs[g][k]).max(), len(set(zip(ys // 16, xs // 16))), (d.shape[0] // 16) * (d.shape[1] // 16)), flush=True)
            bad += 1
print("ticket stress (host_io %d): %d iterations x 4 tickets in %.1f s, %d mismatches" % (mode, iters, time.time() - t0, bad))
