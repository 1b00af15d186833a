"""Randomised soak of the host-side pipelines (staging sets, copy streams, events, look-ahead tickets, zero-copy modes,
workspace plans): a random sequence of entries / buffer kinds / frame sizes / batch sizes, every output compared BIT FOR
BIT with the plain one-frame transfer() of the same frame.      python tools/soak_host_entries.py [iterations] [seed]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")


def run(iters=300, seed=0, verbose=True, big=False):
    rng = np.random.default_rng(seed)
    m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    m.prepare_style(pkg.synth_style(96, 80, kind="smooth", seed=3)); m.clean()
    for i in (0, 5): m.add(pkg.synth_frame(i, 72, 88, kind="smooth"))
    m.compute()
    fsizes = [(40, 40), (100, 40), (40, 100)]              # three padded geometries: the two resident plans per slot get evicted and rebuilt
    if big: fsizes.append((256, 300))                       # padded 384 x 448: 19 frames per sub-batch, so batches of up to 90 run 5 sub-batches through the 4 staging sets
    sizes = [(V.padded_size(h), V.padded_size(w)) for h, w in fsizes]
    raw_sizes = [(40, 56), (67, 33)]                        # unpadded frames for the pad / crop entry
    pool, ref = {}, {}
    for (h, w), s in zip(fsizes, sizes):
        pool[s] = np.stack([V.reflect_pad(pkg.synth_frame(100 + i, h, w, kind="noise"), *s) for i in range(12)])
        ref[s] = np.stack([m.transfer(f) for f in pool[s]])
    rpool, rref = {}, {}
    for s in raw_sizes:
        rpool[s] = np.stack([pkg.synth_frame(200 + i, *s, kind="noise") for i in range(6)])
        rref[s] = m.transfer_frames(list(rpool[s]))
    pins = {}
    def pinned_copy(a):
        p = pkg.pinned_empty(a.shape, a.dtype); p[...] = a; return p
    open_tickets = []          # (ticket, expected)
    counts = {}
    t0 = time.time()
    hist = []
    mode, depth = 0, 2
    def check(got, exp, what):
        if not np.array_equal(got, exp):
            bad = np.argwhere(np.asarray(got) != np.asarray(exp))
            raise AssertionError("iteration %d: %s differs (host_io %d, pipeline %d): %d of %d values, first at %s, got %r expected %r; last operations: %s"
                                 % (it, what, mode, depth, len(bad), np.asarray(exp).size, tuple(bad[0]), np.asarray(got)[tuple(bad[0])], np.asarray(exp)[tuple(bad[0])], hist[-12:]))
    for it in range(iters):
        op = rng.choice(["transfer", "batch", "batch_pinned", "frames", "async", "collect", "host_io", "pipeline"], p=[.12, .2, .15, .12, .2, .1, .06, .05])
        op = str(op); counts[op] = counts.get(op, 0) + 1; hist.append(op)
        s = sizes[rng.integers(len(sizes))]
        if op == "transfer":
            k = int(rng.integers(12))
            check(m.transfer(pool[s][k]), ref[s][k], 'transfer')
        elif op in ("batch", "batch_pinned"):
            n = int(rng.integers(1, 91 if (big and s == sizes[-1]) else 24)); idx = rng.integers(12, size=n)
            frames = pool[s][idx]
            if op == "batch_pinned":
                out = pkg.pinned_empty(ref[s][idx].shape, np.float32); out[...] = -1
                m.transfer_batch(pinned_copy(frames), out=out)
            else:
                out = m.transfer_batch(frames)
            hist[-1] += '(n=%d)' % n; check(out, ref[s][idx], op)
        elif op == "frames":
            rs = raw_sizes[rng.integers(len(raw_sizes))]
            n = int(rng.integers(1, 10)); idx = rng.integers(6, size=n)
            got = m.transfer_frames(list(rpool[rs][idx]))
            hist[-1] += '(n=%d)' % n
            for j, k in enumerate(idx): check(got[j], rref[rs][k], 'transfer_frames[%d of %d]' % (j, n))
        elif op == "async":
            k = int(rng.integers(12))
            if len(open_tickets) >= 4:
                t, exp = open_tickets.pop(0); check(m.result(t), exp, 'oldest ticket')
            open_tickets.append((m.transfer_async(pool[s][k]), ref[s][k]))
        elif op == "collect" and open_tickets:
            j = int(rng.integers(len(open_tickets)))
            t, exp = open_tickets.pop(j); check(m.result(t), exp, 'ticket %d of %d open' % (j, len(open_tickets) + 1))
        elif op == "host_io":
            mode = int(rng.integers(4)); m.set_host_io(mode); open_tickets.clear(); hist[-1] += '(%d)' % mode      # set_host_io retires the open tickets
        elif op == "pipeline":
            depth = int(rng.integers(1, 3)); m.set_pipeline(depth); hist[-1] += '(%d)' % depth
    it = iters
    for t, exp in open_tickets: check(m.result(t), exp, 'ticket at the end')
    m.set_host_io(0)
    m.close()
    if verbose: print("soak of the host entries: %d random operations in %.1f s, every output bit-identical to transfer(); mix %s" % (iters, time.time() - t0, dict(sorted(counts.items()))))


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0, big=len(sys.argv) > 3)
