"""Which layer goes wrong in the rare multi-stream mismatch: one frame per stream in flight, after every round the outputs
are compared; on a mismatch the bad slot's activation tensors are compared with a single-stream run of the same frame."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rerevst-code_amd"); V = importlib.import_module("rerevst-code_amd.video")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = 100, 40
PH, PW = V.padded_size(H), V.padded_size(W)
NAMES = "c11 p1 c21 p2 c31 c32 c33 p3 c41 d f1 f2 f3 xs4 a4 o4 xs3 a3 o3 xs2 a2 o2 dpart".split()
m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
m.prepare_style(pkg.synth_style(96, 80, kind="smooth", seed=3)); m.clean()
for i in (0, 5): m.add(pkg.synth_frame(i, 72, 88, kind="smooth"))
m.compute()
NF = 12
frames = torch.from_numpy(np.stack([V.reflect_pad(pkg.synth_frame(100 + i, H, W, kind="noise"), PH, PW) for i in range(NF)])).cuda()
ref = torch.empty((NF, PH, PW, 3), dtype=torch.float32, device="cuda")
m.set_pipeline(1)
for k in range(NF): m.transfer_batch_device(frames[k].data_ptr(), 1, PH, PW, ref[k].data_ptr())
m.sync()
out = torch.zeros((slots, PH, PW, 3), dtype=torch.float32, device="cuda")
scratch = torch.zeros((PH, PW, 3), dtype=torch.float32, device="cuda")
m.set_pipeline(slots)
found = 0
t0 = time.time()
rng = np.random.default_rng(1)
for it in range(iters):
    ks = rng.integers(NF, size=slots)
    for s in range(slots): m.transfer_batch_device(frames[ks[s]].data_ptr(), 1, PH, PW, out[s].data_ptr())
    m.sync()
    for s in range(slots):
        if not torch.equal(out[s], ref[ks[s]]):
            found += 1
            d = (out[s] != ref[ks[s]]).cpu().numpy(); ys, xs, _ = np.nonzero(d)
            print("iteration %d slot %d frame %d: output differs in rows %d..%d cols %d..%d (%d values)" % (it, s, ks[s], ys.min(), ys.max(), xs.min(), xs.max(), d.sum()), flush=True)
            bad = [m.debug_tensor(s, i, PH, PW) for i in range(23)]
            m.set_pipeline(1)
            # the same frame alone, on the same slot's workspace?  set_pipeline(1) runs slot 0: compare tensor contents, not addresses
            m.transfer_batch_device(frames[ks[s]].data_ptr(), 1, PH, PW, scratch.data_ptr()); m.sync()
            good = [m.debug_tensor(0, i, PH, PW) for i in range(23)]
            m.set_pipeline(slots)
            for i, (b, g) in enumerate(zip(bad, good)):
                if b.shape != g.shape: print("   %s: shape differs" % NAMES[i]); continue
                nd = int((b != g).sum())
                if nd:
                    C = {0: 64, 1: 64, 2: 128, 3: 128, 4: 256, 5: 256, 6: 256, 7: 256, 8: 512, 9: 32, 10: 512, 11: 512, 12: 512, 13: 256, 14: 256, 15: 256, 16: 128, 17: 128, 18: 128, 19: 64, 20: 64, 21: 64}.get(i)
                    msg = ""
                    if C:
                        px = np.nonzero((b != g).reshape(-1, C).any(axis=1))[0]
                        msg = "; ring pixels %d..%d (%d pixels), channels %s" % (px.min(), px.max(), len(px), sorted(set(np.nonzero((b != g).reshape(-1, C).any(axis=0))[0] // 32 * 32))[:8])
                    print("   %-5s differs in %d of %d floats, max|d| %.3g%s" % (NAMES[i], nd, b.size, float(np.abs(b - g).max()), msg), flush=True)
            # detail of the split-K partial sums [H/8+2][W/8+2][256] (slab s = channels 32 s ..: K slice s of the LAST filter)
            bd, gd = bad[22].reshape(PH // 8 + 2, PW // 8 + 2, 256), good[22].reshape(PH // 8 + 2, PW // 8 + 2, 256)
            ys, xs, cs = np.nonzero(bd != gd)
            print("   dpart detail: K slices %s; rows %s; cols %s" % (sorted(set(cs // 32)), sorted(set(ys - 1)), sorted(set(xs - 1))))
            y0, x0, c0 = ys[0], xs[0], cs[0]
            print("   first bad element (row %d col %d channel %d): got %r expected %r; same pixel other slices got/expected: %s" % (y0 - 1, x0 - 1, c0, bd[y0, x0, c0], gd[y0, x0, c0],
                  [(float(bd[y0, x0, c0 % 32 + 32 * k]), float(gd[y0, x0, c0 % 32 + 32 * k])) for k in range(8)]))
            if found >= 3: break
    if found >= 3: break
print("layer hunt: %d rounds of %d streams in %.1f s, %d mismatches" % (it + 1, slots, time.time() - t0, found))
