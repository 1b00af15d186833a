"""Randomised soak of the multi-style entries (state sets per slot / per image, blends and folds on two streams, the
feature cache with and without a cap): transfer_many over random features / weights / group sizes against the
per-feature transfer(), bit for bit — in ONE kernel family (rrv_set_f43 0 or 2: the default mode chooses by the frames per
launch, so a grouped frame and a single one may differ in the low-order bits).      python tools/soak_multistyle.py [iterations] [seed] [mode]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")


def run(iters=200, seed=0, verbose=True, mode=0):
    rng = np.random.default_rng(seed)
    S = 3
    m = pkg.MultiStyleStylization(pkg.synthetic_weights(0), cuda=True, style_num=S)
    m.prepare_style([pkg.synth_style(64, 64, kind="smooth", seed=7 + k) for k in range(S)])
    frames = [V.reflect_pad(pkg.synth_frame(300 + i, 40, 56, kind="noise"), 192, 192) for i in range(10)]
    feats = [m.generate_content_features(f) for f in frames]
    m.clean()
    for i in (0, 4, 9): m.add_patch(feats[i])
    m.compute_norm()
    m.set_f43(mode)
    wpool = [list(rng.dirichlet(np.ones(S))) for _ in range(6)] + [[1.0, 0.0, 0.0], [0.0, 0.5, 0.5]]
    ref = {}
    def reference(fi, wi):
        if (fi, wi) not in ref: ref[(fi, wi)] = m.transfer(feats[fi], wpool[wi]).copy()
        return ref[(fi, wi)]
    t0 = time.time()
    n_frames = 0
    for it in range(iters):
        op = rng.choice(["many", "many", "many", "group", "pipeline", "single", "frame"])
        if op == "group":
            m.set_multistyle_group(int(rng.choice([0, 1, 2, 4, 7, 16])))
        elif op == "pipeline":
            m.set_pipeline(int(rng.integers(1, 3)))
        elif op == "single":
            fi, wi = int(rng.integers(10)), int(rng.integers(len(wpool)))
            want = reference(fi, wi)
            if not np.array_equal(m.transfer(feats[fi], wpool[wi]), want): raise AssertionError("iteration %d: single transfer differs" % it)
        elif op == "frame":     # encoder + blended decoder from the frame: same decoder arithmetic, the encoder normalises in its epilogue (mode 2: the cached features came from the default choice at one frame per launch)
            fi, wi = int(rng.integers(10)), int(rng.integers(len(wpool)))
            got = pkg.Stylization.transfer(m, frames[fi], style_weight=wpool[wi])
            if np.abs(got - reference(fi, wi)).max() > (1e-3 if mode == 0 else 0.05): raise AssertionError("iteration %d: blended frame transfer differs by %g" % (it, np.abs(got - reference(fi, wi)).max()))
        else:
            n = int(rng.integers(1, 14))
            fis, wis = rng.integers(10, size=n), rng.integers(len(wpool), size=n)
            want = [reference(int(a), int(b)) for a, b in zip(fis, wis)]
            got = m.transfer_many([feats[int(a)] for a in fis], [wpool[int(b)] for b in wis])
            n_frames += n
            for k in range(n):
                if not np.array_equal(got[k], want[k]):
                    d = got[k] != want[k]
                    raise AssertionError("iteration %d: transfer_many frame %d of %d differs in %d values, max|d| %g" % (it, k, n, d.sum(), np.abs(got[k] - want[k]).max()))
    m.set_multistyle_group(0); m.set_pipeline(2)
    m.release_features(); m.close()
    if verbose: print("soak of the multi-style entries (rrv_set_f43 %d): %d random operations (%d batched frames) in %.1f s, every output bit-identical to the per-feature transfer()" % (mode, iters, n_frames, time.time() - t0))


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0, mode=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
