"""Device-entry stress: one frame per launch on N streams (rrv_set_pipeline) with 1/share of the CUs per grid
(rrv_set_grid_share), outputs compared bit for bit with the single-stream result.  Small frames = many short kernels
= the most kernel-to-kernel overlap between the streams.  (Round 3: this found one frame in ~10^4 slightly wrong with
two or more streams — a missing barrier between the first item's prologue and its first LDS-DMA in the transform-domain
kernels, present since round 1; tools/layer_hunt.py located it.)
    python tools/device_stream_stress.py [iterations] [slots] [share] [H] [W]"""
import importlib, os, sys, time
import numpy as np


def run(iters=2000, slots=4, share=1, H=100, W=40, verbose=True):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    pkg = importlib.import_module("rerevst-code_amd"); V = importlib.import_module("rerevst-code_amd.video")
    m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    m.prepare_style(pkg.synth_style(96, 80, kind="smooth", seed=3)); m.clean()
    for i in (0, 5): m.add(pkg.synth_frame(i, 72, 88, kind="smooth"))
    m.compute()
    PH, PW = V.padded_size(H), V.padded_size(W)
    NF = 12
    frames = torch.from_numpy(np.stack([V.reflect_pad(pkg.synth_frame(100 + i, H, W, kind="noise"), PH, PW) for i in range(NF)])).cuda()
    ref = torch.empty((NF, PH, PW, 3), dtype=torch.float32, device="cuda")
    m.set_pipeline(1); m.set_grid_share(1)
    for k in range(NF): m.transfer_batch_device(frames[k].data_ptr(), 1, PH, PW, ref[k].data_ptr())
    m.sync()
    out = torch.zeros((NF, PH, PW, 3), dtype=torch.float32, device="cuda")
    m.set_pipeline(slots); m.set_grid_share(share)
    bad = 0
    t0 = time.time()
    for it in range(iters):
        for k in range(NF): m.transfer_batch_device(frames[k].data_ptr(), 1, PH, PW, out[k].data_ptr())
        m.sync()
        if not torch.equal(out, ref):
            for k in range(NF):
                if not torch.equal(out[k], ref[k]):
                    d = (out[k] != ref[k]).cpu().numpy(); ys, xs, cs = np.nonzero(d)
                    if verbose: print("iteration %d frame %d (slot %d): %d of %d values differ; rows %d..%d cols %d..%d; max|d| %.3g" % (it, k, k % slots, d.sum(), d.size, ys.min(), ys.max(), xs.min(), xs.max(), float((out[k] - ref[k]).abs().max())), flush=True)
                    bad += 1
        out.zero_()
    m.set_pipeline(2); m.set_grid_share(1)
    m.close()
    if verbose: print("device stress %dx%d padded %dx%d: %d x %d frames on %d streams, grid share %d, in %.1f s: %d mismatching frames" % (H, W, PH, PW, iters, NF, slots, share, time.time() - t0, bad))
    return bad


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    run(*a)
