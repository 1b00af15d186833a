"""Device-resident rate against frames per launch (8 is the library's sub-batch) and streams in flight, 512 x 512."""
import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rerevst-code_amd"); V = importlib.import_module("rerevst-code_amd.video")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = V.padded_size(S)
m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
m.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7)); m.clean()
for i in (0, 8, 16): m.add(pkg.synth_frame(i, S, S, kind="noise"))
m.compute()
dev = torch.device("cuda", 0)
NB = 32 if S < 1024 else 8
frames = torch.from_numpy(np.stack([V.reflect_pad(pkg.synth_frame(i % 8, S, S, kind="noise"), P, P) for i in range(NB)])).to(dev)
out = torch.empty((2 * NB, P, P, 3), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
for slots in (1, 2):
    m.set_pipeline(slots)
    for B in ((1, 2, 3, 4, 6, 8) if S >= 1024 else (4, 8, 12, 16, 24, 32)):
        n = max(4, (256 if S < 1024 else 64) // B)
        for i in range(3): m.transfer_batch_device(frames.data_ptr(), B, P, P, out[(i & 1) * NB:].data_ptr())
        m.sync()
        t0 = time.perf_counter()
        for i in range(n): m.transfer_batch_device(frames.data_ptr(), B, P, P, out[(i & 1) * NB:].data_ptr())
        m.sync()
        dt = time.perf_counter() - t0
        print("size %d streams %d frames/launch %2d : %.1f frames/s (%.3f ms/frame)" % (S, slots, B, n * B / dt, 1e3 * dt / (n * B)), flush=True)
