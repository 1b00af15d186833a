"""Soak test for latent races in the hand-pipelined kernels: the same batch through the device entry many times, on
alternating streams, must give bit-identical output every time (and equal the single-stream result)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for S in (512, 200):
    P = V.padded_size(S)
    s = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
    s.prepare_style(pkg.synth_style(256, 256)); s.clean()
    for i in (0, 8): s.add(pkg.synth_frame(i, S, S))
    s.compute()
    B = 8
    frames = np.stack([V.reflect_pad(pkg.synth_frame(i, S, S, kind="noise"), P, P) for i in range(B)])
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.empty((4, B, P, P, 3), dtype=torch.float32, device="cuda")
    s.set_pipeline(1)
    s.transfer_batch_device(d_in.data_ptr(), B, P, P, d_out[0].data_ptr()); s.sync()
    ref = d_out[0].clone()
    s.set_pipeline(2)
    bad = 0
    for it in range(N):
        s.transfer_batch_device(d_in.data_ptr(), B, P, P, d_out[1 + it % 3].data_ptr())
        if it % 3 == 2 or it == N - 1:
            s.sync()
            for k in range(1, 2 + it % 3):          # the buffers written since the last check
                if not torch.equal(d_out[k], ref): bad += 1
            d_out[1:].zero_()
    print("size %d: %d batches of %d frames, mismatching buffers: %d" % (S, N, B, bad))
    s.close()
    assert bad == 0
print("soak ok")
