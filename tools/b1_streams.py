import importlib, sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module("rerevst-code_amd"); V = importlib.import_module("rerevst-code_amd.video")
S=512; P=V.padded_size(S)
m = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
m.prepare_style(pkg.synth_style(512,512,kind="noise",seed=7)); m.clean()
for i in (0,8,16): m.add(pkg.synth_frame(i,S,S,kind="noise"))
m.compute()
dev=torch.device("cuda",0)
frames=torch.from_numpy(np.stack([V.reflect_pad(pkg.synth_frame(i,S,S,kind="noise"),P,P) for i in range(8)])).to(dev)
out=torch.empty((8,P,P,3),dtype=torch.float32,device=dev)
torch.cuda.synchronize()
for slots, share in ((1,1),(2,1),(2,2),(3,2),(4,2),(3,3),(4,4)):
    m.set_pipeline(slots); m.set_grid_share(share)
    for B in (1,2,8):
        n=64//B
        for i in range(4): m.transfer_batch_device(frames[(i*B)%8:].data_ptr(), B, P, P, out[(i*B)%8:].data_ptr())
        m.sync()
        t0=time.perf_counter()
        for i in range(n): m.transfer_batch_device(frames[(i*B)%8:].data_ptr(), B, P, P, out[(i*B)%8:].data_ptr())
        m.sync()
        dt=time.perf_counter()-t0
        print("slots %d share %d  B %d : %.1f frames/s (%.3f ms/frame)"%(slots,share,B,n*B/dt,1e3*dt/(n*B)), flush=True)
