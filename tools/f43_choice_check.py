"""Is the kernel choice of the default mode (rrv_set_f43 1: use_f43 in rerevst_hip.hip) the faster one for the CUs a launch
may really use?  Times one device-resident rrv_transfer_batch_device per mode (0: F(2x2,3x3) everywhere, 1: the rule, 2:
conv_f43_k on every packed layer) and prints one JSON line.  (run on the GPU box; tests/test_gpu_f43.py runs it as a child
process with HSA_CU_MASK set, and with a grid share as the look-ahead tickets use)
    python tools/f43_choice_check.py B H W [grid_share]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import state_bounds as T
import torch
pkg = importlib.import_module("rerevst-code_amd")
B, H, W = (int(v) for v in sys.argv[1:4])
share = int(sys.argv[4]) if len(sys.argv) > 4 else 1
s = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
s.set_state(T.load_golden("global_a")["state"])
s.set_pipeline(1)
s.set_grid_share(share)
d_in = torch.from_numpy(np.stack([pkg.synth_frame(i, H, W, kind="noise") for i in range(B)])).to("cuda:0")
d_out = torch.zeros((B, H, W, 3), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
res = {}
for rep in range(2):                      # second pass: clocks and caches settled
    for mode in (0, 1, 2):
        s.set_f43(mode)
        ts = []
        for i in range(14):
            s.sync(); t0 = time.perf_counter()
            s.transfer_batch_device(d_in.data_ptr(), B, H, W, d_out.data_ptr())
            s.sync(); ts.append(time.perf_counter() - t0)
        res[mode] = float(np.median(ts[2:]))
s.close()
print(json.dumps({"B": B, "H": H, "W": W, "grid_share": share, "cu_mask": os.environ.get("HSA_CU_MASK"), "ms": {str(k): round(1e3 * v, 4) for k, v in res.items()}}))
