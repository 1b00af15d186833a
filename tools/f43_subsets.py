import sys, runpy, numpy as np
sys.argv = ["f43_numerics.py"]          # import the tool's definitions without running a case
ns = runpy.run_path(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "f43_numerics.py"))
O, C, pkg, MODE, conv_f43, orig, pre_ratio = ns["O"], ns["C"], ns["pkg"], ns["MODE"], ns["conv_f43"], ns["orig"], ns["pre_ratio"]
state = {"i": 0, "use": set()}
def conv_sel(x, w, b=None):
    Cin, Cout = x.shape[3], w.shape[0]
    if Cin < 64 or Cout < 64: return orig(x, w, b)
    i = state["i"]; state["i"] += 1
    if i in state["use"]:
        MODE["layers"] = "all"; return conv_f43(x, w, b)
    return orig(x, w, b)
O.conv3x3 = conv_sel
O.set_conv_backend("torch")
orig_t = O.conv3x3 if False else None
w = pkg.synthetic_weights(0)
g = C.load_golden("real_default")
o = O.Stylization(w); o.set_state(g["state"])
padded = O.reflect_pad(C.decode_png(g["frame12_png"]), 576, 1152)
names = ["conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv3_4", "conv4_1", "s4.conv1(ups)", "s4.conv2", "s3.conv1(ups)", "s3.conv2", "s2.conv1(ups)", "s2.conv2"]
subsets = {
    "none": set(),
    "every same-resolution layer (no upsample-fused conv1)": {0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13},
    "decoder conv2 x3 (<54>)": {9, 11, 13},
    "decoder conv2 at 64 and 128 channels": {11, 13},
    "encoder 64->64, 128->128 + decoder conv2 x3": {0, 2, 9, 11, 13},
    "whole encoder": {0, 1, 2, 3, 4, 5, 6, 7},
    # round 4: candidate sets for the shipped kernel choice
    "every same-resolution layer except conv4_1 (256->512)": {0, 1, 2, 3, 4, 5, 6, 9, 11, 13},
    "every same-resolution layer except conv4_1 and the 64-channel ones": {1, 2, 3, 4, 5, 6, 9, 11},
    "128/256-channel layers of the encoder (conv2_1..conv3_4) + s4/s3 conv2": {1, 2, 3, 4, 5, 6, 9, 11},
}
import os
if os.environ.get("F43_ONLY"):
    subsets = {k: v for k, v in subsets.items() if os.environ["F43_ONLY"] in k}
for nm, use in subsets.items():
    state["i"] = 0; state["use"] = use
    pre = o.transfer(padded, return_preclamp=True)[0][64:500, 64:1088]
    assert state["i"] == 14, state["i"]
    print("real_default, F(4x4,3x3) on %-58s pre worst ratio %.3f max|d| %.2e" % (nm + ":", *pre_ratio(pre[::4, ::4], g["pre_grid"])), flush=True)
