"""Prints the measured error of the HIP path against the reference goldens next to the stated tolerances
(tests/conftest.py), for DESIGN.md §6.  Run on the GPU box: python tools/parity_margin.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest as T
import rerevst_oracle as O
pkg = importlib.import_module("rerevst-code_amd")
hip = pkg.Stylization(pkg.synthetic_weights(0), cuda=True)
for case in ("global_a", "global_b"):
    g = T.load_golden(case)
    style, frames, ids, tid = T.golden_inputs(pkg, g)
    hip.prepare_style(style); hip.clean()
    for i in ids: hip.add(frames[i])
    hip.compute()
    st = hip.get_state()
    sworst, swhere = T.state_worst(st, g["state"])
    H, W = frames[0].shape[:2]
    PH, PW = O.padded_size(H), O.padded_size(W)
    out = hip.transfer(O.reflect_pad(frames[tid], PH, PW)); pre = hip.preclamp(PH, PW)
    if "pre" in g.files: rp, ro, p, o = g["pre"], g["out"], pre, out
    else: rp, ro, p, o = g["pre_crop"], g["out_crop"], pre[64:64 + H, 64:64 + W], out[64:64 + H, 64:64 + W]
    ep = np.abs(p - rp); bp = T.PRE_ATOL + T.PRE_RTOL * np.abs(rp)
    print("%s: state worst entry at %.0f%% of its bound (%s) | pre-clamp max|d| %.2e (worst err/bound %.3f, ref std %.3f) | image max|d| %.4f grey levels (bound %.2f)"
          % (case, 100 * sworst, swhere, ep.max(), (ep / bp).max(), rp.std(), np.abs(o - ro).max(), T.IMG_ATOL))
