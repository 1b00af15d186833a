"""How far do equally valid float32 evaluations of the SAME preparation pass land from the float64 state on the
ill-conditioned weight set `dec4` (every decoder weight x4; tests/golden/global_a_dec4)?  CPU only (the oracle with
patched convolutions).  VERDICT r3 asked to find the kernel that moves the HIP state to 59x the bound where the
reference's own float32 run sits at 30x: there is none — the distance is rounding noise amplified by the dynamic filters.

    python tools/dec4_conditioning.py > profiles/r04_dec4_conditioning.txt
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import state_bounds as T
import rerevst_oracle as O
pkg = importlib.import_module("rerevst-code_amd")
F32, F64 = np.float32, np.float64
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], F32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], F32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], F32)
direct = O.conv3x3
MODE = {"f23": lambda ci, co: False, "f64": lambda ci, co: False}


def conv_f64acc(x, w, b):        # float32 tensors, every dot product accumulated in float64 and rounded once
    B, H, W, Cin = x.shape
    xp = np.zeros((B, H + 2, W + 2, Cin), F64); xp[:, 1:H + 1, 1:W + 1] = x
    acc = np.zeros((B * H * W, w.shape[0]), F64)
    for ky in range(3):
        for kx in range(3):
            acc += np.ascontiguousarray(xp[:, ky:ky + H, kx:kx + W, :]).reshape(-1, Cin) @ np.ascontiguousarray(w[:, :, ky, kx].T.astype(F64))
    out = acc.reshape(B, H, W, -1)
    return (out if b is None else out + b.astype(F64)).astype(F32)


def conv_f23(x, w, b):           # float32 Winograd F(2x2,3x3), the arithmetic of conv_wino_split_k
    Bn, H, W, Cin = x.shape; Cout = w.shape[0]
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = np.zeros((Bn, th * 2 + 2, tw * 2 + 2, Cin), F32); xp[:, 1:H + 1, 1:W + 1] = x
    U = np.einsum('ia,ocab,jb->ijco', G, w.astype(F32), G, optimize=True).astype(F32).reshape(16, Cin, Cout)
    out = np.zeros((Bn, th * 2, tw * 2, Cout), F32)
    for bi in range(Bn):
        s = xp[bi].strides
        pt = np.lib.stride_tricks.as_strided(xp[bi], shape=(th, tw, 4, 4, Cin), strides=(2 * s[0], 2 * s[1], s[0], s[1], s[2]))
        V = np.einsum('ia,tuabc,jb->ijtuc', BT, pt, BT, optimize=True).astype(F32).reshape(16, th * tw, Cin)
        M = np.matmul(V, U).astype(F32).reshape(4, 4, th, tw, Cout)
        out[bi] = np.einsum('ia,abtuc,jb->tiujc', AT, M, AT, optimize=True).astype(F32).reshape(th * 2, tw * 2, Cout)
    out = out[:, :H, :W]
    return (out if b is None else out + b.astype(F32)).astype(F32)


def conv(x, w, b=None):
    ci, co = x.shape[-1], w.shape[0]
    if MODE["f64"](ci, co): return conv_f64acc(x, w, b)
    if MODE["f23"](ci, co): return conv_f23(x, w, b)
    return direct(x, w, b)


O.conv3x3 = conv
g = T.load_golden("global_a_dec4")
style, frames, ids, tid = T.golden_inputs(pkg, g)
w = pkg.weight_variant("dec4")
never = lambda ci, co: False


def run(name, f23=never, f64=never, backend="numpy"):
    O.set_conv_backend(backend)
    MODE["f23"], MODE["f64"] = f23, f64
    o = O.Stylization(w); o.prepare_style(style); o.clean()
    for i in ids: o.add(frames[i])
    o.compute()
    st = o.get_state()
    O.set_conv_backend("numpy")
    rows = sorted(T.state_fields(st, g["state_fp64"]), key=lambda r: -r[1])
    print("%-78s %6.1fx   worst fields: %s" % (name, rows[0][1], ", ".join("%s %.1f" % (r[0], r[1]) for r in rows[:3])), flush=True)


print("# tools/dec4_conditioning.py — distance of float32 evaluations of prepare_style/add/compute to the FLOAT64 state of the reference")
print("# (tests/golden/global_a_dec4: state_fp64), worst entry / its bound (tests/state_bounds.py); 1.0 = the stated tolerance")
rows = sorted(T.state_fields(g["state"], g["state_fp64"]), key=lambda r: -r[1])
print("%-78s %6.1fx   worst fields: %s" % ("the reference itself, float32 (torch CPU, 8 threads; the committed golden)", rows[0][1],
                                            ", ".join("%s %.1f" % (r[0], r[1]) for r in rows[:4])))
print("#   ... every other field of the reference's own float32 state is within %.2f of its bound" % rows[3][1] if rows[3][1] < 1 else "")
run("oracle, direct form, nine numpy GEMMs per convolution")
run("oracle, direct form, torch conv2d (the reference's primitive), 1 process", backend="torch")
run("oracle, F(2x2,3x3) float32 on every 3x3 layer with Cin, Cout >= 32", f23=lambda ci, co: ci >= 32 and co >= 32)
run("oracle, F(2x2,3x3) on the layers with Cin, Cout >= 64 only (encoder, conv1, conv2)", f23=lambda ci, co: ci >= 64 and co >= 64)
run("oracle, F(2x2,3x3) on the folded KernelFilter convs only (512->32, 32->512)", f23=lambda ci, co: (ci, co) in ((512, 32), (32, 512)))
run("oracle, KernelFilter + FilterPredictor convs accumulated in float64, rest direct float32", f64=lambda ci, co: (ci, co) in ((512, 32), (32, 512)))
run("oracle, EVERY 3x3 convolution accumulated in float64, tensors rounded to float32", f64=lambda ci, co: True)
print("# HIP path (profiles/r03_parity_margin.txt, r04_parity_margin.txt): 59.3x, worst field Filter3.F1.filter")
print("# Reading: storing the tensors in float32 alone (last line) costs as much as the reference's whole float32 run; every float32")
print("# evaluation lands between 0.8x and 4.5x of the reference's own miss depending on where its rounding errors fall, and making")
print("# the dynamic-filter chain exact does not help, because the noise enters through the float32 encoder features.  No kernel to fix.")
