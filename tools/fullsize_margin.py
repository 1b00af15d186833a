"""Pre-clamp error of the HIP path at the BASELINE configurations' FULL size and launch shape, per kernel selection, against
two evaluations of the oracle: its float32 convolutions on torch's conv2d ("torch") and every convolution accumulated in
float64 and rounded once ("torch64": the implementation's own error alone).  (run on the GPU box)
  headline: white-noise 512 x 512 frames padded to 640 x 640, sixteen per launch, the bench's B = 38 state
  config 5: 1152 x 1152, four styles, every style active; features from the batched caching entry
Kernel selections: F(2x2,3x3) everywhere (mode 0), the default rule (mode 1), and conv_f43_k restricted to layer subsets
(RRV_F43_LAYERS on a fresh handle, mode 2)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import state_bounds as T
import rerevst_oracle as O
pkg = importlib.import_module("rerevst-code_amd")
V = importlib.import_module("rerevst-code_amd.video")
W = pkg.synthetic_weights(0)
WHAT = sys.argv[1] if len(sys.argv) > 1 else "headline"
SUBSETS = (("encoder conv1_2, conv2_1, conv2_2", 0x007), ("encoder conv3_1 .. conv3_4", 0x078), ("all seven encoder layers", 0x07f),
           ("slice4/3/2.conv2", 0x380), ("conv1_2 only", 0x001), ("conv2_1 only", 0x002), ("conv2_2 only", 0x004), ("conv3_1 only", 0x008),
           ("conv3_2 only", 0x010), ("conv3_3 only", 0x020), ("conv3_4 only", 0x040))


def stats(tag, pre, refs, img=None, ref_img=None):
    cols = []
    for name, ref in refs:
        r = np.abs(pre.astype(np.float64) - ref) / (T.PRE_ATOL + T.PRE_RTOL * np.abs(ref))
        cols.append("vs %s: worst %.3f, 99.99th pct %.3f, mean %.4f, values over the bound %d" % (name, r.max(), np.percentile(r, 99.99), r.mean(), int((r > 1).sum())))
    extra = ""
    if img is not None:
        d = np.abs(img.astype(np.float64) - ref_img)
        extra = " | image vs torch64: max|d| %.4f, values beyond %.2f: %d" % (d.max(), T.IMG_ATOL, int((d > T.IMG_ATOL).sum()))
    print("%-58s %s%s" % (tag, " | ".join(cols), extra), flush=True)


def oracle_refs(run, with_numpy=False):
    """[("torch", pre), ("torch64", pre)] and the float32 oracle's own distance from the float64-accumulated one (with_numpy:
    also for its default backend, nine numpy GEMMs per convolution)."""
    refs = []
    for be in ("torch", "torch64") + (("numpy",) if with_numpy else ()):
        O.set_conv_backend(be)
        try:
            refs.append((be, run()))
        finally:
            O.set_conv_backend("numpy")
    b = refs[1][1]
    img_b = O.tensor_to_image(b[None])
    for name, a in [refs[0]] + refs[2:]:
        r = np.abs(a.astype(np.float64) - b) / (T.PRE_ATOL + T.PRE_RTOL * np.abs(b))
        d = np.abs(O.tensor_to_image(a[None]).astype(np.float64) - img_b)
        print("the float32 oracle (%s convolutions) against the float64-accumulated one: worst %.3f, 99.99th pct %.3f, mean %.4f, values over the bound %d | image max|d| %.4f, values beyond %.2f: %d; pre-clamp std %.3f"
              % (name, r.max(), np.percentile(r, 99.99), r.mean(), int((r > 1).sum()), d.max(), T.IMG_ATOL, int((d > T.IMG_ATOL).sum()), b.std()), flush=True)
    return refs[:2]


if WHAT == "headline":
    s = pkg.Stylization(W, cuda=True)
    s.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7)); s.clean()
    for i in V.sample_indices(300): s.add(pkg.synth_frame(i, 512, 512, kind="noise"))
    s.compute()
    state = s.get_state()
    frames = np.stack([V.reflect_pad(pkg.synth_frame(1 + i, 512, 512, kind="noise"), 640, 640) for i in range(16)])
    o = O.Stylization(W); o.set_state(state)
    FR = (0, 7)
    print("headline: 640 x 640 white-noise frames, sixteen per launch, B = 38 state; frames %s of the launch" % (FR,))
    refs = {k: oracle_refs(lambda: o.transfer(frames[k], return_preclamp=True)[0], with_numpy=(k == 0)) for k in FR}
    ref_img = {k: O.tensor_to_image(refs[k][1][1][None]) for k in FR}
    for mode, tag in ((0, "F(2x2,3x3) everywhere"), (1, "default rule (conv_f43_k on all ten packed layers)")):
        s.set_f43(mode)
        out = np.array(s.transfer_batch(frames))
        for k in FR:
            stats("frame %d: %s" % (k, tag), s.preclamp(640, 640, image=k), refs[k], out[k], ref_img[k])
    s.close()
    if "--direct" in sys.argv:      # encoder layers on the direct-form kernel (RRV_DIRECT_LAYERS: bit i = vgg conv i, 8 = conv4_1), the rest as the mode says
        for dname, dl in (("conv4_1", 0x100), ("conv3_1 .. conv4_1", 0x1f0), ("conv1_2 .. conv4_1 (the whole encoder)", 0x1fe)):
            for mode, tag in ((0, "F(2x2,3x3) elsewhere"), (1, "default rule elsewhere")):
                os.environ["RRV_DIRECT_LAYERS"] = hex(dl)
                s = pkg.Stylization(W, cuda=True); s.set_state(state); s.set_f43(mode)
                out = np.array(s.transfer_batch(frames))
                for k in FR:
                    stats("frame %d: direct form on %s, %s" % (k, dname, tag), s.preclamp(640, 640, image=k), refs[k], out[k], ref_img[k])
                s.close()
        os.environ.pop("RRV_DIRECT_LAYERS", None)
        sys.exit(0)
    for name, layers in SUBSETS:
        os.environ["RRV_F43_LAYERS"] = hex(layers)
        s = pkg.Stylization(W, cuda=True); s.set_state(state); s.set_f43(2)
        out = np.array(s.transfer_batch(frames))
        for k in FR:
            stats("frame %d: conv_f43_k on %s" % (k, name), s.preclamp(640, 640, image=k), refs[k], out[k], ref_img[k])
        s.close()
else:
    S = 4
    KIND = sys.argv[2] if len(sys.argv) > 2 else "noise"
    styles = [V.resize_bilinear(pkg.synth_style(96, 80, kind="smooth", seed=30 + k), (384, 384)) for k in range(S)]
    padded = [V.ReshapeTool().process(pkg.synth_frame(40 + i, 1024, 1024, kind=KIND)) for i in range(2)]
    wts = V.ramp_weights(150, 300, S, blend="all")

    def build(layers):
        if layers is None: os.environ.pop("RRV_F43_LAYERS", None)
        else: os.environ["RRV_F43_LAYERS"] = hex(layers)
        m = pkg.MultiStyleStylization(W, cuda=True, style_num=S)
        m.prepare_style(styles); m.set_f43(0)
        f22 = [m.generate_content_features(p) for p in padded]
        m.clean()
        for i in (0, 1): m.add_patch(f22[i])
        m.compute_norm()
        return m, f22
    s, f22 = build(None)
    o = O.MultiStylization(W, S)
    for k in range(S): o.per_style[k].set_state(s.get_state(k))
    print("config 5: 1152 x 1152 %s frames, four styles all active, one frame per launch" % KIND)
    refs = oracle_refs(lambda: o.transfer(o.generate_content_features(padded[0]), wts, return_preclamp=True)[0])
    ref_img = O.tensor_to_image(refs[1][1][None])

    def report(tag, m, feat):
        out = m.transfer(feat, wts)
        stats(tag, m.preclamp(1152, 1152), refs, out, ref_img)
    s.set_f43(0); report("features F(2x2), decoder F(2x2)", s, f22[0])
    s.set_f43(1); report("features F(2x2), decoder default rule (conv2 x 3: conv_f43_k)", s, f22[0])
    fb = s.generate_content_features_batch(padded)
    s.set_f43(0); report("features batched entry (default rule), decoder F(2x2)", s, fb[0])
    s.set_f43(1); report("features batched entry, decoder default rule", s, fb[0])
    s.close()
    for name, layers in SUBSETS[:4]:
        s, f22 = build(layers)
        s.set_f43(2)
        feat = s.generate_content_features_batch(padded)[0] if layers & 0x7f else f22[0]
        report("conv_f43_k on " + name, s, feat)
        s.close()
