"""Stated parity tolerances and the comparison helpers built on them.  Plain module (no pytest): imported by
tests/conftest.py, tools/parity_margin.py and __graft_entry__.smoke()."""
import io
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Stated fp32 tolerances (reference fp32-vs-fp64 jitter is 4.4e-6 on a
# pre-clamp output of std 0.165):
PRE_ATOL, PRE_RTOL = 1e-4, 1e-3      # pre-clamp network output (normalised image units)
IMG_ATOL = 0.05                      # final image, grey levels of 255
# Saved-state blob (rel 1e-4).  Every entry |d| <= atol + STATE_RTOL*|ref| with a per-FIELD absolute floor
# that says how well the quantity is defined in fp32 at all (same noise in the reference's own arithmetic, whose
# fp32-vs-fp64 jitter on activations is ~4e-6):
#   mean          2e-5   a sum with cancellation: ~1e-6 absolute error whatever its value
#   std = 1/rstd  2e-7   activations carry ~1e-7 absolute rounding error, so a nearly dead channel (std 2.5e-4 at
#                        relu4_1 with the seeded weights, rstd ~ 4000) has no better-defined spread than that; the
#                        blob stores rstd, which is compared through its reciprocal
#   min(x),max(x) 1e-4   the clamp limits lo/hi = (extremum - mean)*rstd are compared as the RAW extrema they encode
#                        (v/rstd + mean, each side with its own mean/rstd); an extremum is ONE pixel's value, and
#                        everything behind Decoder.norm[0] inherits the noise that the near-dead channels' rstd
#                        amplifies (1e-7 * 4000 = 4e-4 in normalised units, times the next conv's weights)
#   filters, style moments 2e-5
STATE_RTOL = 1e-4
STATE_ATOL = {"mean": 2e-5, "std": 2e-7, "ext": 1e-4, "other": 2e-5}



def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def decode_png(buf):
    """uint8 PNG stream stored in a fixture -> uint8 BGR HWC image (PNG is lossless: the decoded pixels are exactly the
    ones the reference was run on)."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(np.asarray(buf, np.uint8).tobytes())).convert("RGB"))[..., ::-1])


def golden_inputs(pkg, g):
    """Re-create the seeded inputs a golden case was generated from."""
    sh, fh = tuple(int(v) for v in g["style_hw"]), tuple(int(v) for v in g["frame_hw"])
    style = pkg.synth_style(*sh, kind="smooth", seed=7)
    frames = [pkg.synth_frame(i, *fh, kind="smooth") for i in range(int(g["n_frames"]))]
    return style, frames, [int(i) for i in g["sample_ids"]], int(g["transfer_id"])


NORM_CH = [512, 512, 256, 128, 64, 256, 256, 128, 128, 64, 64]      # blob layout: DESIGN.md §3
NORM_NAMES = ["dec.norm0", "dec.norm1", "dec.norm2", "dec.norm3", "dec.norm4", "slice4.norm1", "slice4.norm2",
              "slice3.norm1", "slice3.norm2", "slice2.norm1", "slice2.norm2"]


def state_fields(got, ref):
    """Per FIELD of the saved-state blob (11 norm layers x {mean, std, min(x), max(x)}, 6 filters, 4 x 2 style moments):
    [(name, worst |d| / bound, index of that entry, got, ref)] (bounds: see STATE_ATOL above)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    rows = []
    def add(g, r, name, kind):
        ratio = np.abs(g - r) / (STATE_ATOL[kind] + STATE_RTOL * np.abs(r))
        k = int(np.argmax(ratio))
        rows.append((name, float(ratio[k]), k, float(g[k]), float(r[k]), STATE_ATOL[kind]))
    o = 0
    for name, C in zip(NORM_NAMES, NORM_CH):
        gm, gr, gl, gh = (got[o + i * C:o + (i + 1) * C] for i in range(4))
        rm, rr, rl, rh = (ref[o + i * C:o + (i + 1) * C] for i in range(4))
        add(gm, rm, name + ".mean", "mean")
        add(1.0 / gr, 1.0 / rr, name + ".std", "std")
        add(gl / gr + gm, rl / rr + rm, name + ".min(x)", "ext")
        add(gh / gr + gm, rh / rr + rm, name + ".max(x)", "ext")
        o += 4 * C
    for f in range(6):
        add(got[o:o + 1024], ref[o:o + 1024], "Filter%d.F%d.filter" % (f // 2 + 1, f % 2 + 1), "other")
        o += 1024
    for k, C in enumerate((64, 128, 256, 512)):
        add(got[o:o + C], ref[o:o + C], "style.relu%d_1.mean" % (k + 1), "other")
        add(got[o + C:o + 2 * C], ref[o + C:o + 2 * C], "style.relu%d_1.std" % (k + 1), "other")
        o += 2 * C
    return rows


def state_worst(got, ref):
    """Largest |d| / bound over the saved-state blob and where it sits (bounds: see STATE_ATOL above)."""
    name, ratio, k, g, r, atol = max(state_fields(got, ref), key=lambda row: row[1])
    return ratio, "%s[%d]: got %.6e ref %.6e (atol %.0e)" % (name, k, g, r, atol)


def assert_state_close(got, ref, what="state"):
    worst, where = state_worst(got, ref)
    assert worst <= 1.0, "%s: worst entry at %.0f%% of its bound (atol + %.0e*|ref|): %s" % (what, 100 * worst, STATE_RTOL, where)


def assert_pre_close(got, ref, what="pre-clamp"):
    err = np.abs(got - ref)
    bound = PRE_ATOL + PRE_RTOL * np.abs(ref)
    assert (err <= bound).all(), "%s: max|d|=%.3e (bound %.1e+%.1e|ref|)" % (what, float(err.max()), PRE_ATOL, PRE_RTOL)


def pre_worst(got, ref):
    """Largest |d| / (PRE_ATOL + PRE_RTOL |ref|) over a pre-clamp output, and the largest |d|."""
    err = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
    return float((err / (PRE_ATOL + PRE_RTOL * np.abs(ref))).max()), float(err.max())


# ---- full-size rule (round 5; limits re-set in round 6 from the measured tail) ---------------------------------------------
# At a BASELINE configuration's full size (1.2 - 4 million values per white-noise frame) the every-value bounds above are
# not a property a float32 evaluation of this network has.  Decoder.norm[0] multiplies ~1e-7 of rounding noise in near-dead
# relu4_1 channels by rstd up to 4e3; where such a channel carries signal, a patch of pixels behind it lands at several
# times the bound in EVERY float32 evaluation, measured against the same network with every convolution accumulated in
# float64 ("torch64": the error of the evaluation under test alone).  Which evaluation is hit how hard is a draw per frame.
# The tail, measured over ALL 128 frames of bench.py's first step (white-noise 640 x 640, sixteen per launch, the bench's
# B = 38 state: tools/fullsize_tail.py -> profiles/r06_fullsize_tail.txt), maxima over the frames:
#                                   worst / bound   values outside   99.99th pct   mean     image max |d|   values > 0.05
#   the library, default choice          2.96        16 (1.3e-5)        0.436      0.0144      0.0765        2 (in 1 frame of 128)
#   the reference's own arithmetic       7.40        52 (4.2e-5)        0.650      0.0236      0.0825        9 (in 1 frame)
#     (the oracle on torch's float32 conv2d)
#   the library, F(2x2,3x3) everywhere  15.29       142 (1.2e-4)        1.092      0.0108      0.2427       75 (3 frames)
# The same over 32 frames at 1024 x 1024 (1152 x 1152 padded, four per launch; 3.2x the values per frame, so deeper extremes:
# profiles/r06_fullsize_tail_1024.txt) and 64 frames at 256 x 256 (384 x 384, 32 per launch; r06_fullsize_tail_256.txt):
#   the library, default choice, 1152^2  5.22        84 (2.1e-5)        0.463      0.0149      0.0856       11 (2 frames of 32)
#   the reference's own arithmetic       9.69       100 (2.5e-5)        0.563      0.0244      0.1207       21 (8 frames)
#   the library, default choice, 384^2   1.61         1 (2.3e-6)        0.357      0.0136      0.0172        0
#   the reference's own arithmetic       1.62         5 (1.1e-5)        0.455      0.0224      0.0213        0
# So at full size the HIP path is held, per kernel family, to ABSOLUTE limits = those maxima (over the three sizes) with ~1.3x headroom — no clause
# relative to the float32 oracle's own tail any more (the round-5 rule let the library reach ~45x where the
# oracle sat at 15x).  The DEFAULT choice — what ships and what bench.py times — additionally keeps the stated image
# tolerance on every value in every frame the suite tests (`strict`; 127 of the 128 frames at 640^2, 30 of the 32 at 1152^2 and all
# 64 at 384^2 keep it; none of the tested frames is among the three).  Small frames and every reference golden keep the every-value bounds above.
FULL_PCT = 99.99
FULL_LIMITS = {
    #            mean error / bound, 99.99th percentile, share of values outside the bound, worst value / bound, values beyond IMG_ATOL, image max |d|
    "default": dict(mean=0.02, pct=0.55, over_frac=3e-5, worst=6.5, img_over=15, img_worst=0.11),
    "f22":     dict(mean=0.02, pct=1.40, over_frac=1.6e-4, worst=20.0, img_over=100, img_worst=0.32),
}


def _family(family):
    """A suite forced onto F(2x2,3x3) everywhere (RRV_F43=0 in the environment) is held to THAT family's limits where a test names the default."""
    return "f22" if family == "default" and os.environ.get("RRV_F43") == "0" else family


def pre_full_size(got, ref32, ref64, what="pre-clamp", family="default"):
    """Pre-clamp side of the full-size rule above.  `ref64`: the oracle with every convolution accumulated in float64;
    `ref32`: its convolutions on torch's float32 conv2d (reported beside the result, not part of the rule).  `family`:
    "default" (the library's kernel choice) or "f22" (F(2x2,3x3) everywhere).  Returns (worst, values over the bound, 99.99th
    percentile, mean, the float32 oracle's own worst, its values over the bound) of error / bound."""
    L = FULL_LIMITS[_family(family)]
    r64 = np.asarray(ref64, np.float64)
    bound = PRE_ATOL + PRE_RTOL * np.abs(r64)
    mine = np.abs(np.asarray(got, np.float64) - r64) / bound
    theirs = np.abs(np.asarray(ref32, np.float64) - r64) / bound
    worst, over, p, mean = float(mine.max()), int((mine > 1.0).sum()), float(np.percentile(mine, FULL_PCT)), float(mine.mean())
    t_worst, t_over = float(theirs.max()), int((theirs > 1.0).sum())
    assert mean <= L["mean"], "%s: mean error / bound %.4f (limit %.2f)" % (what, mean, L["mean"])
    assert p <= L["pct"], "%s: %.2fth percentile of error / bound is %.3f (limit %.2f)" % (what, FULL_PCT, p, L["pct"])
    allowed = int(L["over_frac"] * mine.size)
    assert over <= allowed, "%s: %d of %d values outside the bound (allowed %d; the float32 oracle itself: %d)" % (what, over, mine.size, allowed, t_over)
    assert worst <= L["worst"], "%s: worst value at %.2fx the bound from the float64-accumulated oracle (limit %.1fx; the float32 oracle itself: %.2fx)" % (what, worst, L["worst"], t_worst)
    return worst, over, p, mean, t_worst, t_over


def img_full_size(got, ref, what="image", ref32=None, family="default", strict=False):
    """Image side of the full-size rule: `ref` = the float64-accumulated oracle's image.  strict: every value within IMG_ATOL
    (the stated tolerance); else the family's measured limits.  `ref32` (the float32 oracle's image) is only reported.
    Returns (max |d|, values beyond IMG_ATOL)."""
    strict = strict and _family(family) == family
    L = FULL_LIMITS[_family(family)]
    d = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
    worst, over = float(d.max()), int((d > IMG_ATOL).sum())
    theirs = ""
    if ref32 is not None:
        t = np.abs(np.asarray(ref32, np.float64) - np.asarray(ref, np.float64))
        theirs = "; the float32 oracle itself: max |d| %.3f, %d beyond" % (float(t.max()), int((t > IMG_ATOL).sum()))
    if strict:
        assert over == 0, "%s: %d of %d values beyond %.2f grey levels, max |d| %.3f%s" % (what, over, d.size, IMG_ATOL, worst, theirs)
        return worst, over
    assert over <= L["img_over"], "%s: %d of %d values beyond %.2f grey levels (allowed %d%s)" % (what, over, d.size, IMG_ATOL, L["img_over"], theirs)
    assert worst <= L["img_worst"], "%s: max |d| %.3f grey levels (limit %.2f%s)" % (what, worst, L["img_worst"], theirs)
    return worst, over


def _layer_of(field):
    """'dec.norm1.min(x)' -> 'dec.norm1': the four statistics of one normalisation layer come out of one tensor and share
    its conditioning; a filter / a style level is its own group (21 groups)."""
    return field.rsplit(".", 1)[0]


def _field_limits(ref_a, ref_b, factor):
    """Limit per field from the two references' own disagreement: a LAYER in which they disagree by more than the regular
    bound is ill-conditioned — its fields get `factor` x THAT layer's own worst disagreement (the four statistics of a
    normalisation layer share one tensor and its conditioning); every other layer keeps max(1, factor x its own
    disagreement), i.e. essentially the regular bound.  (Until round 4 an ill-conditioned layer borrowed the GLOBAL worst
    disagreement — 30x for a layer whose own was 5x)
    Returns (limit per field, the references' own ratio of that field's layer)."""
    rows = state_fields(ref_a, ref_b)
    layer = {}
    for name, ratio, *_ in rows:
        layer[_layer_of(name)] = max(layer.get(_layer_of(name), 0.0), ratio)
    lim = {}
    for name, *_ in rows:
        own = layer[_layer_of(name)]
        lim[name] = max(1.0, factor * own)
    return lim, {row[0]: layer[_layer_of(row[0])] for row in rows}


def assert_state_close_conditioned(got, ref32, ref64, what="state", factor=2.5):
    """For weight sets whose saved state is ILL-CONDITIONED in float32 (tests/golden/global_a_dec4: the reference's own
    float32 run misses its float64 run by 30x the bound above in `Filter3.F1.filter`, 10x in `Filter3.F2.filter`, 5x in
    `dec.norm1.mean`, and sits inside the bound everywhere else; its 1-thread and 8-thread runs differ by as much).
    Only the layers in which the reference's own float32 run leaves the bound (3 of the 21 groups: the four statistics of a
    normalisation layer, a filter, a style level) are widened — each to `factor` times the reference's own miss IN THAT LAYER; every
    other layer is held to the regular bound (_field_limits).  Why 2.5 and not ~1: the miss is rounding noise amplified by
    the dynamic filters, not a property of an implementation — two direct-form float32 restatements of the same algorithm
    sit at 1.2x (torch conv2d) and 2.0x (nine numpy GEMMs) the reference's own worst field, and a run with EVERY
    convolution accumulated in float64 and rounded once is still at 0.8x (profiles/r04_dec4_conditioning.txt)."""
    lims, theirs = _field_limits(ref32, ref64, factor)
    bad, worst = [], (0.0, "", 0.0)
    for name, ratio, k, g, r, atol in state_fields(got, ref64):
        if ratio > lims[name]:
            bad.append("%s[%d]: %.1fx the bound from the float64 reference (reference float32 itself: %.1fx; allowed %.1fx)" % (name, k, ratio, theirs[name], lims[name]))
        if ratio > worst[0]:
            worst = (ratio, name, theirs[name])
    assert not bad, "%s: %s" % (what, "; ".join(bad))
    return worst[0], max(theirs.values())


def assert_state_close_two_refs(got, ref_a, ref_b, what="state", factor=1.25):
    """When two float32 restatements of the SAME algorithm (the oracle with its convolutions as nine numpy GEMMs, and on
    torch's conv2d) disagree in a layer by more than the regular bound, that layer is ill-conditioned for that input (one
    frame, B = 1: an extremum behind the near-dead relu4_1 channels): in THAT layer the HIP state may be as far from
    restatement A as `factor` times their own disagreement; every other layer keeps the regular bound."""
    lims, spread = _field_limits(ref_b, ref_a, factor)
    bad, worst = [], 0.0
    for name, ratio, k, g, r, atol in state_fields(got, ref_a):
        worst = max(worst, ratio)
        if ratio > lims[name]:
            bad.append("%s[%d]: %.1fx the bound from restatement A (the two restatements differ by %.1fx there; allowed %.1fx)" % (name, k, ratio, spread[name], lims[name]))
    assert not bad, "%s: %s" % (what, "; ".join(bad))
    return worst, max(spread.values())
