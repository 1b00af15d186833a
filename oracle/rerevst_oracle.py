"""CPU ORACLE (test infrastructure, NOT product code) for the ReReVST per-frame path.

A numpy float32 restatement of the reference algorithm behind ``Stylization``
(test/framework.py:56-118) in global-feature-sharing mode
(test/style_network_global.py).  It is written from the operator sequence, NHWC, and is
used ONLY by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
as the checker / the timed CPU port.  The product path (rerevst-code_amd/) never imports
it and fails loudly when the HIP library is missing.

Parity status: the reference has no tests or golden vectors of its own;
this oracle is pinned against outputs of the UNMODIFIED reference network imported in the
authoring container (tests/golden/make_goldens.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).

Each function cites the reference file:line it follows (paths relative to the reference
root).
"""
import numpy as np

F32 = np.float32
MEAN = np.array([0.485, 0.456, 0.406], dtype=F32)   # test/framework.py:31 (RGB order)
STD = np.array([0.229, 0.224, 0.225], dtype=F32)    # test/framework.py:32

VGG_IDX = [0, 2, 5, 7, 10, 12, 14, 16, 19]          # conv indices of vgg19.features[0:21]
POOL_AFTER = {2, 7, 16}                              # MaxPool follows these convs (cfg E)
STYLE_SLICE = {0: 1, 2: 2, 5: 2, 7: 3, 10: 3, 12: 4, 14: 4, 16: 4, 19: 4}
STYLE_TAP_AFTER = {0: "relu1_1", 5: "relu2_1", 10: "relu3_1", 19: "relu4_1"}

# ---- shared-state blob layout (17 536 floats) -------------------------
# 11 norm layers x {mean, rstd, lo, hi}[C], then 6 filters [32][32], then style
# (mean[C], std[C]) for relu1_1..relu4_1.
NORM_NAMES = ["dec.norm0", "dec.norm1", "dec.norm2", "dec.norm3", "dec.norm4",
              "slice4.norm1", "slice4.norm2", "slice3.norm1", "slice3.norm2",
              "slice2.norm1", "slice2.norm2"]
NORM_CH = [512, 512, 256, 128, 64, 256, 256, 128, 128, 64, 64]
FILTER_NAMES = ["Filter1.F1", "Filter1.F2", "Filter2.F1", "Filter2.F2", "Filter3.F1", "Filter3.F2"]
STYLE_NAMES = ["relu1_1", "relu2_1", "relu3_1", "relu4_1"]
STYLE_CH = [64, 128, 256, 512]
STATE_FLOATS = 4 * sum(NORM_CH) + 6 * 1024 + 2 * sum(STYLE_CH)
assert STATE_FLOATS == 17536


# ---- primitive operators ---------------------------------------------------------------

F64 = np.float64


def _mean32(a, **kw):
    """Mean with a float64 accumulator, rounded once to float32.  numpy's float32 reduction over the leading axes adds
    rows one after the other (error ~ N * 6e-8 relative: 7e-6 on a CONSTANT channel over 27 648 pixels), torch's
    cascaded sums are accurate to an ulp; after `InstanceNorm.compute`'s rsqrt(var + 1e-8) such an error is multiplied
    by up to 1e4 (tests/golden/global_a_dead).  The correctly rounded mean is what the reference's own arithmetic
    approximates."""
    return np.mean(a, dtype=F64, **kw).astype(F32)


# "numpy": nine shifted GEMMs (the default; what the parity tests use).  "torch": the same convolution through
# torch.nn.functional.conv2d on the CPU — the primitive the reference itself calls (nn.Conv2d -> oneDNN) — used by
# bench.py's cpu_baseline leg so that the CPU column is timed on the reference's own arithmetic library.
# "torch64": every convolution ACCUMULATED in float64 and rounded once to float32 (tensors stay float32 between layers, as
# in the reference).  Two float32 evaluations of this network differ from each other by the sum of their own rounding
# noise, which Decoder.norm[0] amplifies on near-dead channels (rstd up to 4e3); against this backend a test sees the
# error of the implementation under test alone.  Used where a full-size comparison would otherwise measure the oracle's
# float32 summation order as much as the kernel's (tests/test_gpu_default_choice.py, tools/parity_margin.py).
CONV_BACKEND = "numpy"


def set_conv_backend(name):
    global CONV_BACKEND
    assert name in ("numpy", "torch", "torch64")
    CONV_BACKEND = name


def _conv3x3_torch(x, w, b, f64=False):
    import torch
    import torch.nn.functional as TF
    with torch.no_grad():
        t = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
        wt = torch.from_numpy(np.ascontiguousarray(w))
        bt = None if b is None else torch.from_numpy(np.ascontiguousarray(b, dtype=F32))
        if f64:
            y = TF.conv2d(t.double(), wt.double(), None if bt is None else bt.double(), padding=1).float()
        else:
            y = TF.conv2d(t, wt, bt, padding=1)
        return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())


def conv3x3(x, w, b=None):
    """Zero-pad-1, stride-1 3x3 convolution.  x [B,H,W,Cin] f32, w OIHW [Cout,Cin,3,3].
    (nn.Conv2d(kernel_size=3, padding=1): style_network_global.py:103-104,145,182,186,341;
    vgg19.features convs.)  Evaluated as nine shifted [B*H*W,Cin]x[Cin,Cout] products."""
    if CONV_BACKEND in ("torch", "torch64"):
        return _conv3x3_torch(x, w, b, CONV_BACKEND == "torch64")
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    xp = np.zeros((B, H + 2, W + 2, Cin), dtype=F32)
    xp[:, 1:H + 1, 1:W + 1, :] = x
    out = np.zeros((B, H, W, Cout), dtype=F32)
    # row strips bound the temporary copies for large images
    strip = max(1, (1 << 22) // max(1, W * max(Cin, Cout)))
    for y0 in range(0, H, strip):
        y1 = min(H, y0 + strip)
        acc = np.zeros((B * (y1 - y0) * W, Cout), dtype=F32)
        for ky in range(3):
            for kx in range(3):
                a = np.ascontiguousarray(xp[:, y0 + ky:y1 + ky, kx:kx + W, :]).reshape(-1, Cin)
                acc += a @ np.ascontiguousarray(w[:, :, ky, kx].T)
        out[:, y0:y1] = acc.reshape(B, y1 - y0, W, Cout)
    if b is not None:
        out += b.astype(F32)
    return out


def conv1x1(x, w):
    """Bias-free 1x1 convolution, w [Cout,Cin,1,1] (conv_shortcut :105) or [Cout,Cin]."""
    B, H, W, Cin = x.shape
    w2 = w.reshape(w.shape[0], Cin)
    if CONV_BACKEND == "torch64":
        return (x.reshape(-1, Cin).astype(F64) @ np.ascontiguousarray(w2.T).astype(F64)).astype(F32).reshape(B, H, W, -1)
    return (x.reshape(-1, Cin) @ np.ascontiguousarray(w2.T)).reshape(B, H, W, -1)


def maxpool2(x):
    """nn.MaxPool2d(2,2) (floor): vgg19.features[4,9,18]."""
    B, H, W, C = x.shape
    H2, W2 = H // 2, W // 2
    v = x[:, :H2 * 2, :W2 * 2, :].reshape(B, H2, 2, W2, 2, C)
    return v.max(axis=(2, 4))


def upsample2(x):
    """F.interpolate(mode='nearest', scale_factor=2): style_network_global.py:113."""
    return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)


def relu(x):
    return np.maximum(x, F32(0))


def lrelu(x):
    """nn.LeakyReLU(0.2): style_network_global.py:106,192."""
    return np.where(x >= 0, x, x * F32(0.2)).astype(F32)


# ---- image <-> tensor ------------------------------------------------------------------

def image_to_tensor(img_bgr_u8):
    """numpy2tensor + transform_image (test/framework.py:26-35): BGR->RGB, float, /255,
    (x-mean)/std.  Returns NHWC [1,H,W,3] (RGB channel order)."""
    x = img_bgr_u8[..., ::-1].astype(F32)
    x = x / F32(255.0)
    x = (x - MEAN) / STD
    return x[None]


def tensor_to_image(x):
    """transform_back_image + tensor2numpy (test/framework.py:39-49): x*std+mean,
    clamp(0,1), *255, RGB->BGR; float32 HWC."""
    y = x[0] * STD + MEAN
    y = np.clip(y, F32(0), F32(1)) * F32(255)
    return np.ascontiguousarray(y[..., ::-1]).astype(F32)


def rgb2gray(x):
    """TransformerNet.RGB2Gray (style_network_global.py:487-497).  The grey weights are
    applied to tensor channels 2,1,0 of an RGB tensor (quirk Q5)."""
    img = x * STD + MEAN
    g = img[..., 2:3] * F32(0.299) + img[..., 1:2] * F32(0.587) + img[..., 0:1] * F32(0.114)
    g = np.broadcast_to(g, img.shape)
    return ((g - MEAN) / STD).astype(F32)


# ---- network ---------------------------------------------------------------------------

class Net:
    """Holds OIHW weights keyed by the reference state_dict names."""

    def __init__(self, weights):
        self.w = {k: np.asarray(v, dtype=F32) for k, v in weights.items()}

    def encoder(self, x):
        """Encoder.forward = vgg19.features[0:21] (style_network_global.py:271-281)."""
        for idx in VGG_IDX:
            x = relu(conv3x3(x, self.w["Encoder.slice.%d.weight" % idx], self.w["Encoder.slice.%d.bias" % idx]))
            if idx in POOL_AFTER:
                x = maxpool2(x)
        return x

    def encoder_style(self, x):
        """EncoderStyle.forward + cal_mean_std (style_network_global.py:304-331):
        per-level mean and sqrt(UNBIASED var + 1e-5) (quirk Q3); keeps the relu4_1 map."""
        feats = {}
        for idx in VGG_IDX:
            p = "EncoderStyle.slice%d.%d." % (STYLE_SLICE[idx], idx)
            x = relu(conv3x3(x, self.w[p + "weight"], self.w[p + "bias"]))
            if idx in STYLE_TAP_AFTER:
                C = x.shape[-1]
                flat = x.reshape(-1, C)
                var = flat.var(axis=0, ddof=1, dtype=F64).astype(F32) + F32(1e-5)
                feats[STYLE_TAP_AFTER[idx]] = (_mean32(flat, axis=0), np.sqrt(var).astype(F32))
            if idx in POOL_AFTER:
                x = maxpool2(x)
        feats["map"] = x
        return feats


class NormState:
    """InstanceNorm saved statistics (style_network_global.py:27-84)."""

    def __init__(self):
        self.mean = self.rstd = self.lo = self.hi = None

    def compute(self, x, eps=F32(1e-8)):
        """InstanceNorm.compute :59-77: biased centred mean-square over (batch,H,W),
        rsqrt; min/max of the normalised batch (quirks Q3, Q4)."""
        C = x.shape[-1]
        flat = x.reshape(-1, C)
        self.mean = _mean32(flat, axis=0)
        xc = flat - self.mean
        self.rstd = (F32(1) / np.sqrt(_mean32(xc * xc, axis=0) + eps)).astype(F32)
        xn = xc * self.rstd
        self.hi = xn.max(axis=0)
        self.lo = xn.min(axis=0)
        return xn.reshape(x.shape)

    def forward(self, x):
        """InstanceNorm.forward :43-57."""
        if self.mean is None:
            raise RuntimeError("state not computed: call compute() before transfer()")
        y = (x - self.mean) * self.rstd
        y = np.maximum(self.lo, y)
        return np.minimum(self.hi, y)


def apply_filter(x, filt):
    """KernelFilter.apply_filter (style_network_global.py:194-208): 1x1 conv with
    weight[out=i,in=j] = filter[i,j] (quirk Q2).  The reference zips B input chunks with
    the batch-1 filter, so only frame 0 survives (quirk Q1): x[0:1] is used."""
    x0 = x[0:1]
    return (x0.reshape(-1, x0.shape[-1]) @ np.ascontiguousarray(filt.T)).reshape(x0.shape[:-1] + (filt.shape[0],))


class Decoder:
    def __init__(self, net):
        self.net = net
        self.clean()

    def clean(self):
        """Decoder.clean (style_network_global.py:409-419)."""
        self.norm = [NormState() for _ in range(5)]
        self.bnorm = {b: (NormState(), NormState()) for b in ("slice4", "slice3", "slice2")}
        self.filters = {n: None for n in FILTER_NAMES}

    # -- helpers
    def _w(self, k):
        return self.net.w["Decoder." + k]

    def _predict(self, name, content, style):
        """FilterPredictor.compute (:161-172): FC([mean_{B,HW} down(content),
        mean_{HW} down(style)]) -> [32,32].  `content` is the B-frame normalised feature,
        `style` the normalised style map; FC is a plain nn.Linear (no activation)."""
        p = name + "."
        c = conv3x3(content, self._w(p + "down_sample.0.weight"), self._w(p + "down_sample.0.bias"))
        c = _mean32(_mean32(c.reshape(c.shape[0], -1, c.shape[-1]), axis=1), axis=0)
        s = conv3x3(style, self._w(p + "down_sample.0.weight"), self._w(p + "down_sample.0.bias"))
        s = _mean32(s.reshape(-1, s.shape[-1]), axis=0)
        v = np.concatenate([c, s]).astype(F32)
        f = self._w(p + "FC.weight") @ v + self._w(p + "FC.bias")
        return f.reshape(32, 32).astype(F32)

    def _kernel_filter(self, fname, content, style=None, compute=False):
        """KernelFilter.forward :210-217 / .compute :223-230."""
        p = fname + "."
        d = conv3x3(content, self._w(p + "down_sample.0.weight"), self._w(p + "down_sample.0.bias"))
        if compute:
            self.filters[fname + ".F1"] = self._predict(fname + ".F1", content, style)
        d = lrelu(apply_filter(d, self.filters[fname + ".F1"]))
        if compute:
            self.filters[fname + ".F2"] = self._predict(fname + ".F2", content, style)
        d = apply_filter(d, self.filters[fname + ".F2"])
        u = conv3x3(d, self._w(p + "upsample.0.weight"), self._w(p + "upsample.0.bias"))
        return content + u          # [B,...] + [1,...]: frame-0 residual broadcast (Q1)

    def _resblock(self, blk, x, compute=False):
        """ResidualBlock.forward :111-122 / .compute :124-135."""
        n1, n2 = self.bnorm[blk]
        x = upsample2(x)
        xs = conv1x1(x, self._w(blk + ".conv_shortcut.weight"))
        h = lrelu(conv3x3(x, self._w(blk + ".conv1.weight"), self._w(blk + ".conv1.bias")))
        h = n1.compute(h) if compute else n1.forward(h)
        h = lrelu(conv3x3(h, self._w(blk + ".conv2.weight"), self._w(blk + ".conv2.bias")))
        h = n2.compute(h) if compute else n2.forward(h)
        return xs + h

    def _adain(self, k, x, style_ms, compute=False):
        """Decoder.AdaIN :357-364 / AdaIN_compute :383-390."""
        n = self.norm[k]
        xn = n.compute(x) if compute else n.forward(x)
        return xn * style_ms[1] + style_ms[0]

    def run(self, x, F_style, compute=False):
        """Decoder.forward :441-451 / Decoder.compute :425-439."""
        m4, s4 = F_style["relu4_1"]
        h = self.norm[0].compute(x) if compute else self.norm[0].forward(x)
        style_n = ((F_style["map"] - m4) / s4).astype(F32) if compute else None
        for f in ("Filter1", "Filter2", "Filter3"):
            h = self._kernel_filter(f, h, style_n, compute)
        h = self._adain(1, h, F_style["relu4_1"], compute)
        h = self._resblock("slice4", h, compute)
        h = self._adain(2, h, F_style["relu3_1"], compute)
        h = self._resblock("slice3", h, compute)
        h = self._adain(3, h, F_style["relu2_1"], compute)
        h = self._resblock("slice2", h, compute)
        h = self._adain(4, h, F_style["relu1_1"], compute)
        if compute:
            return None
        return conv3x3(h, self._w("slice1.weight"), self._w("slice1.bias"))


def inorm_frame(x, eps=F32(1e-8)):
    """InstanceNorm.forward of the frame-mode network (test/style_network_frame.py:39-43): per-image,
    per-channel over (H,W), biased, rsqrt; no saved state, no clamp."""
    m = _mean32(x, axis=(1, 2), keepdims=True)
    xc = x - m
    r = (F32(1) / np.sqrt(_mean32(xc * xc, axis=(1, 2), keepdims=True) + eps)).astype(F32)
    return xc * r


class FrameDecoder:
    """Decoder of test/style_network_frame.py (use_Global=False): per-frame statistics and per-frame
    filter prediction; same weights / state_dict keys as the global model."""

    def __init__(self, net):
        self.net = net

    def _w(self, k):
        return self.net.w["Decoder." + k]

    def _predict(self, name, content, style):
        """FilterPredictor.forward (style_network_frame.py:53-62)."""
        p = name + "."
        c = conv3x3(content, self._w(p + "down_sample.0.weight"), self._w(p + "down_sample.0.bias"))
        c = _mean32(c.reshape(-1, c.shape[-1]), axis=0)
        s = conv3x3(style, self._w(p + "down_sample.0.weight"), self._w(p + "down_sample.0.bias"))
        s = _mean32(s.reshape(-1, s.shape[-1]), axis=0)
        f = self._w(p + "FC.weight") @ np.concatenate([c, s]).astype(F32) + self._w(p + "FC.bias")
        return f.reshape(32, 32).astype(F32)

    def _kernel_filter(self, fname, content, style):
        """KernelFilter.forward (style_network_frame.py:97-105)."""
        p = fname + "."
        d = conv3x3(content, self._w(p + "down_sample.0.weight"), self._w(p + "down_sample.0.bias"))
        d = lrelu(apply_filter(d, self._predict(fname + ".F1", content, style)))
        d = apply_filter(d, self._predict(fname + ".F2", content, style))
        return content + conv3x3(d, self._w(p + "upsample.0.weight"), self._w(p + "upsample.0.bias"))

    def _resblock(self, blk, x):
        """ResidualBlock.forward (style_network_frame.py ResidualBlock: one stateless norm used twice)."""
        x = upsample2(x)
        xs = conv1x1(x, self._w(blk + ".conv_shortcut.weight"))
        h = inorm_frame(lrelu(conv3x3(x, self._w(blk + ".conv1.weight"), self._w(blk + ".conv1.bias"))))
        h = inorm_frame(lrelu(conv3x3(h, self._w(blk + ".conv2.weight"), self._w(blk + ".conv2.bias"))))
        return xs + h

    def forward(self, x, F_style):
        """Decoder.forward / AdaIN_filter / AdaIN (style_network_frame.py:313-358)."""
        m4, s4 = F_style["relu4_1"]
        h = inorm_frame(x)
        sn = ((F_style["map"] - m4) / s4).astype(F32)
        for f in ("Filter1", "Filter2", "Filter3"):
            h = self._kernel_filter(f, h, sn)
        h = h * s4 + m4                      # no second normalisation here (unlike the global model)
        h = self._resblock("slice4", h)
        h = inorm_frame(h) * F_style["relu3_1"][1] + F_style["relu3_1"][0]
        h = self._resblock("slice3", h)
        h = inorm_frame(h) * F_style["relu2_1"][1] + F_style["relu2_1"][0]
        h = self._resblock("slice2", h)
        h = inorm_frame(h) * F_style["relu1_1"][1] + F_style["relu1_1"][0]
        return conv3x3(h, self._w("slice1.weight"), self._w("slice1.bias"))


class Stylization:
    """Mirror of the reference ``Stylization`` call surface (test/framework.py:56-118),
    constructed from a weight dict instead of a checkpoint path."""

    def __init__(self, weights, use_Global=True):
        self.use_Global = use_Global
        self.net = Net(weights)
        self.dec = Decoder(self.net)
        self.F_style = None
        self.F_patches = []

    # test/framework.py:99-104 -> TransformerNet.generate_style_features :465-469
    def prepare_style(self, style):
        self.F_style = self.net.encoder_style(image_to_tensor(style))   # style is NOT greyscaled

    # test/framework.py:93 -> TransformerNet.clean :480-485
    def clean(self):
        self.F_patches = []
        self.dec.clean()

    # test/framework.py:82-86 -> TransformerNet.add :471-475
    def add(self, patch):
        self.F_patches.append(self.net.encoder(rgb2gray(image_to_tensor(patch))))

    # test/framework.py:88-91 -> TransformerNet.compute :477-478
    def compute(self):
        self.dec.run(np.concatenate(self.F_patches, axis=0), self.F_style, compute=True)

    # test/framework.py:106-118 -> TransformerNet.forward :499-501
    def transfer(self, frame, return_preclamp=False):
        f = self.net.encoder(rgb2gray(image_to_tensor(frame)))
        if self.use_Global:
            y = self.dec.run(f, self.F_style, compute=False)
        else:
            y = FrameDecoder(self.net).forward(f, self.F_style)
        if return_preclamp:
            return y
        return tensor_to_image(y)

    # ---- shared state blob (layout above) ----
    def get_state(self):
        parts = []
        norms = list(self.dec.norm) + [n for b in ("slice4", "slice3", "slice2") for n in self.dec.bnorm[b]]
        for n in norms:
            parts += [n.mean, n.rstd, n.lo, n.hi]
        for name in FILTER_NAMES:
            parts.append(self.dec.filters[name].reshape(-1))
        for name in STYLE_NAMES:
            parts += list(self.F_style[name])
        blob = np.concatenate([np.asarray(p, dtype=F32).reshape(-1) for p in parts])
        assert blob.size == STATE_FLOATS
        return blob

    def set_state(self, blob):
        blob = np.asarray(blob, dtype=F32).reshape(-1)
        assert blob.size == STATE_FLOATS
        o = 0
        norms = list(self.dec.norm) + [n for b in ("slice4", "slice3", "slice2") for n in self.dec.bnorm[b]]
        for n, C in zip(norms, NORM_CH):
            n.mean, n.rstd, n.lo, n.hi = (blob[o + i * C:o + (i + 1) * C].copy() for i in range(4))
            o += 4 * C
        for name in FILTER_NAMES:
            self.dec.filters[name] = blob[o:o + 1024].reshape(32, 32).copy()
            o += 1024
        if self.F_style is None:
            self.F_style = {}
        for name, C in zip(STYLE_NAMES, STYLE_CH):
            self.F_style[name] = (blob[o:o + C].copy(), blob[o + C:o + 2 * C].copy())
            o += 2 * C


class MultiStylization:
    """Mirror of "Multi-style Interpolation/stylization.py":42-100 (S styles, per-frame
    weight vector).  Per style the preparation pass is the single-style one
    (style_network.py:415-430); per frame every saved quantity q is replaced by
    sum_s w_s q_s (:35-53 mean/rstd/min/max, :135-139 filters, :348-360 style moments)
    before the same decoder forward (:432-460).  The encoder runs in
    generate_content_features and its output is what add_patch / transfer receive."""

    def __init__(self, weights, style_num=1):
        self.net = Net(weights)
        self.style_num = style_num
        self.per_style = [Stylization(weights) for _ in range(style_num)]
        for p in self.per_style:
            p.net = self.net
            p.dec.net = self.net
        self.F_patches = []

    def prepare_style(self, style_images):          # stylization.py:71-79
        for sid, img in enumerate(style_images):
            self.per_style[sid].prepare_style(img)

    def generate_content_features(self, content):   # stylization.py:87-92
        return self.net.encoder(rgb2gray(image_to_tensor(content)))

    def add_patch(self, feature):                   # stylization.py:66-67
        self.F_patches.append(feature)

    def compute_norm(self):                         # stylization.py:81-83, style_network.py:498-500
        x = np.concatenate(self.F_patches, axis=0)
        for p in self.per_style:
            p.dec.run(x, p.F_style, compute=True)
        self.F_patches = []

    def clean(self):                                # stylization.py:85
        for p in self.per_style:
            p.dec.clean()

    def get_state(self, style_id):
        return self.per_style[style_id].get_state()

    def transfer(self, feature, style_weight=(1.0,), return_preclamp=False):   # stylization.py:94-100
        blob = np.zeros(STATE_FLOATS, dtype=F32)
        for sid in range(self.style_num):
            blob = blob + self.per_style[sid].get_state() * F32(style_weight[sid])
        tmp = Stylization({})
        tmp.net, tmp.dec.net = self.net, self.net
        tmp.set_state(blob.astype(F32))
        y = tmp.dec.run(feature, tmp.F_style, compute=False)
        return y if return_preclamp else tensor_to_image(y)


def sample_indices_multistyle(n, interval=16):
    """VideoStylization.SeqNormPrePare ("Multi-style Interpolation/test.py":72-85): frames
    s*16 for s < (n-1)//16+1, then the last frame (again, if it was already sampled)."""
    return [s * interval for s in range((n - 1) // interval + 1)] + [n - 1]


# ---- driver-side helpers (test/generate_real_video.py) ----------------------------------

def padded_size(n):
    """ReshapeTool.process (generate_real_video.py:66-76): roundup64(n+128)."""
    m = n + 128
    if m % 64:
        m += 64 - m % 64
    return m


def reflect_pad(img, PH, PW):
    """cv2.copyMakeBorder(img, 64, PH-64-H, 64, PW-64-W, BORDER_REFLECT) (:81-82).
    BORDER_REFLECT is edge-inclusive = numpy 'symmetric' (quirk Q6)."""
    H, W = img.shape[:2]
    return np.pad(img, ((64, PH - 64 - H), (64, PW - 64 - W), (0, 0)), mode="symmetric")


def sample_indices(n, interval=8):
    """Sampling schedule of generate_real_video.py:129-143: every 8th frame for
    s < (n-1)//8, then the LAST frame."""
    return [s * interval for s in range((n - 1) // interval)] + [n - 1]
