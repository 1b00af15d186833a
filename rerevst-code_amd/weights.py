"""Weight table, seeded synthetic generator and checkpoint reader.

The reference loads a 107-key ``state_dict`` with ``load_state_dict(torch.load(ckpt))``
(test/framework.py:74-75).  The released checkpoint is a 0-byte placeholder in the
reference tree (README.md:83-85), so parity is proven on seeded synthetic weights that
carry the real key/shape set; a real ``.pth`` supplied by a user is read through the same
table.

Only the keys the inference path touches are listed (the ``Vgg19.*`` module is deleted by
the reference itself on first use: test/style_network_global.py:467-469).
"""
import zlib

import numpy as np

# (Cin, Cout) of vgg19.features[0:21] convolutions, keyed by their nn.Sequential index
# (torchvision cfg "E": conv,relu,conv,relu,pool,...).  test/style_network_global.py:271-281
VGG_CONVS = [(0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256),
             (12, 256, 256), (14, 256, 256), (16, 256, 256), (19, 256, 512)]
# EncoderStyle splits the same 9 convs over four slices (:295-302)
_STYLE_SLICE = {0: 1, 2: 2, 5: 2, 7: 3, 10: 3, 12: 4, 14: 4, 16: 4, 19: 4}

INNER = 32  # KernelFilter / FilterPredictor inner_channel (:143,:179)


def weight_table():
    """Ordered {state_dict key: shape} for Encoder, EncoderStyle and Decoder."""
    t = {}
    for blk, (cin, cout) in (("slice4", (512, 256)), ("slice3", (256, 128)), ("slice2", (128, 64))):
        p = "Decoder.%s." % blk
        t[p + "conv1.weight"] = (cout, cin, 3, 3)
        t[p + "conv1.bias"] = (cout,)
        t[p + "conv2.weight"] = (cout, cout, 3, 3)
        t[p + "conv2.bias"] = (cout,)
        t[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
    t["Decoder.slice1.weight"] = (3, 64, 3, 3)
    t["Decoder.slice1.bias"] = (3,)
    for f in (1, 2, 3):
        p = "Decoder.Filter%d." % f
        t[p + "down_sample.0.weight"] = (INNER, 512, 3, 3)
        t[p + "down_sample.0.bias"] = (INNER,)
        t[p + "upsample.0.weight"] = (512, INNER, 3, 3)
        t[p + "upsample.0.bias"] = (512,)
        for g in ("F1", "F2"):
            q = p + g + "."
            t[q + "down_sample.0.weight"] = (INNER, 512, 3, 3)
            t[q + "down_sample.0.bias"] = (INNER,)
            t[q + "FC.weight"] = (INNER * INNER, 2 * INNER)
            t[q + "FC.bias"] = (INNER * INNER,)
    for idx, cin, cout in VGG_CONVS:
        t["Encoder.slice.%d.weight" % idx] = (cout, cin, 3, 3)
        t["Encoder.slice.%d.bias" % idx] = (cout,)
    for idx, cin, cout in VGG_CONVS:
        p = "EncoderStyle.slice%d.%d." % (_STYLE_SLICE[idx], idx)
        t[p + "weight"] = (cout, cin, 3, 3)
        t[p + "bias"] = (cout,)
    return t


def _std_for(key, shape):
    if key.endswith(".bias"):
        if ".FC." in key:
            return 1.0 / np.sqrt(INNER)          # dynamic 32x32 filter entries ~ O(1/sqrt(32))
        return 0.05
    fan_in = int(np.prod(shape[1:]))
    if ".FC." in key:
        return 0.5 / np.sqrt(fan_in * INNER)
    if key.startswith("Encoder"):                # ReLU stacks: He scaling keeps activations O(1)
        return float(np.sqrt(2.0 / fan_in))
    return float(np.sqrt(1.0 / fan_in))


def synthetic_weights(seed=0):
    """Deterministic float32 weights for every key of :func:`weight_table`.

    Each tensor comes from its own PCG64 stream seeded by (seed, crc32(key)), so the
    values do not depend on generation order and are reproducible on the GPU box.
    """
    out = {}
    for key, shape in weight_table().items():
        rng = np.random.default_rng([int(seed), zlib.crc32(key.encode())])
        w = rng.standard_normal(shape, dtype=np.float32) * np.float32(_std_for(key, shape))
        out[key] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def weight_variant(name):
    """Named weight sets of the parity fixtures (tests/golden/make_goldens.py; all derived from the seeded generator,
    so only the name travels).  One draw is one draw: the variants change the dynamic range the kernels see.

    seed0 / seed1  two independent draws
    dec2 / dec4    seed 0 with every Decoder.* weight tensor (convs and FilterPredictor FCs, not the biases) x 2 / x 4:
                   larger dynamic filters (entries up to 24 at x 4), residuals and pre-clamp outputs.  At x 4 the saved
                   state is ill-conditioned in fp32 — the reference's own 1-thread and 8-thread runs differ by 30x the
                   state bound — so that fixture also carries the reference's float64 state (see tests/conftest.py)
    dead           seed 0 with dead / constant channels: 16 relu4_1 channels and 8 relu2_1 channels of both encoders
                   exactly zero (zero weights, bias -1 before the ReLU: variance 0, rstd = 1e4, style std = sqrt(1e-5)),
                   and 4 constant channels (zero weights, bias 0.3) out of Decoder.slice3.conv1 — the saved-statistics
                   normalisation of a channel without any spread
    """
    if name in ("seed0", "seed1"):
        return synthetic_weights(int(name[-1]))
    w = synthetic_weights(0)
    if name in ("dec2", "dec4"):
        for k in w:
            if k.startswith("Decoder.") and k.endswith(".weight"):
                w[k] = (w[k] * np.float32(name[-1])).astype(np.float32)
        return w
    if name == "dead":
        for pre, n in (("Encoder.slice.19.", 16), ("EncoderStyle.slice4.19.", 16), ("Encoder.slice.5.", 8), ("EncoderStyle.slice2.5.", 8)):
            w[pre + "weight"][:n] = 0.0
            w[pre + "bias"][:n] = -1.0
        w["Decoder.slice3.conv1.weight"][:4] = 0.0
        w["Decoder.slice3.conv1.bias"][:4] = 0.3
        return w
    raise ValueError("unknown weight variant %r" % (name,))


def load_checkpoint(path):
    """Read a reference ``.pth`` (torch.save'd state_dict) into the same {key: ndarray} form."""
    import torch  # plumbing only: unpickles the tensors

    sd = torch.load(path, map_location="cpu")
    table = weight_table()
    out = {}
    for key, shape in table.items():
        if key not in sd:
            raise KeyError("checkpoint %s lacks key %s" % (path, key))
        a = sd[key].detach().cpu().numpy().astype(np.float32, copy=False)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("checkpoint key %s has shape %s, expected %s" % (key, a.shape, shape))
        out[key] = np.ascontiguousarray(a)
    return out
