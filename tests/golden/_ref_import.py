"""Harness that imports the UNMODIFIED reference network from /root/reference
(authoring container only) under three tiny sys.modules stubs:

* ``torchvision.models.vgg19(pretrained=...)`` -> object whose ``.features`` is the
  standard cfg-E conv3x3/ReLU/MaxPool ``nn.Sequential`` (the reference only slices
  indices 0..20: test/style_network_global.py:241-253,275-278,288-302);
* ``cv2.cvtColor`` as a channel reversal (test/framework.py:27,42);
* empty ``kornia`` (only imported, never used, by test/style_network_frame.py:12).

Nothing in here travels to the GPU box; it is used only by make_goldens.py to
generate the committed fixtures.
"""
import sys
import types
import importlib

import numpy as np
import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


def _vgg19_features():
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
           512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


def install_stubs():
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvu = types.ModuleType("torchvision.utils")

        class _V:
            def __init__(self):
                self.features = _vgg19_features()

        tvm.vgg19 = lambda pretrained=False, **kw: _V()
        tv.models, tv.utils = tvm, tvu
        sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.utils": tvu})
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.COLOR_BGR2RGB, cv2.COLOR_RGB2BGR = 4, 4
        cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])
        sys.modules["cv2"] = cv2
    if "kornia" not in sys.modules:
        sys.modules["kornia"] = types.ModuleType("kornia")


def import_reference(subdir, *modules):
    """Import `modules` from /root/reference/<subdir> (fresh each call)."""
    install_stubs()
    path = "%s/%s" % (REF_ROOT, subdir)
    for name in ("framework", "style_network_global", "style_network_frame",
                 "style_network", "stylization"):
        sys.modules.pop(name, None)
    sys.path.insert(0, path)
    try:
        mods = [importlib.import_module(m) for m in modules]
        return mods[0] if len(mods) == 1 else mods
    finally:
        sys.path.remove(path)
