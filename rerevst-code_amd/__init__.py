"""MI355X-native per-frame stylization path of ReReVST (reference: daooshee/ReReVST-Code).

The product is the HIP library ``librerevst_hip.so`` (csrc/, C-ABI in include/rerevst_hip.h)
plus this thin host-side mirror of the reference's ``Stylization`` interface
(test/framework.py:56-118).  Importing the package is cheap; constructing ``Stylization``
loads the HIP library and fails loudly if it is missing.
"""
from .weights import weight_table, synthetic_weights, weight_variant, load_checkpoint  # noqa: F401
from .synth import synth_frame, synth_style  # noqa: F401


def __getattr__(name):
    if name in ("Stylization", "RRVError", "MultiStyleStylization", "ContentFeature", "pinned_empty"):
        from . import framework
        return getattr(framework, name)
    raise AttributeError(name)
