import os
import sys
import importlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from state_bounds import (PRE_ATOL, PRE_RTOL, IMG_ATOL, STATE_RTOL, STATE_ATOL, NORM_CH, NORM_NAMES, GOLDEN,   # noqa: F401,E402
                          load_golden, decode_png, golden_inputs, state_worst, assert_state_close, assert_pre_close,
                          pre_worst, pre_full_size, img_full_size, assert_state_close_conditioned, assert_state_close_two_refs)


# The library's default (rrv_set_f43 mode 1) picks F(4x4,3x3) or F(2x2,3x3) per layer from the launch geometry (frames per
# launch, frame size, CUs a launch may use), so a frame's low-order bits depend on how it was submitted.  Every test that
# compares with the oracle or a reference golden runs in that DEFAULT — it is what bench.py times.  Only the cross-entry
# invariants (batched == one frame per call == tickets == pad/crop entry ..., bit for bit) need a FIXED kernel choice and ask
# for it with `fixed_kernels(...)` below.  RRV_F43=0 / 2 in the environment still runs the whole suite on one kernel family.
import contextlib


@contextlib.contextmanager
def fixed_kernels(*handles, mode=0):
    """Inside the block every handle given — and every handle created meanwhile, through RRV_F43 — runs ONE kernel family
    (mode 0: F(2x2,3x3) everywhere, 2: conv_f43_k on every packed layer): the precondition of the bit-identity invariants."""
    env = os.environ.get("RRV_F43")
    if env in ("0", "2"):
        mode = int(env)                 # a whole-suite run on one family keeps it
    os.environ["RRV_F43"] = str(mode)
    for h in handles:
        h.set_f43(mode)
    try:
        yield
    finally:
        if env is None:
            del os.environ["RRV_F43"]
        else:
            os.environ["RRV_F43"] = env
        for h in handles:
            if getattr(h, "_h", None):
                h.set_f43(int(env) if env is not None else 1)


# Under RRV_F43=2 five multi-style tests are skipped: they assert that two ENTRIES agree to 1e-3 grey levels or bit for bit
# (blend transfer of a frame vs decoder on its cached feature; grouped per-image-state launches vs one frame per launch), and
# the cached features / the per-image-state kernels are always F(2x2,3x3) while the other side of each comparison then runs
# F(4x4,3x3) — the two differ by the kernels' rounding (<= 0.03 grey levels, inside the parity bound of 0.05), not by a defect.
_F43_MODE2_SKIPS = ("test_config5_full_size_1024_four_styles_vs_oracle", "test_multistyle_batched_transfer_equals_per_frame",
                    "test_random_sequence_of_multistyle_entries_is_bit_exact", "test_multistyle_feature_api_matches_reference",
                    "test_real_multistyle_matches_reference",
                    # round 6: the one-frame caching entry is F(2x2,3x3) in every mode, the batched one follows the mode
                    "test_batched_feature_caching_equals_per_frame", "test_multistyle_command_line_driver_end_to_end")
# a suite forced onto ONE family cannot show that the DEFAULT picks another one per launch / per conditioning
_FORCED_FAMILY_SKIPS = ("test_ill_conditioned_state_keeps_f43_off_the_encoder",)


def forced_family():
    """RRV_F43=0 / 2 in the environment: the whole suite runs on one kernel family (tests then drop the assertions that the
    default choice differs from a pinned family)."""
    return os.environ.get("RRV_F43") in ("0", "2")


def pytest_collection_modifyitems(config, items):
    if not forced_family():
        return
    skip2 = pytest.mark.skip(reason="multi-style cross-entry comparison: cached features and per-image-state kernels are F(2x2,3x3) in every mode")
    skipf = pytest.mark.skip(reason="needs the default kernel choice (RRV_F43 forces one family)")
    for it in items:
        name = it.name.split("[")[0]
        if os.environ.get("RRV_F43") == "2" and name in _F43_MODE2_SKIPS:
            it.add_marker(skip2)
        if name in _FORCED_FAMILY_SKIPS:
            it.add_marker(skipf)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("rerevst-code_amd")


@pytest.fixture(scope="session")
def oracle():
    import rerevst_oracle
    return rerevst_oracle


@pytest.fixture(scope="session")
def weights(pkg):
    return pkg.synthetic_weights(0)


