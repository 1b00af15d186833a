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
                          pre_worst, assert_state_close_conditioned, assert_state_close_two_refs)


# The suite's cross-entry invariants (batched == one frame per call == tickets == ..., bit for bit) hold for a FIXED kernel
# choice; the library's default picks F(4x4,3x3) by the number of frames per launch (rrv_set_f43).  Everything but
# tests/test_gpu_f43.py therefore runs with F(2x2,3x3) pinned; RRV_F43=2 in the environment runs the whole suite on the
# F(4x4,3x3) kernels instead (profiles/r04_gpu_suite_f43_mode2.txt).
os.environ.setdefault("RRV_F43", "0")


# Under RRV_F43=2 five multi-style tests are skipped: they assert that two ENTRIES agree to 1e-3 grey levels or bit for bit
# (blend transfer of a frame vs decoder on its cached feature; grouped per-image-state launches vs one frame per launch), and
# the cached features / the per-image-state kernels are always F(2x2,3x3) while the other side of each comparison then runs
# F(4x4,3x3) — the two differ by the kernels' rounding (<= 0.03 grey levels, inside the parity bound of 0.05), not by a defect.
_F43_MODE2_SKIPS = ("test_config5_full_size_1024_four_styles_vs_oracle", "test_multistyle_batched_transfer_equals_per_frame",
                    "test_random_sequence_of_multistyle_entries_is_bit_exact", "test_multistyle_feature_api_matches_reference",
                    "test_real_multistyle_matches_reference")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("RRV_F43") != "2":
        return
    skip = pytest.mark.skip(reason="multi-style cross-entry comparison: cached features and per-image-state kernels are F(2x2,3x3) in every mode")
    for it in items:
        if it.name.split("[")[0] in _F43_MODE2_SKIPS:
            it.add_marker(skip)


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


