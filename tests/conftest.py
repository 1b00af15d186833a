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


