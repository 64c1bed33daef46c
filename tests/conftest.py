import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def samples():
    return dict(np.load(os.path.join(GOLDEN, "samples.npz")))


@pytest.fixture(scope="session")
def oracle_vectors():
    return dict(np.load(os.path.join(GOLDEN, "oracle_vectors.npz")))


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


def model_path(name):
    return os.path.join(MODELS, name + ".tflite")


# scripts/switch_matrix.sh runs the parity tests under every A/B switch of the library (MF_NO_QUAD=1, MF_NO_TABLE=1, ...).
# The assertions about WHICH kernel runs describe the default routing only and are skipped then; every parity assertion stays.
# Only the switches that change which kernel runs count: a stray debug variable (MF_VERBOSE, MF_DEBUG_EPI, MF_DQ_VERBOSE, ...)
# must not switch the routing assertions off.
# The library itself ignores every routing / tuning switch unless MF_DEV=1 is set (csrc/switches.cpp), and so does this list.
_NOT_ROUTING = ("MF_VERBOSE", "MF_DEBUG", "MF_DQ_VERBOSE", "MF_CHAIN_VERBOSE", "MF_STAGE_DIAG", "MF_TAIL3_DIAG", "MF_DEV", "MF_ALLOW_DIAG_BUILD",
                "MF_EXTRA_HIPCC_FLAGS", "MF_TIME_BATCH")
ROUTING_SWITCHED = (sorted(k for k in os.environ if k.startswith("MF_") and not k.startswith(_NOT_ROUTING))
                    if os.environ.get("MF_DEV", "")[:1] == "1" else [])
if ROUTING_SWITCHED:
    print("tests/conftest.py: kernel-routing assertions are SKIPPED because of %s" % ", ".join(ROUTING_SWITCHED), file=sys.stderr)
