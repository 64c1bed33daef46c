"""The C-ABI library loads and exports every symbol include/microflow_amd.h declares.
CPU only: no compute entry point is called."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "microflow_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # drop comments
    names = re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("mf_model_create", "mf_model_predict", "mf_model_predict_quantized",
                 "mf_fully_connected_create", "mf_conv_2d_create", "mf_depthwise_conv_2d_create",
                 "mf_average_pool_2d_create", "mf_softmax_create", "mf_op_run", "mf_quantize",
                 "mf_dequantize", "mf_preprocess_fully_connected"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from microflow_rs_amd import _lib
    L = ctypes.CDLL(_lib.lib_path())
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    from microflow_rs_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    assert _lib.lib().mf_abi_version() == 3


def test_no_torch_types_cross_the_abi():
    text = open(os.path.join(ROOT, "include", "microflow_amd.h")).read()
    assert "torch" not in text and "at::" not in text and "#include <hip" not in text


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under microflow_rs_amd/ may import,
    link or execute it."""
    pkg = os.path.join(ROOT, "microflow_rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.replace("no CPU fallback", ""), os.path.join(dirpath, f)


def test_fails_loudly_without_gpu():
    import numpy as np
    import microflow_rs_amd as mf
    from microflow_rs_amd import _lib
    if _lib.lib().mf_device_count() > 0:
        pytest.skip("GPU present")
    m = mf.model(os.path.join(ROOT, "models", "sine.tflite"))
    with pytest.raises(mf.MicroflowError) as ei:
        m.predict(np.array([0.5], np.float32))
    assert ei.value.status == _lib.MF_ERR_NO_DEVICE
    h = ctypes.c_void_p()
    st = _lib.lib().mf_softmax_create(0, 1, 4, 0.1, 0.0039, -128, ctypes.byref(h))
    assert st == _lib.MF_ERR_NO_DEVICE
    assert b"no CPU fallback" in _lib.lib().mf_last_error()
