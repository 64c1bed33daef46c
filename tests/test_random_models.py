"""Randomized model-level parity: small generated .tflite networks that touch every operator
the reference supports (microflow-macros/src/lib.rs:138-148) with shapes and options the three
reference models never use -- K x K Conv2D with strides and VALID padding, rectangular
depthwise filters, per-channel / per-tensor / non-zero weight zero points, both tails
(Reshape + FullyConnected + Softmax and Conv2D head + Reshape + Softmax), both element types.
CPU: the library's parser + constant preparation against the oracle's (two independent
FlatBuffers readers, C++ vs C).  GPU: every layer of every model, bit for bit."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tflite_writer as tw  # noqa: E402
from tests.conftest import ROUTING_SWITCHED  # noqa: E402

CASES = [(seed, elem, pc, wz, tail)
         for seed, (elem, pc, wz, tail) in enumerate([
             (tw.INT8, True, False, "fc"), (tw.INT8, True, True, "conv"), (tw.INT8, False, True, "fc"),
             (tw.INT8, False, False, "conv"), (tw.UINT8, True, True, "fc"), (tw.UINT8, False, False, "conv"),
             (tw.INT8, True, True, "fc"), (tw.UINT8, True, False, "conv"), (tw.INT8, True, False, "conv"),
             (tw.UINT8, False, True, "fc")])]


def _ids(c):
    return "s%d-%s-%s-%s-%s" % (c[0], "u8" if c[1] == tw.UINT8 else "i8", "perch" if c[2] else "pert",
                                 "wzp" if c[3] else "wz0", c[4])


def _blob(case):
    seed, elem, pc, wz, tail = case
    return tw.random_cnn(np.random.default_rng(1000 + seed), elem, pc, wz, tail)


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_random_model_parse_matches_oracle(O, case):
    mf = importlib.import_module("microflow_rs_amd")
    blob = _blob(case)
    pm, om = mf.Model(blob), O.Model(blob)
    assert pm.dtype == om.dtype and pm.num_ops == om.num_ops >= 6
    assert (pm.input_shape, pm.output_shape) == (om.in_shape, om.out_shape)
    for i in range(pm.num_ops):
        d, o = pm.op(i), om.ops[i]
        for k in ("kind", "in_shape", "out_shape", "KH", "KW", "sh", "sw", "pad", "act", "n_c0", "n_c1", "in_zp", "out_zp"):
            assert d[k] == o[k], (i, k, d[k], o[k])
        a, b = pm.op_constants(i), om.op_constants(i)
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(np.asarray(x).view(np.int32), np.asarray(y).view(np.int32)), (i, d["name"])
        assert a[3] == b[3]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_random_model_runs_bit_exact(O, case):
    mf = importlib.import_module("microflow_rs_amd")
    blob = _blob(case)
    m, om = mf.Model(blob), O.Model(blob)
    rng = np.random.default_rng(case[0])
    lo, hi = (0, 256) if m.dtype == np.uint8 else (-128, 128)
    n = 7
    xq = rng.integers(lo, hi, (n, m.input_elems)).astype(m.dtype)
    want = om.run_quantized_batch(xq)
    got = m.run_quantized(xq).reshape(n, -1)
    if not np.array_equal(got, want):                      # localise the first differing layer
        _, layers = om.run_quantized(xq[0], layers=True)
        for i, lay in enumerate(layers):
            g = np.asarray(m.run_until(xq[:1], i)).reshape(-1)
            assert np.array_equal(g, lay.reshape(-1)), (i, m.op(i)["name"], m.op(i)["kernel"])
    assert np.array_equal(got, want)
    pf = m.predict_quantized(xq).reshape(n, -1)
    wf = np.stack([om.predict_quantized(x).reshape(-1) for x in xq])
    assert np.array_equal(pf.view(np.uint32), wf.view(np.uint32))


SPEECH_CASES = [(0, tw.INT8, 0, True, "relu"), (1, tw.INT8, 5, True, "relu"), (2, tw.UINT8, -3, False, "relu6"),
                (3, tw.INT8, 0, False, "none"), (4, tw.UINT8, 0, True, "relu")]


@pytest.mark.parametrize("case", SPEECH_CASES, ids=lambda c: "s%d-%s-wzp%d" % (c[0], "u8" if c[1] == tw.UINT8 else "i8", c[2]))
def test_speech_like_parse_matches_oracle(O, case):
    mf = importlib.import_module("microflow_rs_amd")
    blob = tw.speech_like(np.random.default_rng(2000 + case[0]), *case[1:])
    pm, om = mf.Model(blob), O.Model(blob)
    assert pm.num_ops == om.num_ops == 4 and pm.input_elems == 1960 and pm.output_elems == 4
    for i in range(pm.num_ops):
        a, b = pm.op_constants(i), om.op_constants(i)
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(np.asarray(x).view(np.int32), np.asarray(y).view(np.int32)), i
        assert a[3] == b[3]


@pytest.mark.gpu
@pytest.mark.parametrize("case", SPEECH_CASES, ids=lambda c: "s%d-%s-wzp%d" % (c[0], "u8" if c[1] == tw.UINT8 else "i8", c[2]))
def test_speech_like_one_launch_bit_exact(O, case):
    """The speech-shaped network as ONE launch (k_dwfc.hip: depthwise taps on the matrix pipe -> FullyConnected ->
    Softmax), against the oracle and against the operator-by-operator path, for batch sizes around the 16 images of
    a workgroup step; incl. the FullyConnected weight zero point (row-sum term) speech.tflite itself never has."""
    mf = importlib.import_module("microflow_rs_amd")
    blob = tw.speech_like(np.random.default_rng(2000 + case[0]), *case[1:])
    m, om = mf.Model(blob), O.Model(blob)
    m.prepare(1)
    names = [m.op(i)["kernel"] for i in range(m.num_ops)]
    if not ROUTING_SWITCHED:
        assert names[1].startswith("dwc1_fc_softmax"), names
    rng = np.random.default_rng(case[0])
    lo, hi = (0, 256) if m.dtype == np.uint8 else (-128, 128)
    for n in (1, 15, 16, 17, 33, 200):
        xq = rng.integers(lo, hi, (n, m.input_elems)).astype(m.dtype)
        if n == 17:
            xq[3], xq[4] = lo, hi - 1  # constant extremes
        got = m.run_quantized(xq).reshape(n, -1)
        m.set_fusion(False)
        layerwise = m.run_quantized(xq).reshape(n, -1)
        m.set_fusion(True)
        assert np.array_equal(got, layerwise), n
        k = min(n, 24)
        assert np.array_equal(got[:k], om.run_quantized_batch(xq[:k])), n


# ---- person_detect's architecture at other sizes: the run-time-geometry kernels end to end ------------------------
PD_LIKE = [
    # side, width, pairs in the middle run, element type, weight zero points
    (64, 1.0, 5, tw.INT8, False),     # 32x32x8 ... 4x4x128 x5 ... 2x2x256
    (128, 1.0, 5, tw.INT8, False),    # 64x64x8 (row bands) ... 8x8x128 x5 ... 4x4x256
    (96, 0.5, 5, tw.INT8, False),     # width 0.5: 4, 8, 16, 32, 64, 128 channels
    (96, 0.75, 5, tw.INT8, False),    # width 0.75: 8, 12, 24, 48, 96, 192 channels
    (96, 1.0, 6, tw.INT8, False),     # the shipped shapes with SIX 6x6x128 pairs: one stage kernel over all six
    (96, 1.0, 2, tw.INT8, False),     # ... and with two
    (64, 1.0, 3, tw.UINT8, False),
    (80, 1.0, 4, tw.INT8, True),      # weight zero points everywhere (40x40x8 ... 5x5x128 ... 3x3x256)
    (72, 0.5, 5, tw.UINT8, True),
]


def _pd_id(c):
    return "%dx%d-w%s-n%d-%s-%s" % (c[0], c[0], c[1], c[2], "u8" if c[3] == tw.UINT8 else "i8", "wzp" if c[4] else "wz0")


@pytest.mark.gpu
@pytest.mark.parametrize("case", PD_LIKE, ids=_pd_id)
def test_person_detect_like_models_run_on_fast_kernels(O, case):
    """tools/tflite_writer.person_detect_like: person_detect's layer structure at input sizes / widths / run lengths the
    table kernels were not compiled for.  Bit-exact against the oracle at EVERY layer, fused and layer-wise, and (weight
    zero points == 0) no operator on a shape-generic `*_generic` kernel."""
    mf = importlib.import_module("microflow_rs_amd")
    side, width, nst, elem, wz = case
    blob = tw.person_detect_like(np.random.default_rng(side * 7 + nst), side, width, elem, wz, nst)
    m, om = mf.Model(blob), O.Model(blob)
    rng = np.random.default_rng(side)
    lo, hi = (0, 256) if m.dtype == np.uint8 else (-128, 128)
    n = 6
    xq = rng.integers(lo, hi, (n, m.input_elems)).astype(m.dtype)
    xq[0], xq[1] = hi - 1, lo
    want = om.run_quantized_batch(xq)
    for fusion in (True, False):
        m.set_fusion(fusion)
        got = m.run_quantized(xq).reshape(n, -1)
        if not np.array_equal(got, want):                  # localise the first differing layer
            _, layers = om.run_quantized(xq[2], layers=True)
            for i, lay in enumerate(layers):
                g = np.asarray(m.run_until(xq[2:3], i)).reshape(-1)
                assert np.array_equal(g, lay.reshape(-1)), (fusion, i, m.op(i)["name"], m.op(i)["kernel"])
        assert np.array_equal(got, want), fusion
    # every reference tensor, layer by layer
    _, layers = om.run_quantized(xq[3], layers=True)
    for i in (0, 1, 2, 3, 8, 13, 14, len(layers) - 5, len(layers) - 1):
        assert np.array_equal(np.asarray(m.run_until(xq[3:4], i)).reshape(-1), layers[i].reshape(-1)), (i, m.op(i)["kernel"])
    if ROUTING_SWITCHED:   # the assertions below describe the default routing (scripts/switch_matrix.sh: parity under every switch)
        return
    m.set_fusion(False)
    kernels = [m.op(i)["kernel"] for i in range(m.num_ops)]
    if not wz:
        assert not [k for k in kernels if k.endswith("_generic")], kernels
        assert any(k.startswith("dw3x3_rt") for k in kernels) or side == 96, kernels
    else:
        assert any(k.endswith(",wzp>") for k in kernels), kernels
    m.set_fusion(True)
    fused = [m.op(i)["kernel"] for i in range(m.num_ops)]
    if side == 96 and width == 1.0 and not wz:
        assert ("stage_6x6x128<4,512,%d>" % nst) in fused, fused
