"""Pins the CPU oracle (oracle/mf_oracle.c) against every known answer the
reference holds for the hot path.  CPU only."""
import csv
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN, model_path
from tests.synth import layer_checksum, synth_i8

ACT = {"none": 0, "relu": 1, "relu6": 3}
PAD = {"same": 0, "valid": 1}
f32 = np.float32


def test_quantize_value(O, kats):
    k = kats["quantize_value"]
    assert O.quantize(k["value"], k["scale"], k["zero_point"]) == k["quantized"]
    assert O.dequantize(k["quantized"], k["scale"], k["zero_point"]) == f32(k["dequantized"])


def test_activation(O, kats):
    k = kats["activation"]
    assert O.relu(k["relu_inactive"][0], k["zero_point"]) == k["relu_inactive"][1]
    assert O.relu(k["relu_active"][0], k["zero_point"]) == k["relu_active"][1]
    assert O.relu6(k["relu6_saturated"][0], k["scale"], k["zero_point"]) == k["relu6_saturated"][1]
    s = k["softmax_sum"]
    outs = [O.softmax_scalar(x, s, k["scale"], k["zero_point"]) for x in k["softmax_inputs"]]
    assert outs[0] == k["softmax_output_1"]
    assert sum(outs) == k["softmax_total"]


def test_roundf_half_away(O):
    for x, y in [(0.5, 1.0), (-0.5, -1.0), (1.5, 2.0), (2.5, 3.0), (-2.5, -3.0),
                 (0.49999997, 0.0), (-0.49999997, -0.0), (8388609.0, 8388609.0), (1e30, 1e30)]:
        assert O.roundf(x) == f32(y)


def test_expf_sanity(O):
    # the restated libm algorithm must stay within 1 ulp of a correctly rounded exp
    xs = np.linspace(-20, 20, 4001).astype(np.float32)
    for x in xs:
        got = f32(O.expf(x))
        ref = f32(np.exp(np.float64(x)))
        assert abs(np.float64(got) - np.float64(ref)) <= np.spacing(ref), x
    assert O.expf(0.0) == 1.0


def test_tensor_2d(O, kats):
    k = kats["tensor_2d"]
    q = O.quantize_array(k["buffer"], k["scale"], k["zero_point"])
    assert q.tolist() == k["quantized"]
    d = [[O.dequantize(v, k["scale"], k["zero_point"]) for v in row] for row in k["quantized"]]
    assert np.array_equal(np.array(d, f32), np.array(k["dequantized"], f32))


def test_tensor_4d(O, kats):
    k = kats["tensor_4d"]
    q = O.quantize_array(k["buffer"], k["scale"], k["zero_point"])
    assert q.tolist() == k["quantized"]
    d = np.array([O.dequantize(v, k["scale"], k["zero_point"]) for v in q.reshape(-1)], f32)
    assert np.array_equal(d.reshape(q.shape), np.array(k["buffer"], f32))
    v = k["view"]
    buf, mask, n = O.view(q[v["batch"]], v["focus"], v["shape"], PAD[v["padding"]], v["strides"])
    assert buf.tolist() == v["buffer"] and mask.tolist() == v["mask"] and n == v["len"]
    # 4D -> 2D flatten is logical NHWC order (src/tensor.rs:103-114): a reshape in row-major memory
    assert q.reshape(2, -1).tolist() == k["to_2d"]


def test_fully_connected_layer(O, kats):
    k = kats["fully_connected_layer"]
    w_nk = np.array(k["weights_kxn"], np.int8).T
    c = k["constants"]
    out = O.fully_connected(k["input"], w_nk, k["weights_zero_point"], k["output_scale"],
                            k["output_zero_point"], ACT[k["activation"]], c["c0"], c["c1"],
                            c["c2"], c["c3"])
    assert out.tolist() == k["output"]
    # and the constants themselves follow from preprocess()
    c0, c1, c2, c3 = O.preprocess_fully_connected(
        k["input_scale"], k["input_zero_point"], 3, w_nk, k["weights_scale"],
        k["weights_zero_point"], k["biases"], k["biases_scale"], k["biases_zero_point"],
        k["output_scale"])
    # (the KAT's float constants were not produced by preprocess(): compare to a few ulp only)
    assert np.allclose(c0, np.array(c["c0"], f32), rtol=1e-6) and np.isclose(c1, c["c1"], rtol=1e-6)
    assert c2.tolist() == c["c2"] and c3 == c["c3"]


def test_conv_2d_layer(O, kats):
    k = kats["conv_2d_layer"]
    c = k["constants"]
    out = O.conv_2d(k["input"], k["filters"], k["filters_zero_point"], k["input_zero_point"],
                    k["output_scale"], k["output_zero_point"], ACT[k["activation"]],
                    PAD[k["padding"]], k["strides"], (2, 3), c["c0"], c["c1"])
    assert out.tolist() == k["output"]
    c0, c1 = O.preprocess_conv(k["input_scale"], k["biases"], k["biases_scale"],
                               k["biases_zero_point"], k["filters_scale"], k["output_scale"])
    assert np.allclose(c0, np.array(c["c0"], f32), rtol=1e-6)
    assert np.allclose(c1, np.array(c["c1"], f32), rtol=1e-6)


def test_depthwise_conv_2d_layer(O, kats):
    k = kats["depthwise_conv_2d_layer"]
    c = k["constants"]
    out = O.depthwise_conv_2d(k["input"], k["weights"], k["weights_zero_point"],
                              k["input_zero_point"], k["output_scale"], k["output_zero_point"],
                              ACT[k["activation"]], PAD[k["padding"]], k["strides"], (2, 3),
                              c["c0"], c["c1"])
    assert out.tolist() == k["output"]
    c0, c1 = O.preprocess_conv(k["input_scale"], k["biases"], k["biases_scale"],
                               k["biases_zero_point"], k["weights_scale"], k["output_scale"])
    assert np.allclose(c0, np.array(c["c0"], f32), rtol=1e-6)
    assert np.allclose(c1, np.array(c["c1"], f32), rtol=1e-6)


def test_average_pool_2d_layer(O, kats):
    k = kats["average_pool_2d_layer"]
    c = k["constants"]
    out = O.average_pool_2d(k["input"], k["filter_shape"], k["output_scale"],
                            k["output_zero_point"], ACT[k["activation"]], PAD[k["padding"]],
                            k["strides"], (2, 3), c["c0"], c["c1"])
    assert out.tolist() == k["output"]
    c0, c1 = O.preprocess_average_pool_2d(k["input_scale"], k["input_zero_point"],
                                          k["output_scale"], k["output_zero_point"])
    assert np.isclose(c0, c["c0"], rtol=1e-6) and np.isclose(c1, c["c1"], rtol=1e-6)


def test_softmax_layer(O, kats):
    k = kats["softmax_layer"]
    out = O.softmax(k["input"], k["input_scale"], k["output_scale"], k["output_zero_point"])
    assert out.tolist() == k["output"]


def test_preprocess_kats(O, kats):
    k = kats["fully_connected_preprocess"]
    w_nk = np.array(k["weights_kxn"], np.int8).T
    c0, c1, c2, c3 = O.preprocess_fully_connected(
        k["input_scale"], k["input_zero_point"], k["input_shape"][1], w_nk, k["weights_scale"],
        k["weights_zero_point"], k["biases"], k["biases_scale"], k["biases_zero_point"],
        k["output_scale"])
    assert np.array_equal(c0, np.array(k["c0"], f32)) and c1 == f32(k["c1"])
    assert c2.tolist() == k["c2"] and c3 == k["c3"]
    for name, wkey in (("conv_2d_preprocess", "filters_scale"),
                       ("depthwise_conv_2d_preprocess", "weights_scale")):
        k = kats[name]
        c0, c1 = O.preprocess_conv(k["input_scale"], k["biases"], k["biases_scale"],
                                   k["biases_zero_point"], k[wkey], k["output_scale"])
        assert np.array_equal(c0, np.array(k["c0"], f32)), name
        assert np.array_equal(c1, np.array(k["c1"], f32)), name
    k = kats["average_pool_2d_preprocess"]
    c0, c1 = O.preprocess_average_pool_2d(k["input_scale"], k["input_zero_point"],
                                          k["output_scale"], k["output_zero_point"])
    assert c0 == f32(k["c0"]) and c1 == f32(k["c1"])


@pytest.mark.parametrize("name", ["sine", "speech", "person_detect"])
def test_whole_model_vectors(O, kats, name):
    k = kats[name + "_model"]
    m = O.Model(model_path(name))
    out = m.predict(np.full(m.in_elems, k["input_fill"], f32))
    assert np.array_equal(out.reshape(-1), np.array(k["output"], f32)), (out, k["output"])


def test_sine_500_recorded_outputs(O):
    m = O.Model(model_path("sine"))
    rows = list(csv.reader(open(os.path.join(GOLDEN, "sine_microflow.csv"))))[1:]
    assert len(rows) == 500
    bad = [(x, y) for x, y in rows if m.predict(np.array([f32(x)]))[0, 0] != f32(y)]
    assert not bad, bad[:5]


def test_model_structure(O):
    p = O.Model(model_path("person_detect"))
    kinds = [o["name"] for o in p.ops]
    assert kinds.count("depthwise_conv_2d") == 14 and kinds.count("conv_2d") == 14
    assert kinds.count("average_pool_2d") == 1 and kinds[-2:] == ["reshape", "softmax"]
    assert p.in_shape == (1, 96, 96, 1) and p.out_shape == (1, 2)
    s = O.Model(model_path("speech"))
    assert [o["name"] for o in s.ops] == ["reshape", "depthwise_conv_2d", "fully_connected",
                                          "softmax"]


def test_oracle_vectors_are_reproducible(O, oracle_vectors, samples):
    """The committed oracle-generated vectors must match what the oracle computes now."""
    for model in ("sine", "speech", "person_detect"):
        n, cfg = oracle_vectors[model + "_n"]
        m = O.Model(model_path(model))
        x = synth_i8(int(cfg), 0, int(n), m.in_elems)
        for i in range(min(int(n), 4)):
            out, layers = m.run_quantized(x[i], layers=True)
            assert np.array_equal(out, oracle_vectors[model + "_final"][i])
            sums = [layer_checksum(lay) for lay in layers]
            assert sums == oracle_vectors[model + "_layer_checksums"][i].tolist()
    pm = O.Model(model_path("person_detect"))
    sm = O.Model(model_path("speech"))
    for name, m in (("PERSON", pm), ("NO_PERSON", pm), ("YES", sm), ("NO", sm)):
        assert np.array_equal(m.run_quantized(samples[name]), oracle_vectors["sample_" + name])
    # survey appendix C: restatement-only values for the sample images
    assert oracle_vectors["sample_PERSON_f32"].reshape(-1).tolist() == [0.26953125, 0.73046875]
    assert oracle_vectors["sample_NO_PERSON_f32"].reshape(-1).tolist() == [0.6171875, 0.3828125]


def test_padding_quirk_is_pinned(O):
    """SAME padding uses shift (K-1)/2 for every stride (src/tensor.rs:193); for stride 2 on
    an even input this differs from TFLite.  The edge tap pattern below is the reference's."""
    x = np.arange(1, 17, dtype=np.int8).reshape(4, 4, 1)
    buf, mask, n = O.view(x, (0, 0), (3, 3), 0, (2, 2))
    assert mask.tolist() == [[False, False, False], [False, True, True], [False, True, True]]
    assert n == 4
    buf, mask, n = O.view(x, (1, 1), (3, 3), 0, (2, 2))   # rows 1..3, cols 1..3: all in range
    assert n == 9 and buf[..., 0].tolist() == [[6, 7, 8], [10, 11, 12], [14, 15, 16]]
