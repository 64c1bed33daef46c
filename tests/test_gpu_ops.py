"""Parity of every operator kernel against the reference's unit KATs and the CPU
oracle, called through the C ABI (via the Python mirror of microflow::ops).
Bit-exact: all comparisons are array_equal on int8."""
import os

import numpy as np
import pytest

from tests.conftest import ROOT, ROUTING_SWITCHED

pytestmark = pytest.mark.gpu

f32 = np.float32


@pytest.fixture(scope="module")
def mf():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import microflow_rs_amd as m
    assert m.lib().mf_device_count() > 0
    return m


ACT = {"none": 0, "relu": 1, "relu6": 3}
PAD = {"same": 0, "valid": 1}


def _acts(mf, name):
    return mf.FusedActivation(ACT[name])


# ---- the reference's own unit tests, through the reference-shaped API -------------
def test_fully_connected_layer(mf, kats):
    k = kats["fully_connected_layer"]
    c = k["constants"]
    out = mf.ops.fully_connected(
        mf.Tensor2D(np.array(k["input"], np.int8), [k["input_scale"]], [k["input_zero_point"]]),
        mf.Tensor2D(np.array(k["weights_kxn"], np.int8), [k["weights_scale"]], [k["weights_zero_point"]]),
        [k["output_scale"]], [k["output_zero_point"]],
        mf.ops.FullyConnectedOptions(_acts(mf, k["activation"])),
        (c["c0"], c["c1"], c["c2"], c["c3"]))
    assert out.buffer.tolist() == k["output"]
    assert out.scale == [k["output_scale"]] and out.zero_point == [k["output_zero_point"]]


def test_conv_2d_layer(mf, kats):
    k = kats["conv_2d_layer"]
    c = k["constants"]
    out = mf.ops.conv_2d(
        mf.Tensor4D(np.array(k["input"], np.int8)[None], [k["input_scale"]], [k["input_zero_point"]]),
        mf.Tensor4D(np.array(k["filters"], np.int8), k["filters_scale"], k["filters_zero_point"]),
        [k["output_scale"]], [k["output_zero_point"]],
        mf.ops.Conv2DOptions(_acts(mf, k["activation"]), mf.TensorViewPadding(PAD[k["padding"]]),
                             tuple(k["strides"])),
        (c["c0"], c["c1"]), (2, 3))
    assert out.buffer[0].tolist() == k["output"]   # includes saturation to 127


def test_depthwise_conv_2d_layer(mf, kats):
    k = kats["depthwise_conv_2d_layer"]
    c = k["constants"]
    out = mf.ops.depthwise_conv_2d(
        mf.Tensor4D(np.array(k["input"], np.int8)[None], [k["input_scale"]], [k["input_zero_point"]]),
        mf.Tensor4D(np.array(k["weights"], np.int8)[None], k["weights_scale"], k["weights_zero_point"]),
        [k["output_scale"]], [k["output_zero_point"]],
        mf.ops.DepthwiseConv2DOptions(_acts(mf, k["activation"]),
                                      mf.TensorViewPadding(PAD[k["padding"]]), tuple(k["strides"])),
        (c["c0"], c["c1"]), (2, 3))
    assert out.buffer[0].tolist() == k["output"]


def test_average_pool_2d_layer(mf, kats):
    k = kats["average_pool_2d_layer"]
    c = k["constants"]
    out = mf.ops.average_pool_2d(
        mf.Tensor4D(np.array(k["input"], np.int8)[None], [k["input_scale"]], [k["input_zero_point"]]),
        tuple(k["filter_shape"]), [k["output_scale"]], [k["output_zero_point"]],
        mf.ops.AveragePool2DOptions(_acts(mf, k["activation"]),
                                    mf.TensorViewPadding(PAD[k["padding"]]), tuple(k["strides"])),
        (c["c0"], c["c1"]), (2, 3))
    assert out.buffer[0].tolist() == k["output"]   # divides by the in-bounds count


def test_softmax_layer(mf, kats):
    k = kats["softmax_layer"]
    out = mf.ops.softmax(mf.Tensor2D(np.array(k["input"], np.int8), [k["input_scale"]],
                                     [k["input_zero_point"]]),
                         [k["output_scale"]], [k["output_zero_point"]])
    assert out.buffer.tolist() == k["output"]      # sum over the WHOLE 2x3 tensor


def test_reshape_layer(mf, kats):
    k = kats["reshape_layer"]
    out = mf.ops.reshape(mf.Tensor2D(np.array(k["input"], np.int8), [0.7], [8]), k["output_shape"])
    assert out.buffer.tolist() == k["output"] and isinstance(out, mf.Tensor4D)


def test_quantize_dequantize(mf, kats, O):
    k = kats["tensor_2d"]
    q = mf.ops.quantize(np.array(k["buffer"], f32), k["scale"], k["zero_point"])
    assert q.tolist() == k["quantized"]
    d = mf.ops.dequantize(np.array(k["quantized"], np.int8), k["scale"], k["zero_point"])
    assert np.array_equal(d, np.array(k["dequantized"], f32))
    k4 = kats["tensor_4d"]
    q = mf.ops.quantize(np.array(k4["buffer"], f32), k4["scale"], k4["zero_point"])
    assert q.tolist() == k4["quantized"]
    k = kats["quantize_value"]
    assert mf.ops.quantize(np.array([k["value"]], f32), k["scale"], k["zero_point"]).tolist() == [k["quantized"]]
    # random + adversarial values (ties, saturation, the pred(0.5) case, odd lengths)
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(0, 3, 4099).astype(f32),
                        np.array([0.5, -0.5, 1.5, 2.5, -2.5, 0.49999997, -0.49999997, 1e9, -1e9, 0.0], f32)])
    for scale, zp in ((0.0078431377, -1), (0.1, 3), (1.0, 0), (0.0235294122, -128)):
        got = mf.ops.quantize(x, scale, zp)
        want = O.quantize_array(x, scale, zp)
        assert np.array_equal(got, want), (scale, zp)
        qq = rng.integers(-128, 128, 1001).astype(np.int8)
        dg = mf.ops.dequantize(qq, scale, zp)
        dw = np.array([O.dequantize(v, scale, zp) for v in qq], f32)
        assert np.array_equal(dg.view(np.uint32), dw.view(np.uint32))


# ---- randomized parity vs the oracle ------------------------------------------------
def _rand_consts(rng, n, taps, per_channel=True):
    c0 = rng.uniform(-30, 30, n).astype(f32)
    s = 40.0 / (5476.0 * np.sqrt(taps))
    c1 = (rng.uniform(0.5, 1.5, n if per_channel else 1) * s).astype(f32)
    return c0, c1


DW_CASES = [
    # H, W, Cin, C, KH, KW, sh, sw, pad, OH, OW, wzp_nonzero, act
    (48, 48, 8, 8, 3, 3, 1, 1, 0, 48, 48, False, 3),
    (48, 48, 16, 16, 3, 3, 2, 2, 0, 24, 24, False, 3),
    (24, 24, 32, 32, 3, 3, 1, 1, 0, 24, 24, False, 3),
    (24, 24, 32, 32, 3, 3, 2, 2, 0, 12, 12, False, 1),
    (12, 12, 64, 64, 3, 3, 1, 1, 0, 12, 12, False, 3),
    (12, 12, 64, 64, 3, 3, 2, 2, 0, 6, 6, False, 0),
    (6, 6, 128, 128, 3, 3, 1, 1, 0, 6, 6, False, 3),
    (6, 6, 128, 128, 3, 3, 2, 2, 0, 3, 3, False, 3),
    (3, 3, 256, 256, 3, 3, 1, 1, 0, 3, 3, False, 3),
    (96, 96, 1, 8, 3, 3, 2, 2, 0, 48, 48, False, 3),     # stem
    (49, 40, 1, 8, 10, 8, 2, 2, 0, 25, 20, False, 1),    # speech op 1
    (13, 11, 1, 5, 5, 7, 1, 3, 0, 13, 4, False, 0),      # one input channel: odd filter, stride 3, N < 8
    (9, 9, 1, 8, 2, 2, 1, 1, 1, 8, 8, False, 3),         # one input channel, VALID, 2x2
    (10, 12, 1, 3, 3, 9, 2, 1, 0, 5, 12, False, 0),      # one input channel, 9-wide rows (3 dword groups)
    (30, 30, 1, 8, 1, 1, 1, 1, 0, 30, 30, False, 1),     # one input channel, 1x1 filter, > 512 pixels
    (7, 9, 3, 3, 3, 3, 1, 1, 0, 7, 9, True, 0),          # generic, non-zero weight zp
    (8, 8, 4, 4, 2, 3, 1, 2, 1, 7, 3, True, 3),          # VALID, rectangular
    (5, 6, 2, 6, 3, 3, 1, 1, 0, 5, 6, True, 0),          # channels beyond Cin read channel 0
    # C % 16 == 0 outside the 3x3 SAME stride-1/2 family: conv_mm_rt's depthwise mode (dw_mm_rt<KHxKW>)
    (24, 24, 32, 32, 5, 5, 1, 1, 0, 24, 24, False, 3),   # 5x5
    (20, 20, 16, 16, 3, 3, 1, 1, 1, 18, 18, False, 1),   # 3x3 VALID
    (17, 13, 48, 48, 5, 3, 2, 1, 0, 9, 13, False, 0),    # rectangular filter, unequal strides
    (12, 12, 64, 64, 7, 7, 2, 2, 0, 6, 6, False, 3),     # 7x7 stride 2
    (9, 11, 16, 16, 3, 3, 2, 1, 1, 4, 9, False, 3),      # 3x3 VALID, unequal strides
    (8, 8, 128, 128, 2, 2, 2, 2, 1, 4, 4, False, 0),     # 2x2 VALID stride 2
    (96, 96, 16, 16, 5, 5, 1, 1, 0, 96, 96, False, 3),   # an image too large for one tile: row bands
]
DW_MM_RT = {c for c in DW_CASES if c[2] == c[3] and c[2] % 16 == 0 and not c[11]
            and not (c[4] == 3 and c[5] == 3 and c[6] == c[7] and c[6] in (1, 2) and c[8] == 0)}


@pytest.mark.parametrize("case", DW_CASES, ids=lambda c: "x".join(map(str, c[:8])))
def test_depthwise_vs_oracle(mf, O, case):
    H, W, Cin, C, KH, KW, sh, sw, pad, OH, OW, wzp_nz, act = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    batch = 5
    x = rng.integers(-128, 128, (batch, H, W, Cin)).astype(np.int8)
    w = rng.integers(-128, 128, (KH, KW, C)).astype(np.int8)
    wzp = rng.integers(-20, 20, C).astype(np.int8) if wzp_nz else np.zeros(C, np.int8)
    izp = int(rng.integers(-128, 128))
    oscale, ozp = 0.0235294122, int(rng.integers(-128, 0))
    c0, c1 = _rand_consts(rng, C, KH * KW)
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), (sh, sw))
    op = mf.ops.prepare_depthwise_conv_2d((H, W, Cin), w, wzp, izp, oscale, ozp, opts, (c0, c1), (OH, OW))
    want = np.stack([O.depthwise_conv_2d(x[i], w, wzp, izp, oscale, ozp, act, pad, (sh, sw), (OH, OW),
                                         c0, c1) for i in range(batch)])
    got = op(x)
    assert np.array_equal(got, want), (op.kernel, np.argwhere(got != want)[:5])
    if case in DW_MM_RT:
        assert ROUTING_SWITCHED or op.kernel == "dw_mm_rt<%dx%d>" % (KH, KW), op.kernel
    if op.kernel != "dwconv_generic":  # the shape-generic kernel must agree as well
        fast_kernel = op.kernel
        op.set_generic(True)
        assert ROUTING_SWITCHED or op.kernel == "dwconv_generic"
        assert np.array_equal(op(x), want), fast_kernel


PW_CASES = [(48, 48, 8, 16), (24, 24, 16, 32), (24, 24, 32, 32), (12, 12, 32, 64), (12, 12, 64, 64),
            (6, 6, 64, 128), (6, 6, 128, 128), (3, 3, 128, 256), (3, 3, 256, 256), (1, 1, 256, 2)]


@pytest.mark.parametrize("case", PW_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("batch", [1, 3, 37])
def test_pointwise_conv_vs_oracle(mf, O, case, batch):
    H, W, C, N = case
    rng = np.random.default_rng(hash(case) % (2 ** 32) + batch)
    x = rng.integers(-128, 128, (batch, H, W, C)).astype(np.int8)
    f = rng.integers(-128, 128, (N, 1, 1, C)).astype(np.int8)
    fzp = np.zeros(N, np.int8)
    izp, oscale, ozp, act = -128, 0.0235294122, -128, 3
    c0, c1 = _rand_consts(rng, N, C)
    opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding.SAME, (1, 1))
    op = mf.ops.prepare_conv_2d((H, W, C), f, fzp, izp, oscale, ozp, opts, (c0, c1), (H, W))
    if N % 16 == 0:
        assert ROUTING_SWITCHED or op.kernel.startswith("pw_mfma"), op.kernel
    want = np.stack([O.conv_2d(x[i], f, fzp, izp, oscale, ozp, act, 0, (1, 1), (H, W), c0, c1)
                     for i in range(batch)])
    got = op(x)
    assert np.array_equal(got, want), (op.kernel, np.argwhere(got != want)[:5])
    op.set_generic(True)
    assert np.array_equal(op(x), want)


CONV_CASES = [
    # H, W, C, N, KH, KW, sh, sw, pad, OH, OW, per_channel_zp
    (6, 7, 3, 5, 3, 3, 1, 1, 0, 6, 7, True),
    (9, 9, 4, 8, 3, 3, 2, 2, 0, 5, 5, True),
    (8, 8, 2, 3, 2, 3, 1, 1, 1, 7, 6, False),
    (5, 5, 16, 16, 1, 1, 1, 1, 0, 5, 5, True),    # 1x1 but non-zero filter zp -> generic
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c[:8])))
def test_conv_generic_vs_oracle(mf, O, case):
    H, W, C, N, KH, KW, sh, sw, pad, OH, OW, pc = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    batch = 4
    x = rng.integers(-128, 128, (batch, H, W, C)).astype(np.int8)
    f = rng.integers(-128, 128, (N, KH, KW, C)).astype(np.int8)
    fzp = rng.integers(-30, 30, N if pc else 1).astype(np.int8)
    izp, oscale, ozp, act = int(rng.integers(-128, 128)), 0.05, int(rng.integers(-100, 100)), 1
    c0, c1 = _rand_consts(rng, N, KH * KW * C, per_channel=pc)
    opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), (sh, sw))
    op = mf.ops.prepare_conv_2d((H, W, C), f, fzp, izp, oscale, ozp, opts, (c0, c1), (OH, OW))
    # few input channels with image rows in whole dwords: conv_rows_lds; a 1x1 filter with weight zero points: pw_rt
    # (k_rt.hip); 3 channels x 7 columns = 21-byte rows: the shape-generic kernel
    want_kernel = "pw_rt<16,16,wzp>" if KH * KW == 1 else ("conv_rows_lds<wzp>" if (W * C) % 4 == 0 else "conv2d_generic")
    assert ROUTING_SWITCHED or op.kernel == want_kernel, op.kernel
    want = np.stack([O.conv_2d(x[i], f, fzp, izp, oscale, ozp, act, pad, (sh, sw), (OH, OW), c0, c1)
                     for i in range(batch)])
    assert np.array_equal(op(x), want)
    op.set_generic(True)
    assert np.array_equal(op(x), want)


@pytest.mark.parametrize("case", [(3, 3, 256, 3, 3, 2, 2, 1, 1, 1), (6, 8, 5, 2, 3, 1, 1, 0, 6, 8),
                                  (7, 7, 4, 3, 3, 2, 2, 0, 4, 4)],
                         ids=lambda c: "x".join(map(str, c)))
def test_average_pool_vs_oracle(mf, O, case):
    H, W, C, FH, FW, sh, sw, pad, OH, OW = case
    rng = np.random.default_rng(7)
    batch = 6
    x = rng.integers(-128, 128, (batch, H, W, C)).astype(np.int8)
    iscale, izp, oscale, ozp = 0.0235294122, -128, 0.0186093301, -128
    c0, c1 = O.preprocess_average_pool_2d(iscale, izp, oscale, ozp)
    for act in (0, 1, 3):
        opts = mf.ops.AveragePool2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), (sh, sw))
        op = mf.ops.prepare_average_pool_2d((H, W, C), (FH, FW), oscale, ozp, opts, (c0, c1), (OH, OW))
        want = np.stack([O.average_pool_2d(x[i], (FH, FW), oscale, ozp, act, pad, (sh, sw), (OH, OW), c0, c1)
                         for i in range(batch)])
        assert np.array_equal(op(x), want), act


@pytest.mark.parametrize("case", [(1, 1, 16, 0), (1, 16, 16, 0), (1, 16, 1, 5), (1, 4000, 4, 0),
                                  (3, 37, 11, -7), (2, 512, 8, 3), (1, 4096, 2, 0)],
                         ids=lambda c: "x".join(map(str, c)))
def test_fully_connected_vs_oracle(mf, O, case):
    M, K, N, wzp = case
    rng = np.random.default_rng(K * 131 + N)
    batch = 9
    x = rng.integers(-128, 128, (batch, M, K)).astype(np.int8)
    w = rng.integers(-128, 128, (N, K)).astype(np.int8)
    bias = rng.integers(-2000, 2000, N).astype(np.int32)
    iscale, izp, wscale, oscale, ozp = 0.05, int(rng.integers(-128, 128)), 0.01, 0.05 * np.sqrt(K), 14
    c0, c1, c2, c3 = O.preprocess_fully_connected(iscale, izp, K, w, wscale, wzp, bias, iscale * wscale, 0, oscale)
    for act in (0, 1):
        op = mf.ops.prepare_fully_connected(M, w, wzp, oscale, ozp, mf.ops.FullyConnectedOptions(mf.FusedActivation(act)),
                                            (c0, c1, c2, c3))
        want = np.stack([O.fully_connected(x[i], w, wzp, oscale, ozp, act, c0, c1, c2, c3) for i in range(batch)])
        got = op(x)
        assert np.array_equal(got, want), (op.kernel, act)
        if op.kernel != "fc_generic":
            op.set_generic(True)
            assert np.array_equal(op(x), want)


def test_softmax_vs_oracle(mf, O):
    rng = np.random.default_rng(3)
    for rows, cols, iscale, oscale, ozp in ((1, 4, 0.0917319208, 1 / 256, -128), (1, 2, 0.0125187514, 1 / 256, -128),
                                            (2, 3, 0.7, 0.9, 10), (1, 10, 0.3, 1 / 256, -128)):
        x = rng.integers(-128, 128, (300, rows, cols)).astype(np.int8)
        op = mf.ops.prepare_softmax(rows, cols, iscale, oscale, ozp)
        want = np.stack([O.softmax(x[i], iscale, oscale, ozp) for i in range(x.shape[0])])
        assert np.array_equal(op(x), want), (rows, cols)


def test_softmax_every_int8_pair(mf, O):
    """All 65536 (a, b) int8 pairs for person_detect's 2-class softmax against the oracle: exhaustive parity, every pair."""
    a, b = np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), indexing="ij")
    x = np.stack([a.reshape(-1), b.reshape(-1)], -1).astype(np.int8).reshape(-1, 1, 2)
    op = mf.ops.prepare_softmax(1, 2, 0.0125187514, 1 / 256, -128)
    got = op(x)
    want = np.stack([O.softmax(x[i], 0.0125187514, 1 / 256, -128) for i in range(x.shape[0])])
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]


def test_softmax_speech_four_classes_every_value_per_position(mf, O):
    """speech's 4-class softmax (src/ops/softmax.rs:20-27; input scale of speech.tflite's FullyConnected output): every one of
    the 256 int8 values in each of the four positions against a stride of backgrounds for the other three (4 x 256 x 125 rows),
    plus all 4^4 combinations of the extreme and middle values."""
    iscale, oscale, ozp = 0.0917319208, 1 / 256, -128
    bg = np.array([[b0, b1, b2] for b0 in (-128, -77, -3, 40, 127) for b1 in (-128, -50, 0, 64, 127) for b2 in (-128, -20, 5, 90, 127)], np.int16)
    rows = []
    for pos in range(4):
        for v in range(-128, 128):
            r = np.insert(bg, pos, v, axis=1)
            rows.append(r)
    ext = np.array([-128, -1, 0, 127])
    g = np.stack(np.meshgrid(ext, ext, ext, ext, indexing="ij"), -1).reshape(-1, 4)
    x = np.concatenate(rows + [g]).astype(np.int8).reshape(-1, 1, 4)
    op = mf.ops.prepare_softmax(1, 4, iscale, oscale, ozp)
    got = op(x)
    want = np.stack([O.softmax(x[i], iscale, oscale, ozp) for i in range(x.shape[0])])
    assert x.shape[0] == 4 * 256 * 125 + 256 and np.array_equal(got, want), np.argwhere(got != want)[:5]


def test_device_tensors_stay_on_device(mf):
    import torch
    x = torch.randint(-128, 128, (4, 6, 6, 128), dtype=torch.int8, device="cuda")
    w = np.ones((3, 3, 128), np.int8)
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation.RELU6, mf.TensorViewPadding.SAME, (1, 1))
    op = mf.ops.prepare_depthwise_conv_2d((6, 6, 128), w, np.zeros(1, np.int8), -128, 0.02, -128, opts,
                                          (np.zeros(128, f32), np.full(1, 0.01, f32)), (6, 6))
    y = op(x)
    assert isinstance(y, torch.Tensor) and y.is_cuda and y.shape == (4, 6, 6, 128)


@pytest.mark.parametrize("case", [(128, 128, 128, 0), (256, 512, 384, 0), (128, 256, 128, 9), (384, 128, 256, -5)],
                         ids=lambda c: "x".join(map(str, c)))
def test_fully_connected_mfma_gemm_vs_oracle(mf, O, case):
    """The int8 MFMA GEMM path (rows = whole 128-row tiles), incl. non-zero weight zero point."""
    M, K, N, wzp = case
    rng = np.random.default_rng(M + K + N)
    x = rng.integers(-128, 128, (M, K)).astype(np.int8)
    w = rng.integers(-128, 128, (N, K)).astype(np.int8)     # asymmetric operands: a transposed
    bias = rng.integers(-5000, 5000, N).astype(np.int32)    # write-back cannot pass
    iscale, izp, wscale = 1 / 128, -128, 1 / 128
    oscale, ozp = float(iscale * wscale * np.sqrt(K) * 74 * 74 * 3 / 127), 3
    c0, c1, c2, c3 = O.preprocess_fully_connected(iscale, izp, K, w, wscale, wzp, bias, iscale * wscale, 0, oscale)
    for act in (0, 1):
        op = mf.ops.prepare_fully_connected(M, w, wzp, oscale, ozp, mf.ops.FullyConnectedOptions(mf.FusedActivation(act)),
                                            (c0, c1, c2, c3))
        assert ROUTING_SWITCHED or op.kernel == "fc_mfma"
        want = O.fully_connected(x, w, wzp, oscale, ozp, act, c0, c1, c2, c3)
        got = op(x)
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]
        assert len(np.unique(got)) > 50          # the outputs really spread over the int8 range
        # M rows per inference x batch of 2 -> the same tiles, twice
        got2 = op(np.stack([x, x[::-1]]))
        assert np.array_equal(got2[0], want) and np.array_equal(got2[1], want[::-1])
        op.set_generic(True)
        assert np.array_equal(op(x), want)
    # a row count that is not a multiple of 128 falls back to the generic kernel, same results
    op = mf.ops.prepare_fully_connected(100, w, wzp, oscale, ozp, mf.ops.FullyConnectedOptions(), (c0, c1, c2, c3))
    assert np.array_equal(op(x[:100]), O.fully_connected(x[:100], w, wzp, oscale, ozp, 0, c0, c1, c2, c3))


_TILE256_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import microflow_rs_amd as mf
from oracle import oracle as O
for (M, K, N, wzp) in [(256, 128, 256, 0), (256, 256, 256, 0), (512, 384, 256, 7), (256, 640, 512, 0), (768, 1024, 256, -3)]:
    rng = np.random.default_rng(M + K + N)
    x = rng.integers(-128, 128, (M, K)).astype(np.int8)
    w = rng.integers(-128, 128, (N, K)).astype(np.int8)
    bias = rng.integers(-5000, 5000, N).astype(np.int32)
    oscale = float(np.sqrt(K) * 74 * 74 * 3 / 127 / 16384)
    c = O.preprocess_fully_connected(1 / 128, -128, K, w, 1 / 128, wzp, bias, 1 / 16384, 0, oscale)
    op = mf.ops.prepare_fully_connected(M, w, wzp, oscale, 3, mf.ops.FullyConnectedOptions(), c)
    assert op.kernel == "fc_mfma", op.kernel
    want = O.fully_connected(x, w, wzp, oscale, 3, 0, *c)
    for rep in range(5):                       # repeated launches: a staging race would not be stable
        got = op(x)
        assert np.array_equal(got, want), (M, K, N, rep, np.argwhere(got != want)[:4])
print("tile256 ok")
"""


def test_fully_connected_mfma_256_tile_schedule():
    """The 256x256-tile kernel (staggered wave rows) is normally chosen only for >= 192 tiles;
    MF_FC_TILE=256 forces it so that K = 1, 2, 3, 5, 8 staging steps are all checked against
    the oracle (first/last-tile paths of the software pipeline)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MF_DEV="1", MF_FC_TILE="256")
    r = subprocess.run([sys.executable, "-c", _TILE256_SCRIPT.format(root=root)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "tile256 ok" in r.stdout, r.stdout + r.stderr


def test_pointwise_accumulator_extremes(mf, O):
    """Worst-case accumulators for the bit-pattern int->float conversion of the fast kernels
    (|acc| < 2^22 is what the host checks): K = 128, |w| = 128, v - izp = 255 gives
    |acc| = 4 177 920, just inside; the outputs must still equal the oracle's."""
    H, W, C, N = 6, 6, 128, 128
    rng = np.random.default_rng(77)
    f = np.where(rng.integers(0, 2, (N, 1, 1, C)) > 0, 127, -128).astype(np.int8)
    f[0] = -128                                   # channel 0: every weight -128
    f[1] = 127                                    # channel 1: every weight +127
    x = rng.integers(-128, 128, (4, H, W, C)).astype(np.int8)
    x[0] = 127                                    # v - izp = 255 everywhere -> extreme sums
    x[1, :, :, :] = np.where(f[2, 0, 0] > 0, 127, -128)   # aligned with channel 2's signs
    fzp = np.zeros(N, np.int8)
    izp, oscale, ozp, act = -128, 0.05, 0, 0
    c0 = rng.uniform(-3, 3, N).astype(f32)
    c1 = np.full(N, 100.0 / 4177920.0, f32) * rng.uniform(0.9, 1.1, N).astype(f32)
    opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding.SAME, (1, 1))
    op = mf.ops.prepare_conv_2d((H, W, C), f, fzp, izp, oscale, ozp, opts, (c0, c1), (H, W))
    assert ROUTING_SWITCHED or op.kernel.startswith("pw_mfma"), op.kernel
    want = np.stack([O.conv_2d(v, f, fzp, izp, oscale, ozp, act, 0, (1, 1), (H, W), c0, c1) for v in x])
    got = op(x)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    assert want[0, 0, 0, 0] < -90 and want[0, 0, 0, 1] > 90   # the extremes are not clamped away
    op.set_generic(True)
    assert np.array_equal(op(x), want)


_NO_MAGIC_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import microflow_rs_amd as mf
from oracle import oracle as O
from tests.synth import synth_i8
for name, cfg, n in (("person_detect", 3, 5), ("speech", 2, 9)):
    path = {root!r} + "/models/" + name + ".tflite"
    m, om = mf.model(path), O.Model(path)
    x = synth_i8(cfg, 0, n, m.input_elems)
    assert np.array_equal(m.run_quantized(x).reshape(n, -1), om.run_quantized_batch(x)), name
    m.set_fusion(False)
    assert np.array_equal(m.run_quantized(x).reshape(n, -1), om.run_quantized_batch(x)), name
print("no-magic ok")
"""


def test_models_without_magic_accumulators():
    """MF_NO_MAGIC=1 forces the v_cvt_f32_i32 form of every fast kernel (the variant used when
    an operator's worst-case accumulator could reach 2^22): same results."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MF_DEV="1", MF_NO_MAGIC="1")
    r = subprocess.run([sys.executable, "-c", _NO_MAGIC_SCRIPT.format(root=root)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "no-magic ok" in r.stdout, r.stdout + r.stderr


def test_quantize_every_float_near_the_int8_range(mf):
    """Device roundf + saturating cast, exhaustively: EVERY f32 with 2^-3 <= |x| < 2^9 (11 binades
    x 2^23 mantissas x 2 signs = 185 M values, plus the denormal-to-0.125 range sampled) through
    mf_quantize(scale 1, zp 0) against round-half-away-from-zero + saturation evaluated in f64."""
    import torch
    for e in range(124, 136):                      # biased exponents: 2^-3 .. 2^8
        bits = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(e << 23))
        for sign in (0, 0x80000000):
            x = (bits | np.uint32(sign)).view(f32)
            got = mf.ops.quantize(torch.as_tensor(x).cuda(), 1.0, 0).cpu().numpy()
            x64 = x.astype(np.float64)
            want = np.clip(np.trunc(x64 + np.copysign(0.5, x64)), -128, 127).astype(np.int8)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (e, sign, x[bad[:4]], got[bad[:4]], want[bad[:4]])
    small = np.arange(0, 124 << 23, 4099, dtype=np.uint32).view(f32)   # [0, 0.125): all round to 0
    assert not mf.ops.quantize(torch.as_tensor(small).cuda(), 1.0, 0).cpu().numpy().any()
    special = np.array([np.nan, np.inf, -np.inf, 3.4e38, -3.4e38, -0.0], f32)
    assert mf.ops.quantize(special, 1.0, 0).tolist() == [0, 127, -128, 127, -128, 0]


@pytest.mark.parametrize("c1,c0s", [(2.0 ** -6, [0.5, -0.5, 3.25, -7.5, 0.0, 100.5, -100.5, 0.49999997]),
                                    (0.0123456, [0.1, -0.9, 17.3, -40.2, 0.5, 126.9, -127.4, 1e-3])],
                         ids=["exact-ties", "generic"])
def test_requantize_dense_accumulator_sweep(mf, O, c1, c0s):
    """The f32 epilogue over a DENSE accumulator range: FullyConnected with K = 2, weights
    (1, 127) and every (x0, x1) pair gives every integer acc in [-16384, 16256]; with c1 = 2^-6
    and half-integer c0 the sum lands on exact .5 ties for both signs (round half away from
    zero), and the clamp saturates at both ends."""
    x = np.stack(np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), indexing="ij"), -1).reshape(-1, 2).astype(np.int8)
    w = np.tile(np.array([[1, 127]], np.int8), (8, 1))
    c0 = np.array(c0s, f32)
    c2, c3 = np.zeros(8, np.int32), 0
    for act, ozp in ((0, 0), (1, -3), (3, -128)):
        op = mf.ops.prepare_fully_connected(x.shape[0], w, 0, 0.05, ozp, mf.ops.FullyConnectedOptions(mf.FusedActivation(act)),
                                            (c0, c1, c2, c3))
        want = O.fully_connected(x, w, 0, 0.05, ozp, act, c0, c1, c2, c3)
        got = op(x)
        assert np.array_equal(got, want), (act, np.argwhere(got != want)[:5])
    assert len(np.unique(want)) > 100


def test_requantize_dense_sweep_fast_pointwise(mf, O):
    """The same dense accumulator sweep through the MFMA pointwise kernel, whose epilogue uses
    the bit-pattern int->f32 conversion (requant_t<true>): a 256x256x8 'image' whose pixels hold
    every (x0, x1) pair, filters (1, 127, 0...) -> acc = x0 + 127 x1, exact-tie constants."""
    H = W = 256
    px = np.stack(np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), indexing="ij"), -1).reshape(H, W, 2)
    x = np.zeros((1, H, W, 8), np.int8)
    x[0, :, :, :2] = px
    f = np.zeros((16, 1, 1, 8), np.int8)
    f[:, 0, 0, 0], f[:, 0, 0, 1] = 1, 127
    f[8:, 0, 0, 1] = -127                                        # second half: acc = x0 - 127 x1
    c0 = np.array([0.5, -0.5, 3.25, -7.5, 0.0, 100.5, -100.5, 0.49999997] * 2, f32)
    c1 = np.array([2.0 ** -6] * 4 + [0.0123456] * 4 + [2.0 ** -7] * 4 + [0.05] * 4, f32)
    for act, ozp in ((0, 0), (3, -128)):
        opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding.SAME, (1, 1))
        op = mf.ops.prepare_conv_2d((H, W, 8), f, np.zeros(16, np.int8), 0, 0.05, ozp, opts, (c0, c1), (H, W))
        assert ROUTING_SWITCHED or op.kernel == "pw_mfma<8,16>", op.kernel
        want = O.conv_2d(x[0], f, np.zeros(16, np.int8), 0, 0.05, ozp, act, 0, (1, 1), (H, W), c0, c1)
        got = op(x)[0]
        assert np.array_equal(got, want), (act, np.argwhere(got != want)[:5])


@pytest.mark.parametrize("case", [(1, 1, 256, 2, True), (3, 3, 64, 8, True), (2, 5, 12, 3, False), (4, 4, 4, 1, True)],
                         ids=lambda c: "x".join(map(str, c[:4])))
def test_conv_1x1_few_outputs_rowwave(mf, O, case):
    """Conv2D 1x1 with N <= 8 outputs (the head conv of person_detect: 256 -> 2) on its row-wave kernel: non-zero
    filter zero points (the value-sum term), per-channel and per-tensor constants, a NaN constant, ragged batch."""
    H, W, C, N, pc = case
    rng = np.random.default_rng(1000 + C + N)
    batch = 37
    x = rng.integers(-128, 128, (batch, H, W, C)).astype(np.int8)
    f = rng.integers(-128, 128, (N, 1, 1, C)).astype(np.int8)
    fzp = rng.integers(-30, 30, N if pc else 1).astype(np.int8)
    izp, oscale, ozp, act = int(rng.integers(-128, 128)), 0.05, int(rng.integers(-100, 100)), int(rng.integers(0, 2))
    c0, c1 = _rand_consts(rng, N, C, per_channel=pc)
    if N >= 3:
        c0[1] = np.nan
    opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding.SAME, (1, 1))
    op = mf.ops.prepare_conv_2d((H, W, C), f, fzp, izp, oscale, ozp, opts, (c0, c1), (H, W))
    assert ROUTING_SWITCHED or op.kernel == "conv1x1_rowwave"
    want = np.stack([O.conv_2d(x[i], f, fzp, izp, oscale, ozp, act, 0, (1, 1), (H, W), c0, c1) for i in range(batch)])
    assert np.array_equal(op(x), want)


def test_non_finite_constants_follow_rust_casts(mf, O):
    """Degenerate models (NaN / Inf constants): Rust's `as i8` maps NaN to 0 and saturates +-Inf,
    and the activation is applied afterwards.  Such operators are kept on the shape-generic
    kernels, which reproduce that; the result equals the oracle's for conv, depthwise and FC."""
    rng = np.random.default_rng(91)
    H, W, C, N = 6, 6, 32, 32                      # a shape that normally takes the MFMA pointwise kernel
    x = rng.integers(-128, 128, (2, H, W, C)).astype(np.int8)
    f = rng.integers(-128, 128, (N, 1, 1, C)).astype(np.int8)
    c0, c1 = _rand_consts(rng, N, C)
    c0[[1, 5]] = [np.nan, -np.inf]
    c1[[2, 7, 9]] = [np.nan, np.inf, -np.inf]
    for act, ozp in ((0, 5), (1, -20), (3, -128)):
        opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding.SAME, (1, 1))
        op = mf.ops.prepare_conv_2d((H, W, C), f, np.zeros(N, np.int8), -128, 0.0235294122, ozp, opts, (c0, c1), (H, W))
        assert ROUTING_SWITCHED or op.kernel == "conv2d_generic"
        want = np.stack([O.conv_2d(v, f, np.zeros(N, np.int8), -128, 0.0235294122, ozp, act, 0, (1, 1), (H, W), c0, c1) for v in x])
        assert np.array_equal(op(x), want), act
    w = rng.integers(-128, 128, (3, 3, C)).astype(np.int8)
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(1), mf.TensorViewPadding.SAME, (1, 1))
    op = mf.ops.prepare_depthwise_conv_2d((H, W, C), w, np.zeros(C, np.int8), -128, 0.05, -7, opts, (c0, c1), (H, W))
    assert ROUTING_SWITCHED or op.kernel == "dwconv_generic"
    want = np.stack([O.depthwise_conv_2d(v, w, np.zeros(C, np.int8), -128, 0.05, -7, 1, 0, (1, 1), (H, W), c0, c1) for v in x])
    assert np.array_equal(op(x), want)
    xf = rng.integers(-128, 128, (128, 256)).astype(np.int8)
    wf = rng.integers(-128, 128, (128, 256)).astype(np.int8)
    fc0 = rng.uniform(-5, 5, 128).astype(f32)
    fc0[3] = np.nan
    for fc1 in (np.float32(np.nan), np.float32(np.inf), np.float32(1e-4)):
        op = mf.ops.prepare_fully_connected(128, wf, 0, 0.05, 3, mf.ops.FullyConnectedOptions(), (fc0, fc1, np.zeros(128, np.int32), 0))
        assert ROUTING_SWITCHED or op.kernel == "fc_generic"
        want = O.fully_connected(xf, wf, 0, 0.05, 3, 0, fc0, fc1, np.zeros(128, np.int32), 0)
        assert np.array_equal(op(xf), want)


# ---- the epilogue forms of the fast kernels (k_common.hpp) -----------------------------
def test_epilogue_rounding_identity_exhaustive(mf):
    """roundf(x) (libm, half away from zero) == round-to-nearest-even of x with its lowest mantissa bit set, followed
    by the clamp / `as T`: checked on the device over EVERY float bit pattern in the form's domain, for the v_med3 form
    (mode 1, several clamps, both element types) and the saturating-pack form (mode 2, whole range).  Mode 3 is the
    negative control -- the same without the mantissa bit -- and must report the ties it gets wrong."""
    import ctypes as C
    from microflow_rs_amd import _lib
    L = _lib.lib()
    bad = C.c_uint64(77)
    for mode, u8, lo, hi in ((1, 0, -128, 127), (1, 0, -128, 100), (1, 0, -7, 127), (1, 0, 3, 3), (1, 1, 0, 255),
                             (1, 1, 128, 200), (2, 0, -128, 127), (2, 1, 0, 255)):
        _lib.check(L.mf_selftest_rounding(0, mode, u8, lo, hi, C.byref(bad)))
        assert bad.value == 0, (mode, u8, lo, hi, bad.value)
    _lib.check(L.mf_selftest_rounding(0, 3, 0, -128, 127, C.byref(bad)))
    assert bad.value >= 255, bad.value     # every tie k + 0.5 with k even and positive (or odd and negative), at least
    assert L.mf_selftest_rounding(0, 2, 0, -128, 100, C.byref(bad)) == _lib.MF_ERR_INVALID_ARG  # mode 2: whole range only


def test_epilogue_requant_every_accumulator(mf, O):
    """The whole epilogue as the kernels call it (requant_pack4, modes 1 and 2) against the plain form for every
    accumulator in (-2^22, 2^22), at the (A, S) of person_detect's channels with the most rounding ties, at extreme
    scales, and at constants that put x exactly on ties."""
    import ctypes as C
    from microflow_rs_amd import _lib
    L = _lib.lib()
    bad = C.c_uint64(77)
    cases = [(-127.5, 0.5), (-128.0, 0.25), (0.5, 1.0), (-3.25, 0.001953125), (-127.31, 0.0021), (12.7, 3.1e-5),
             (-100.0, 0.31964308), (-128.0, 0.00053977553), (5.0, 0.007), (-64.03125, 0.015625)]
    from tests.conftest import model_path
    om = O.Model(model_path("person_detect"))
    for i in (1, 2, 12, 24, 26):
        c0, c1 = om.op_constants(i)[:2]
        ozp = om.ops[i]["out_zp"]
        for c in (0, len(c0) // 2, len(c0) - 1):
            cases.append((float(f32(f32(ozp) + f32(c0[c]))), float(c1[c if c < len(c1) else 0])))
    for A, S in cases:
        for mode, u8, lo, hi in ((1, 0, -128, 127), (1, 0, -128, 90), (2, 0, -128, 127), (1, 1, 0, 255), (2, 1, 0, 255)):
            a = A + (128.0 if u8 else 0.0)
            _lib.check(L.mf_selftest_requant(0, mode, u8, C.c_float(a), C.c_float(S), lo, hi, C.byref(bad)))
            assert bad.value == 0, (A, S, mode, u8, lo, hi, bad.value)


_NO_SAT_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import microflow_rs_amd as mf
from oracle import oracle as O
from tests.synth import synth_i8
for name, cfg, n in (("person_detect", 3, 5), ("speech", 2, 9)):
    path = {root!r} + "/models/" + name + ".tflite"
    m, om = mf.model(path), O.Model(path)
    x = synth_i8(cfg, 0, n, m.input_elems)
    assert np.array_equal(m.run_quantized(x).reshape(n, -1), om.run_quantized_batch(x)), name
    m.set_fusion(False)
    assert np.array_equal(m.run_quantized(x).reshape(n, -1), om.run_quantized_batch(x)), name
print("no-sat ok")
"""


def test_models_without_saturating_pack():
    """MF_NO_SAT_PACK=1 keeps every fast kernel on the v_med3 form of the epilogue (mode 1; the form used whenever an
    operator's clamp is narrower than its element type, e.g. relu6 with a small output scale): same results."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MF_DEV="1", MF_NO_SAT_PACK="1")
    r = subprocess.run([sys.executable, "-c", _NO_SAT_SCRIPT.format(root=root)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "no-sat ok" in r.stdout, r.stdout + r.stderr


# ---- FullyConnected GEMM: every row, ragged row counts, both tiles --------------------------
# (4096^3, 5000 x 4096 and 8192 x 8192 run the weight-zero-point term from row sums formed in the GEMM launch's own prologue --
# fc_mfma<..., RSP> --; 8192 x 8192 is 1024 tiles = four residency waves, where an epilogue may find its producers not started
# and sums its rows itself; the others use the fc_rowsum pre-pass launch)
@pytest.mark.parametrize("M,K,N", [(4096, 4096, 4096), (4099, 1024, 512), (193, 256, 384), (64, 128, 128), (1000, 640, 768),
                                   (5000, 384, 4096), (8192, 256, 8192)])
def test_fc_mfma_equals_generic_kernel_on_every_row(mf, O, M, K, N):
    """The MFMA GEMM against the shape-generic FullyConnected kernel over the WHOLE output matrix (the oracle checks of
    the 4096^3 test sample 128 rows), including row counts that are not a multiple of the 128- / 256-row tile
    (src/ops/fully_connected.rs:42-81 has no such restriction): the ragged last tile re-reads the last row and stores
    nothing for the rows past it.  A few rows also go through the oracle."""
    import torch
    rng = np.random.default_rng(M * 31 + K)
    x = rng.integers(-128, 128, (M, K)).astype(np.int8)
    w = rng.integers(-128, 128, (N, K)).astype(np.int8)
    c0 = rng.uniform(-3, 3, N).astype(f32)
    c1 = f32(127.0 / (74 * 74 * 3 * np.sqrt(K)))
    wzp, izp = -5, -128
    c2 = (izp * w.astype(np.int64).sum(axis=1)).astype(np.int32)
    c3 = int(K * izp * wzp)
    op = mf.ops.prepare_fully_connected(M, w, wzp, 0.05, 3, mf.ops.FullyConnectedOptions(), (c0, c1, c2, c3))
    assert ROUTING_SWITCHED or op.kernel == "fc_mfma", op.kernel
    xd = torch.as_tensor(x).cuda()
    guard = torch.full((M * N + 4096,), 0x5A, dtype=torch.int8, device="cuda")   # the ragged tile must not write past row M-1
    got = op(xd).clone()
    op.set_generic(True)
    assert ROUTING_SWITCHED or op.kernel == "fc_generic"
    want = op(xd)
    assert torch.equal(got, want), int((got != want).sum())
    rows = sorted({0, 1, M // 2, M - 2, M - 1})
    ref = O.fully_connected(x[rows], w, wzp, 0.05, 3, 0, c0, c1, c2, c3)
    assert np.array_equal(got.cpu().numpy().reshape(M, N)[rows], ref)
    del guard


@pytest.mark.parametrize("switch", ["MF_FC_ROWSUM_FOLD", "MF_FC_ROWSUM_PREPASS"])
def test_fc_mfma_row_sums_inside_the_gemm(switch):
    """The two other ways of forming the weight-zero-point term's row sums -- MF_FC_ROWSUM_FOLD=1: v_dot4 between the GEMM's MFMAs
    (fc_mfma<..., RS>, both tile sizes, ragged M); MF_FC_ROWSUM_PREPASS=1: the fc_rowsum launch in front of the GEMM everywhere --
    the whole output against the shape-generic kernel.  A child process: the switches are read once per process."""
    import subprocess
    import sys as _sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import microflow_rs_amd as mf
f32 = np.float32
for M, K, N in ((4096, 2048, 4096), (1000, 640, 768), (193, 256, 384)):
    rng = np.random.default_rng(M + K)
    x = rng.integers(-128, 128, (M, K)).astype(np.int8)
    w = rng.integers(-128, 128, (N, K)).astype(np.int8)
    c0 = rng.uniform(-3, 3, N).astype(f32)
    c1 = f32(127.0 / (74 * 74 * 3 * np.sqrt(K)))
    wzp, izp = 7, -128
    c2 = (izp * w.astype(np.int64).sum(axis=1)).astype(np.int32)
    op = mf.ops.prepare_fully_connected(M, w, wzp, 0.05, 3, mf.ops.FullyConnectedOptions(), (c0, c1, c2, int(K * izp * wzp)))
    assert op.kernel == "fc_mfma", op.kernel
    xd = torch.as_tensor(x).cuda()
    got = op(xd).clone()
    op.set_generic(True)
    want = op(xd)
    assert torch.equal(got, want), (M, K, N, int((got != want).sum()))
print("FOLD_OK")
''' % ROOT
    env = dict(os.environ, MF_DEV="1", **{switch: "1"})
    r = subprocess.run([_sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "FOLD_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
