"""Element type u8 (the reference's `T = u8` instantiation, src/quantize.rs:43-53,
microflow-macros/src/lib.rs:118-128).

The reference holds no u8 known-answer test, so the u8 oracle is anchored indirectly:
  * oracle/mf_oracle_ops.inc is ONE restatement instantiated for int8_t and uint8_t; the
    int8_t instantiation is the one every reference KAT pins (tests/test_oracle_golden.py);
  * a second, independent numpy restatement below is first checked against the pinned i8
    oracle and then against the u8 oracle on the same randomized cases.
GPU tests then compare the HIP path with the u8 oracle bit for bit.
"""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_fc_model import synthetic_fc  # noqa: E402
from make_u8_model import to_u8  # noqa: E402
from tests.conftest import ROUTING_SWITCHED  # noqa: E402

f32 = np.float32
DTYPES = [np.int8, np.uint8]


def _range(dt):
    return (0, 256) if dt == np.uint8 else (-128, 128)


def _rand_q(rng, dt, shape):
    lo, hi = _range(dt)
    return rng.integers(lo, hi, shape).astype(dt)


def _rand_consts(rng, n, taps, per_channel=True):
    c0 = rng.uniform(-30, 30, n).astype(f32)
    s = 40.0 / (5476.0 * np.sqrt(taps))
    c1 = (rng.uniform(0.5, 1.5, n if per_channel else 1) * s).astype(f32)
    return c0, c1


# ---- independent numpy restatement (element-type agnostic) --------------------------------
def _roundf(x):  # half away from zero on f32 scalars / arrays
    x = np.asarray(x, f32)
    return np.where(np.abs(x) < 8388608.0, np.trunc(x + np.copysign(f32(0.49999997), x)), x).astype(f32)


def _sat(x, dt):
    lo, hi = _range(dt)
    x = np.nan_to_num(np.asarray(x, f32), nan=0.0)
    return np.clip(np.trunc(x), lo, hi - 1).astype(np.int64)


def _act(y, act, oscale, ozp, dt):
    if act in (1, 3):
        y = np.maximum(y, ozp)
    if act == 3:
        six = _sat(_roundf(f32(f32(6.0) / f32(oscale)) + f32(ozp)), dt)
        y = np.minimum(y, six)
    return y


def np_conv(x, w, wzp, izp, oscale, ozp, act, pad, strides, out_hw, c0, c1, depthwise):
    """x [H][W][Cin]; conv w [N][KH][KW][C], depthwise w [KH][KW][N].  Per-output view with
    zero fill + mask, the three integer sums, then the f32 epilogue."""
    dt = x.dtype.type
    H, W, Cin = x.shape
    if depthwise:
        KH, KW, N = w.shape
    else:
        N, KH, KW, _ = w.shape
    OH, OW = out_hw
    sh, sw = strides
    py, px = ((KH - 1) // 2, (KW - 1) // 2) if pad == 0 else (0, 0)
    out = np.zeros((OH, OW, N), np.int64)
    xi, wi = x.astype(np.int64), w.astype(np.int64)
    for oy in range(OH):
        for ox in range(OW):
            for n in range(N):
                z = int(wzp[n] if n < len(wzp) else wzp[0])
                acc = 0
                for ky in range(KH):
                    for kx in range(KW):
                        iy, ix = oy * sh + ky - py, ox * sw + kx - px
                        if not (0 <= iy < H and 0 <= ix < W):
                            continue
                        if depthwise:
                            v = xi[iy, ix, n if n < Cin else 0]
                            acc += (int(v) - izp) * (int(wi[ky, kx, n]) - z)
                        else:
                            acc += int(((xi[iy, ix] - izp) * (wi[n, ky, kx] - z)).sum())
                a = f32(f32(ozp) + f32(c0[n]))
                b = f32(f32(c1[n] if n < len(c1) else c1[0]) * f32(acc))
                out[oy, ox, n] = _sat(_roundf(f32(a + b)), dt)
    return _act(out, act, oscale, ozp, dt).astype(dt)


def np_fc(x, w_nk, wzp, oscale, ozp, act, c0, c1, c2, c3):
    dt = x.dtype.type
    xi, wi = x.astype(np.int64), w_nk.astype(np.int64)
    acc = xi @ wi.T - int(wzp) * xi.sum(axis=1, keepdims=True) - np.asarray(c2, np.int64)[None, :] + int(c3)
    a = (f32(ozp) + np.asarray(c0, f32)).astype(f32)
    b = (f32(c1) * acc.astype(f32)).astype(f32)
    return _act(_sat(_roundf((a[None, :] + b).astype(f32)), dt), act, oscale, ozp, dt).astype(dt)


def np_pool(x, fshape, oscale, ozp, act, pad, strides, out_hw, c0, c1):
    dt = x.dtype.type
    H, W, C_ = x.shape
    KH, KW = fshape
    OH, OW = out_hw
    py, px = ((KH - 1) // 2, (KW - 1) // 2) if pad == 0 else (0, 0)
    out = np.zeros((OH, OW, C_), np.int64)
    for oy in range(OH):
        for ox in range(OW):
            ys = [oy * strides[0] + k - py for k in range(KH)]
            xs = [ox * strides[1] + k - px for k in range(KW)]
            ys = [v for v in ys if 0 <= v < H]
            xs = [v for v in xs if 0 <= v < W]
            s = x[np.ix_(ys, xs)].astype(np.int64).sum(axis=(0, 1))
            inv = f32(f32(1.0) / f32(len(ys) * len(xs)))
            m = (inv * s.astype(f32)).astype(f32)
            y = ((f32(c0) * m).astype(f32) + f32(c1)).astype(f32)
            out[oy, ox] = _sat(_roundf(y), dt)
    return _act(out, act, oscale, ozp, dt).astype(dt)


CONV_CASES = [
    # H, W, C, N, KH, KW, sh, sw, pad, OH, OW, act
    (5, 6, 3, 4, 3, 3, 1, 1, 0, 5, 6, 3),
    (7, 7, 2, 5, 3, 3, 2, 2, 0, 4, 4, 1),
    (6, 5, 4, 3, 2, 3, 1, 1, 1, 5, 3, 0),
    (4, 4, 8, 16, 1, 1, 1, 1, 0, 4, 4, 3),
]
DW_CASES = [
    # H, W, Cin, C, KH, KW, sh, sw, pad, OH, OW, act
    (6, 6, 4, 4, 3, 3, 1, 1, 0, 6, 6, 3),
    (7, 5, 3, 3, 3, 3, 2, 2, 0, 4, 3, 1),
    (6, 7, 1, 5, 4, 3, 1, 2, 1, 3, 3, 0),
    (5, 6, 2, 6, 3, 3, 1, 1, 0, 5, 6, 0),
]
POOL_CASES = [(3, 3, 16, 3, 3, 2, 2, 1, 1, 1, 3), (6, 8, 5, 2, 3, 1, 1, 0, 6, 8, 0), (6, 6, 3, 2, 2, 2, 2, 1, 3, 3, 1)]
FC_CASES = [(1, 16, 16, 3), (3, 40, 7, 0), (2, 64, 4, 1)]


def _conv_inputs(rng, dt, case, depthwise):
    H, W, Cin, N, KH, KW, sh, sw, pad, OH, OW, act = case
    lo, hi = _range(dt)
    x = _rand_q(rng, dt, (H, W, Cin))
    w = _rand_q(rng, dt, (KH, KW, N) if depthwise else (N, KH, KW, Cin))
    mid = (lo + hi) // 2
    wzp = rng.integers(mid - 20, mid + 20, N).astype(dt)
    izp = int(rng.integers(lo, hi))
    oscale, ozp = 0.0235294122, int(rng.integers(lo, lo + 128))
    c0, c1 = _rand_consts(rng, N, KH * KW * (1 if depthwise else Cin))
    return x, w, wzp, izp, oscale, ozp, act, pad, (sh, sw), (OH, OW), c0, c1


def _fc_inputs(rng, dt, case, O):
    M, K, N, act = case
    lo, hi = _range(dt)
    x = _rand_q(rng, dt, (M, K))
    w = _rand_q(rng, dt, (N, K))
    izp, wzp = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    bias = rng.integers(-2000, 2000, N).astype(np.int32)
    oscale, ozp = 0.05, int(rng.integers(lo, lo + 128))
    consts = O.preprocess_fully_connected(0.02, izp, K, w, 0.01 * 40 / np.sqrt(K), wzp, bias, 0.0002, 0, oscale)
    return x, w, wzp, oscale, ozp, act, consts


# ---- CPU: the oracle's two instantiations against the independent restatement -------------
@pytest.mark.parametrize("dt", DTYPES, ids=["i8", "u8"])
def test_oracle_conv_matches_numpy_restatement(O, dt):
    rng = np.random.default_rng(11)
    for case in CONV_CASES:
        a = _conv_inputs(rng, dt, case, False)
        assert np.array_equal(O.conv_2d(*a), np_conv(*a, depthwise=False)), case
    for case in DW_CASES:
        a = _conv_inputs(rng, dt, case, True)
        assert np.array_equal(O.depthwise_conv_2d(*a), np_conv(*a, depthwise=True)), case


@pytest.mark.parametrize("dt", DTYPES, ids=["i8", "u8"])
def test_oracle_fc_pool_match_numpy_restatement(O, dt):
    rng = np.random.default_rng(12)
    lo, hi = _range(dt)
    for case in FC_CASES:
        x, w, wzp, oscale, ozp, act, (c0, c1, c2, c3) = _fc_inputs(rng, dt, case, O)
        got = O.fully_connected(x, w, wzp, oscale, ozp, act, c0, c1, c2, c3)
        assert got.dtype == dt
        assert np.array_equal(got, np_fc(x, w, wzp, oscale, ozp, act, c0, c1, c2, c3)), case
    for case in POOL_CASES:
        H, W, C_, FH, FW, sh, sw, pad, OH, OW, act = case
        x = _rand_q(rng, dt, (H, W, C_))
        izp, ozp = int(rng.integers(lo, hi)), int(rng.integers(lo, lo + 100))
        c0, c1 = O.preprocess_average_pool_2d(0.05, izp, 0.04, ozp) if dt == np.int8 else \
            (f32(f32(0.05) / f32(0.04)), f32(f32(ozp) - f32(f32(f32(0.05) * f32(izp)) / f32(0.04))))
        got = O.average_pool_2d(x, (FH, FW), 0.04, ozp, act, pad, (sh, sw), (OH, OW), c0, c1)
        assert np.array_equal(got, np_pool(x, (FH, FW), 0.04, ozp, act, pad, (sh, sw), (OH, OW), c0, c1)), case


@pytest.mark.parametrize("dt", DTYPES, ids=["i8", "u8"])
def test_oracle_fc_preprocess_sums(O, dt):
    """c2[j] = izp * sum_k w[j][k], c3 = shape1 * izp * wzp with T's own signedness."""
    rng = np.random.default_rng(13)
    lo, hi = _range(dt)
    w = _rand_q(rng, dt, (5, 24))
    izp, wzp = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    c0, c1, c2, c3 = O.preprocess_fully_connected(0.02, izp, 24, w, 0.01, wzp, np.arange(5, dtype=np.int32), 0.0002, 0, 0.05)
    assert np.array_equal(c2, (w.astype(np.int64).sum(axis=1) * izp).astype(np.int32))
    assert c3 == 24 * izp * wzp


def test_oracle_scalar_primitives_u8(O):
    # quantize saturates to [0, 255]; relu / relu6 clamp in the u8 domain
    assert O.quantize(-3.0, 0.1, 5, np.uint8) == 0
    assert O.quantize(1e9, 0.1, 5, np.uint8) == 255
    assert O.quantize(0.25, 0.1, 5, np.uint8) == 8      # 2.5 + 5 = 7.5 -> 8 (half away from zero)
    assert O.dequantize(200, 0.5, 100, np.uint8) == f32(50.0)
    assert O.relu(3, 10, np.uint8) == 10 and O.relu(200, 10, np.uint8) == 200
    assert O.relu6(250, 0.05, 10, np.uint8) == 130      # quantize(6.0) = 120 + 10


def _u8_model_bytes(name):
    with open(os.path.join(ROOT, "models", name + ".tflite"), "rb") as f:
        return to_u8(f.read())


@pytest.mark.parametrize("name", ["sine", "speech", "person_detect"])
def test_u8_model_parse_matches_oracle(O, name):
    """Host-side parse + preprocess of a u8 model (no GPU): metadata and constants bit-equal."""
    mf = importlib.import_module("microflow_rs_amd")
    data = _u8_model_bytes(name)
    pm, om = mf.Model(data), O.Model(data)
    assert pm.dtype == np.uint8 and om.dtype == np.uint8
    assert (pm.input_zero_point, pm.output_zero_point) == (om.in_zp, om.out_zp)
    assert 0 <= pm.input_zero_point <= 255 and 0 <= pm.output_zero_point <= 255
    assert pm.num_ops == om.num_ops
    for i in range(pm.num_ops):
        d, o = pm.op(i), om.ops[i]
        assert (d["in_zp"], d["out_zp"]) == (o["in_zp"], o["out_zp"])
        a, b = pm.op_constants(i), om.op_constants(i)
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(np.asarray(x).view(np.int32), np.asarray(y).view(np.int32)), (name, i)
        assert a[3] == b[3]


def test_mixed_element_types_rejected():
    """INT8 input feeding UINT8 operators does not type-check in the reference; here it is
    MF_ERR_UNSUPPORTED at model creation."""
    mf = importlib.import_module("microflow_rs_amd")
    with open(os.path.join(ROOT, "models", "sine.tflite"), "rb") as f:
        i8 = f.read()
    u8 = to_u8(i8)
    diff = [k for k in range(len(i8)) if i8[k] != u8[k]]
    # flip only the FIRST changed tensor-type byte (9 -> 3): one tensor becomes u8, the rest stay i8
    first_type = next(k for k in diff if i8[k] == 9 and u8[k] == 3)
    mixed = bytearray(i8)
    mixed[first_type] = 3
    with pytest.raises(mf.MicroflowError) as e:
        mf.Model(bytes(mixed))
    assert e.value.status == mf._lib.MF_ERR_UNSUPPORTED


def test_wrong_input_dtype_is_an_error():
    mf = importlib.import_module("microflow_rs_amd")
    m = mf.Model(_u8_model_bytes("sine"))
    with pytest.raises(TypeError):
        m.run_quantized(np.zeros(1, np.int8))


# ---- GPU: HIP path vs the u8 oracle -----------------------------------------------------------
@pytest.fixture(scope="module")
def mf():
    return importlib.import_module("microflow_rs_amd")


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("sine", 64), ("speech", 12), ("person_detect", 3)])
def test_u8_models_bit_exact(mf, O, name, n):
    data = _u8_model_bytes(name)
    m, om = mf.Model(data), O.Model(data)
    rng = np.random.default_rng(21)
    xq = rng.integers(0, 256, (n, m.input_elems)).astype(np.uint8)
    want = om.run_quantized_batch(xq)
    got = m.run_quantized(xq)
    assert got.dtype == np.uint8
    assert np.array_equal(got.reshape(n, -1), want)
    # per-layer localisation on the first input
    _, layers = om.run_quantized(xq[0], layers=True)
    for i, lay in enumerate(layers):
        g = m.run_until(xq[:1], i)
        assert np.array_equal(np.asarray(g).reshape(-1), lay.reshape(-1)), (name, i, m.op(i)["kernel"])
    # predict_quantized / predict (f32 boundary: quantize + dequantize in the u8 domain)
    pq = m.predict_quantized(xq)
    assert np.array_equal(pq.reshape(n, -1).view(np.uint32),
                          np.stack([om.predict_quantized(x).reshape(-1) for x in xq]).view(np.uint32))
    xf = ((xq.astype(f32) - f32(om.in_zp)) * om.in_scale + rng.normal(0, om.in_scale / 3, xq.shape)).astype(f32)
    pf = m.predict(xf)
    wf = np.stack([om.predict(x).reshape(-1) for x in xf])
    assert np.array_equal(pf.reshape(n, -1).view(np.uint32), wf.view(np.uint32))


@pytest.mark.gpu
def test_u8_person_detect_uses_the_fast_kernels(mf, O):
    """A u8 model whose weight zero points are all 128 (what tools/make_u8_model.py produces from
    the i8 person_detect) runs on the same shape-specialised and fused kernels as the i8 model,
    instantiated with the u8 store (XR4 = 0x80808080); ragged batch, fused and layer-wise."""
    data = _u8_model_bytes("person_detect")
    m, om = mf.Model(data), O.Model(data)
    rng = np.random.default_rng(23)
    n = 37
    xq = rng.integers(0, 256, (n, m.input_elems)).astype(np.uint8)
    want = om.run_quantized_batch(xq)
    assert np.array_equal(m.run_quantized(xq).reshape(n, -1), want)
    names = [m.op(i)["kernel"] for i in range(m.num_ops)]
    if not ROUTING_SWITCHED:  # (the default routing; scripts/switch_matrix.sh runs the parity part under every switch)
        # 13 pairs: pair kernels, five of them inside the stage kernel, the last inside pair3_tail (all with the u8 store)
        assert names[0].startswith(("dw3x3_stem8", "penta_rr")), names   # (penta_rr: the stem + ops 1..4 in one launch)
        npairs = sum(k.startswith(("dwpw_rr", "dwpw_mm", "dwpw3x3")) for k in names)
        npairs += 5 * sum(k.startswith("stage_6x6x128") for k in names) + sum(k.startswith("pair3_tail") for k in names)
        npairs += 2 * sum(k.startswith("pair_front_tail") for k in names)   # ops 23..30 in one launch (k_tail3.hip FRONT)
        npairs += 2 * sum(k.startswith(("quad_rr", "penta_rr", "quad_mm")) for k in names)   # two pairs per quad launch (k_quad.hip, k_quad_mm.hip)
        assert npairs == 13, names
        assert names[23].startswith("pair_front_tail") and names[25].startswith("(fused"), names
        assert names[13].startswith("stage_6x6x128"), names
    m.set_fusion(False)
    assert np.array_equal(m.run_quantized(xq).reshape(n, -1), want)
    names = [m.op(i)["kernel"] for i in range(m.num_ops)]
    if not ROUTING_SWITCHED:
        assert sum(k.startswith(("dw3x3_mm", "dw3x3_nhwc")) for k in names) == 13 and sum(k.startswith("pw_mfma") for k in names) == 13, names
    m.set_generic(True)
    assert np.array_equal(m.run_quantized(xq[:5]).reshape(5, -1), want[:5])


@pytest.mark.gpu
def test_u8_model_device_tensors(mf, O):
    import torch
    data = _u8_model_bytes("speech")
    m, om = mf.Model(data), O.Model(data)
    rng = np.random.default_rng(22)
    xq = rng.integers(0, 256, (9, m.input_elems)).astype(np.uint8)
    got = m.run_quantized(torch.as_tensor(xq).cuda())
    assert got.is_cuda and got.dtype == torch.uint8
    assert np.array_equal(got.cpu().numpy().reshape(9, -1), om.run_quantized_batch(xq))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CONV_CASES + [(12, 12, 32, 64, 1, 1, 1, 1, 0, 12, 12, 3)],
                         ids=lambda c: "x".join(map(str, c)))
def test_u8_conv_vs_oracle(mf, O, case):
    rng = np.random.default_rng(31)
    x, w, wzp, izp, oscale, ozp, act, pad, st, ohw, c0, c1 = _conv_inputs(rng, np.uint8, case, False)
    xb = np.stack([x, _rand_q(rng, np.uint8, x.shape), _rand_q(rng, np.uint8, x.shape)])
    opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), st)
    op = mf.ops.prepare_conv_2d(x.shape, w, wzp, izp, oscale, ozp, opts, (c0, c1), ohw)
    want = np.stack([O.conv_2d(v, w, wzp, izp, oscale, ozp, act, pad, st, ohw, c0, c1) for v in xb])
    got = op(xb)
    assert got.dtype == np.uint8 and np.array_equal(got, want), op.kernel


@pytest.mark.gpu
@pytest.mark.parametrize("case", DW_CASES + [(24, 24, 32, 32, 3, 3, 1, 1, 0, 24, 24, 3)],
                         ids=lambda c: "x".join(map(str, c)))
def test_u8_depthwise_vs_oracle(mf, O, case):
    rng = np.random.default_rng(32)
    x, w, wzp, izp, oscale, ozp, act, pad, st, ohw, c0, c1 = _conv_inputs(rng, np.uint8, case, True)
    xb = np.stack([x, _rand_q(rng, np.uint8, x.shape)])
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), st)
    op = mf.ops.prepare_depthwise_conv_2d(x.shape, w, wzp, izp, oscale, ozp, opts, (c0, c1), ohw)
    want = np.stack([O.depthwise_conv_2d(v, w, wzp, izp, oscale, ozp, act, pad, st, ohw, c0, c1) for v in xb])
    got = op(xb)
    assert got.dtype == np.uint8 and np.array_equal(got, want), op.kernel


@pytest.mark.gpu
@pytest.mark.parametrize("case", POOL_CASES, ids=lambda c: "x".join(map(str, c)))
def test_u8_average_pool_vs_oracle(mf, O, case):
    H, W, C_, FH, FW, sh, sw, pad, OH, OW, act = case
    rng = np.random.default_rng(33)
    x = _rand_q(rng, np.uint8, (4, H, W, C_))
    izp, ozp = 130, 7
    c0 = f32(f32(0.05) / f32(0.04))
    c1 = f32(f32(ozp) - f32(f32(f32(0.05) * f32(izp)) / f32(0.04)))
    opts = mf.ops.AveragePool2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), (sh, sw))
    op = mf.ops.prepare_average_pool_2d((H, W, C_), (FH, FW), 0.04, ozp, opts, (c0, c1), (OH, OW), dtype=np.uint8)
    want = np.stack([O.average_pool_2d(v, (FH, FW), 0.04, ozp, act, pad, (sh, sw), (OH, OW), c0, c1) for v in x])
    assert np.array_equal(op(x), want)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FC_CASES + [(1, 4000, 4, 0), (128, 256, 128, 1), (256, 128, 256, 0)],
                         ids=lambda c: "x".join(map(str, c)))
def test_u8_fully_connected_vs_oracle(mf, O, case):
    rng = np.random.default_rng(34)
    x, w, wzp, oscale, ozp, act, consts = _fc_inputs(rng, np.uint8, case, O)
    M, K, N, _ = case
    batch = 3 if M < 64 else 1
    xb = _rand_q(rng, np.uint8, (batch, M, K))
    op = mf.ops.prepare_fully_connected(M, w, wzp, oscale, ozp,
                                        mf.ops.FullyConnectedOptions(mf.FusedActivation(act)), consts)
    want = np.stack([O.fully_connected(v, w, wzp, oscale, ozp, act, *consts) for v in xb])
    got = op(xb)
    assert got.dtype == np.uint8 and np.array_equal(got, want), op.kernel
    if M >= 128:
        assert op.kernel == "fc_mfma"


@pytest.mark.gpu
def test_u8_softmax_quantize_dequantize(mf, O):
    rng = np.random.default_rng(35)
    x = _rand_q(rng, np.uint8, (6, 1, 10))
    op = mf.ops.prepare_softmax(1, 10, 0.07, 1.0 / 256, 0, dtype=np.uint8)
    want = np.stack([O.softmax(v, 0.07, 1.0 / 256, 0) for v in x])
    assert np.array_equal(op(x), want)
    v = np.concatenate([rng.normal(0, 6, 4099).astype(f32),
                        np.array([0.5, -0.5, 1.5, 2.5, 0.49999997, 1e9, -1e9, 0.0, 25.45, 25.55], f32)])
    for scale, zp in ((0.1, 3), (0.0235294122, 0), (1.0, 128), (0.05, 255)):
        got = mf.ops.quantize(v, scale, zp, dtype=np.uint8)
        assert got.dtype == np.uint8
        assert np.array_equal(got, O.quantize_array(v, scale, zp, np.uint8)), (scale, zp)
        q = _rand_q(rng, np.uint8, 1001)
        dg = mf.ops.dequantize(q, scale, zp)
        dw = np.array([O.dequantize(int(t), scale, zp, np.uint8) for t in q], f32)
        assert np.array_equal(dg.view(np.uint32), dw.view(np.uint32))


@pytest.mark.gpu
def test_u8_fc_model_through_predict(mf, O, tmp_path):
    """A synthetic u8 FullyConnected model (tools/make_fc_model.py --u8) through the model path:
    the int8 MFMA GEMM with the u8 zero-point folding."""
    data = synthetic_fc(256, 256, 128, wzp=131, seed=9, u8=True)
    m, om = mf.Model(data), O.Model(data)
    assert m.dtype == np.uint8
    rng = np.random.default_rng(36)
    xq = rng.integers(0, 256, (2, m.input_elems)).astype(np.uint8)
    assert np.array_equal(m.run_quantized(xq).reshape(2, -1), om.run_quantized_batch(xq))
    assert m.op(0)["kernel"] == "fc_mfma"
