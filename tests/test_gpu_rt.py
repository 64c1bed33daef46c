"""The run-time-geometry fast kernels (k_rt.hip: dw3x3_rt, pw_rt) against the oracle: shapes that are in none of the
person_detect tables -- other resolutions, odd sizes, channel counts that are not powers of two, images large enough
to be cut into row bands, non-zero weight zero points (src/ops/depthwise_conv_2d.rs:57-63, conv_2d.rs:57-63), u8.
The reference compiles for any shape (const generics); these are the kernels such shapes run on here."""
import numpy as np
import pytest

from tests.conftest import ROUTING_SWITCHED  # noqa: E402

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def mf():
    import torch
    assert torch.cuda.is_available()
    import microflow_rs_amd as m
    return m


def _consts(rng, n, taps, per_channel=True):
    c0 = rng.uniform(-30, 30, n).astype(f32)
    s = 40.0 / (5476.0 * np.sqrt(taps))
    return c0, (rng.uniform(0.5, 1.5, n if per_channel else 1) * s).astype(f32)


DW_RT = [
    # H, W, C, stride, weight zero points, activation, batch
    (10, 10, 8, 1, False, 3, 7),
    (32, 32, 8, 1, False, 3, 5),       # 64x64-input person_detect shapes
    (32, 32, 16, 2, False, 3, 5),
    (16, 16, 32, 1, False, 3, 9),
    (16, 16, 32, 2, False, 1, 9),
    (8, 8, 64, 1, False, 3, 19),
    (8, 8, 64, 2, False, 0, 19),
    (4, 4, 128, 1, False, 3, 33),
    (4, 4, 128, 2, False, 3, 33),
    (2, 2, 256, 1, False, 3, 35),
    (1, 1, 64, 1, False, 3, 3),
    (64, 64, 8, 1, False, 3, 3),       # 128x128-input shapes: row bands
    (64, 64, 16, 2, False, 3, 3),
    (128, 128, 4, 2, False, 3, 2),     # bands, C = 4
    (13, 12, 12, 1, False, 0, 4),      # odd height, C = 12 (3 channel groups: 510 of 512 threads active)
    (13, 12, 12, 2, False, 1, 4),      # odd height, stride 2
    (9, 20, 24, 2, False, 3, 6),
    (7, 4, 48, 1, False, 3, 5),
    (5, 5, 48, 1, True, 0, 5),         # weight zero points
    (32, 32, 16, 1, True, 3, 3),
    (17, 8, 16, 2, True, 1, 4),
    (48, 48, 8, 1, True, 3, 3),        # a person_detect shape WITH weight zero points (the table kernels need wzp == 0)
    (200, 8, 8, 1, False, 3, 2),       # tall and narrow: bands
    (6, 200, 8, 2, False, 3, 2),       # wide rows: several 1 KiB DMA pieces per row
]


@pytest.mark.parametrize("case", DW_RT, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("u8", [False, True], ids=["i8", "u8"])
def test_depthwise_rt_vs_oracle(mf, O, case, u8):
    H, W, C, S, wz, act, batch = case
    rng = np.random.default_rng(hash(case) % (2 ** 32) + int(u8))
    OH, OW = (H + S - 1) // S, (W + S - 1) // S
    dt = np.uint8 if u8 else np.int8
    lo, hi = (0, 256) if u8 else (-128, 128)
    x = rng.integers(lo, hi, (batch, H, W, C)).astype(dt)
    x[0] = hi - 1
    x[-1, ::2] = lo
    w = rng.integers(lo, hi, (3, 3, C)).astype(dt)
    wzp = (rng.integers(-20, 20, C) + (128 if u8 else 0)).astype(dt) if wz else np.full(C, 128 if u8 else 0, dt)
    izp = int(rng.integers(lo, hi))
    oscale, ozp = 0.0235294122, int(rng.integers(lo, lo + 128))
    c0, c1 = _consts(rng, C, 9)
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(0), (S, S))
    op = mf.ops.prepare_depthwise_conv_2d((H, W, C), w, wzp, izp, oscale, ozp, opts, (c0, c1), (OH, OW))
    assert ROUTING_SWITCHED or op.kernel == "dw3x3_rt<%d%s>" % (S, ",wzp" if wz else ""), op.kernel
    want = np.stack([O.depthwise_conv_2d(x[i], w, wzp, izp, oscale, ozp, act, 0, (S, S), (OH, OW), c0, c1)
                     for i in range(batch)])
    got = op(x)
    assert np.array_equal(got, want), (op.kernel, np.argwhere(got != want)[:5])
    op.set_generic(True)
    assert np.array_equal(op(x), want)


PW_RT = [
    # H, W, K, N, weight zero points, batch
    (16, 16, 16, 16, False, 5),
    (8, 8, 32, 48, False, 7),          # N = 48
    (8, 8, 48, 96, False, 7),          # K = 48: the 64-deep k step hangs over
    (4, 4, 96, 24, False, 9),          # N = 24: not a multiple of 16 (dword stores)
    (4, 4, 192, 64, False, 9),         # K = 192 = 3 k steps
    (2, 2, 512, 128, False, 11),       # K = 512
    (5, 3, 80, 20, False, 3),          # odd pixel count, ragged last chunk
    (32, 32, 8, 32, False, 3),         # K = 8: pixel pairs
    (7, 6, 8, 24, False, 3),           # K = 8, N = 24
    (16, 16, 4, 8, False, 3),          # K = 4: pixel quads (width-0.5 person_detect)
    (16, 16, 16, 16, True, 5),         # weight zero points
    (6, 6, 128, 128, True, 4),         # a person_detect shape WITH weight zero points
    (3, 3, 256, 256, True, 4),
    (9, 7, 48, 20, True, 3),
    (1, 1, 64, 12, False, 70),
]


@pytest.mark.parametrize("case", PW_RT, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("u8", [False, True], ids=["i8", "u8"])
def test_pointwise_rt_vs_oracle(mf, O, case, u8):
    H, W, K, N, wz, batch = case
    rng = np.random.default_rng(hash(case) % (2 ** 32) + int(u8))
    dt = np.uint8 if u8 else np.int8
    lo, hi = (0, 256) if u8 else (-128, 128)
    x = rng.integers(lo, hi, (batch, H, W, K)).astype(dt)
    f = rng.integers(lo, hi, (N, 1, 1, K)).astype(dt)
    fzp = (rng.integers(-30, 30, N) + (128 if u8 else 0)).astype(dt) if wz else np.full(N, 128 if u8 else 0, dt)
    izp, oscale, ozp, act = int(rng.integers(lo, hi)), 0.0235294122, int(rng.integers(lo, lo + 100)), int(rng.choice([0, 1, 3]))
    c0, c1 = _consts(rng, N, K)
    opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding.SAME, (1, 1))
    op = mf.ops.prepare_conv_2d((H, W, K), f, fzp, izp, oscale, ozp, opts, (c0, c1), (H, W))
    assert ROUTING_SWITCHED or op.kernel == "pw_rt<%d,%d%s>" % (K, N, ",wzp" if wz else ""), op.kernel
    want = np.stack([O.conv_2d(x[i], f, fzp, izp, oscale, ozp, act, 0, (1, 1), (H, W), c0, c1) for i in range(batch)])
    got = op(x)
    assert np.array_equal(got, want), (op.kernel, np.argwhere(got != want)[:5])
    op.set_generic(True)
    assert np.array_equal(op(x), want)


def test_rt_kernels_on_a_large_batch_and_accumulator_extremes(mf, O):
    """A batch large enough for every workgroup to walk many steps through the dynamic queue (ragged last step), with
    constant +127 / -128 images (the worst-case accumulators the epilogue mode was chosen for)."""
    import torch
    rng = np.random.default_rng(77)
    H, W, C, S, batch = 20, 12, 8, 1, 4099
    x = rng.integers(-128, 128, (batch, H, W, C)).astype(np.int8)
    x[5], x[6] = 127, -128
    w = rng.integers(-128, 128, (3, 3, C)).astype(np.int8)
    w[:, :, 0], w[:, :, 1] = 127, -128
    wzp = np.zeros(C, np.int8)
    c0, c1 = _consts(rng, C, 9)
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(3), mf.TensorViewPadding(0), (S, S))
    op = mf.ops.prepare_depthwise_conv_2d((H, W, C), w, wzp, -128, 0.0235294122, -128, opts, (c0, c1), (H, W))
    assert ROUTING_SWITCHED or op.kernel == "dw3x3_rt<1>"
    got = op(torch.as_tensor(x).cuda()).cpu().numpy()
    idx = [0, 1, 5, 6, 7, 511, 512, 2048, 4090, 4096, 4097, 4098]
    want = np.stack([O.depthwise_conv_2d(x[i], w, wzp, -128, 0.0235294122, -128, 3, 0, (S, S), (H, W), c0, c1) for i in idx])
    assert np.array_equal(got[idx], want)
    op.set_generic(True)
    assert np.array_equal(op(torch.as_tensor(x).cuda()).cpu().numpy(), got)


ROWS = [
    # kind, H, W, C, N, KH, KW, sh, sw, pad (0 SAME / 1 VALID), weight zero points, batch
    ("conv", 96, 96, 3, 16, 3, 3, 2, 2, 0, False, 5),     # a colour MobileNet stem (rows of 288 bytes)
    ("conv", 32, 32, 3, 16, 3, 3, 2, 2, 0, False, 9),
    ("conv", 20, 16, 4, 8, 5, 5, 1, 1, 0, False, 7),      # 5x5 SAME
    ("conv", 17, 12, 2, 24, 3, 3, 1, 2, 1, False, 6),     # VALID, stride (1, 2), N = 24
    ("conv", 9, 8, 1, 5, 3, 3, 1, 1, 0, False, 11),       # N = 5: byte stores
    ("conv", 12, 12, 4, 64, 3, 3, 2, 2, 0, False, 4),     # N = 64
    ("conv", 10, 8, 3, 12, 1, 1, 1, 1, 0, False, 5),      # 1x1 with 3 input channels
    ("conv", 9, 9, 4, 8, 3, 3, 2, 2, 0, True, 5),         # filter zero points
    ("conv", 14, 16, 3, 10, 4, 2, 2, 1, 1, True, 5),      # even-sized filter, VALID, filter zero points
    ("dw", 128, 128, 1, 8, 3, 3, 2, 2, 0, False, 3),      # the person_detect stem at 128 x 128 (dw3x3_stem_rt takes 4 / 8 outputs)
    ("dw", 128, 128, 1, 6, 3, 3, 2, 2, 0, False, 3),      # ... with 6 outputs: dw_c1_lds
    ("dw", 40, 40, 1, 16, 3, 3, 2, 2, 0, False, 5),       # depth multiplier 16 (dw_c1_lds stops at 8)
    ("dw", 30, 28, 1, 12, 5, 3, 1, 1, 0, True, 4),        # weight zero points
]


@pytest.mark.parametrize("case", ROWS, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("u8", [False, True], ids=["i8", "u8"])
def test_conv_rows_vs_oracle(mf, O, case, u8):
    kind, H, W, C, N, KH, KW, sh, sw, pad, wz, batch = case
    rng = np.random.default_rng(hash(case) % (2 ** 32) + int(u8))
    dt = np.uint8 if u8 else np.int8
    lo, hi = (0, 256) if u8 else (-128, 128)
    if pad == 0:
        OH, OW = -(-H // sh), -(-W // sw)
    else:
        OH, OW = (H - KH) // sh + 1, (W - KW) // sw + 1
    x = rng.integers(lo, hi, (batch, H, W, C)).astype(dt)
    x[0] = hi - 1
    izp, oscale, ozp, act = int(rng.integers(lo, hi)), 0.0235294122, int(rng.integers(lo, lo + 100)), int(rng.choice([0, 1, 3]))
    zp = (rng.integers(-25, 25, N) + (128 if u8 else 0)).astype(dt) if wz else np.full(N, 128 if u8 else 0, dt)
    c0, c1 = _consts(rng, N, KH * KW * C)
    if kind == "conv":
        f = rng.integers(lo, hi, (N, KH, KW, C)).astype(dt)
        opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), (sh, sw))
        op = mf.ops.prepare_conv_2d((H, W, C), f, zp, izp, oscale, ozp, opts, (c0, c1), (OH, OW))
        assert ROUTING_SWITCHED or op.kernel == "conv_rows_lds" + ("<wzp>" if wz else ""), op.kernel
        want = np.stack([O.conv_2d(x[i], f, zp, izp, oscale, ozp, act, pad, (sh, sw), (OH, OW), c0, c1) for i in range(batch)])
    else:
        w = rng.integers(lo, hi, (KH, KW, N)).astype(dt)
        opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), (sh, sw))
        op = mf.ops.prepare_depthwise_conv_2d((H, W, C), w, zp, izp, oscale, ozp, opts, (c0, c1), (OH, OW))
        # (3x3 stride-2 stems with 4 / 8 outputs have their own kernel, tested in test_stem_rt_vs_oracle)
        assert ROUTING_SWITCHED or op.kernel in ("dw_rows_lds" + ("<wzp>" if wz else ""), "dw_c1_lds", "dw3x3_stem_rt<%d>" % N), op.kernel
        want = np.stack([O.depthwise_conv_2d(x[i], w, zp, izp, oscale, ozp, act, pad, (sh, sw), (OH, OW), c0, c1) for i in range(batch)])
    got = op(x)
    assert np.array_equal(got, want), (op.kernel, np.argwhere(got != want)[:5])
    op.set_generic(True)
    assert np.array_equal(op(x), want)


STEM_RT = [
    # H, W, outputs, activation, batch
    (128, 128, 8, 3, 3),      # the person_detect stem at 128 x 128 (one image per step)
    (64, 64, 8, 3, 11),       # 4 images per step, a ragged last step
    (96, 96, 4, 3, 5),        # width 0.5: four pixels per 16-byte group, v_mfma_i32_16x16x64_i8
    (48, 48, 8, 1, 3),
    (80, 80, 8, 3, 4),        # 50 tiles per image: a ragged last quad
    (112, 112, 8, 0, 2),
    (160, 160, 8, 3, 2),
    (33, 48, 4, 3, 7),        # odd height: the bottom row of windows reads the izp row
    (17, 16, 8, 1, 9),        # 16 wide: four 16-byte groups per output row, odd height
    (2, 16, 4, 3, 70),        # the smallest image
    (30, 240, 8, 3, 3),       # wide rows
]


@pytest.mark.parametrize("case", STEM_RT, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("u8", [False, True], ids=["i8", "u8"])
def test_stem_rt_vs_oracle(mf, O, case, u8):
    """dw3x3_stem_rt (k_rt.hip): DepthwiseConv2D 3x3 stride 2 SAME, one input channel -> 4 / 8 outputs at any resolution
    (src/ops/depthwise_conv_2d.rs:28-105 with every output channel reading input channel 0)."""
    H, W, N, act, batch = case
    rng = np.random.default_rng(hash(case) % (2 ** 32) + int(u8))
    dt = np.uint8 if u8 else np.int8
    lo, hi = (0, 256) if u8 else (-128, 128)
    OH, OW = (H + 1) // 2, W // 2
    x = rng.integers(lo, hi, (batch, H, W, 1)).astype(dt)
    x[0] = hi - 1
    x[-1, ::2] = lo
    w = rng.integers(lo, hi, (3, 3, N)).astype(dt)
    wzp = np.full(N, 128 if u8 else 0, dt)
    izp, oscale, ozp = int(rng.integers(lo, hi)), 0.0235294122, int(rng.integers(lo, lo + 100))
    c0, c1 = _consts(rng, N, 9)
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(0), (2, 2))
    op = mf.ops.prepare_depthwise_conv_2d((H, W, 1), w, wzp, izp, oscale, ozp, opts, (c0, c1), (OH, OW))
    assert ROUTING_SWITCHED or op.kernel == "dw3x3_stem_rt<%d>" % N, op.kernel
    want = np.stack([O.depthwise_conv_2d(x[i], w, wzp, izp, oscale, ozp, act, 0, (2, 2), (OH, OW), c0, c1) for i in range(batch)])
    got = op(x)
    assert np.array_equal(got, want), (op.kernel, np.argwhere(got != want)[:5])
    op.set_generic(True)
    assert np.array_equal(op(x), want)


def test_stem_rt_on_a_large_batch(mf, O):
    """Many steps per workgroup (the step queue, both staging buffers): 4099 images of 64 x 64 equal the shape-generic kernel's."""
    import torch
    rng = np.random.default_rng(77)
    B = 4099
    x = rng.integers(-128, 128, (B, 64, 64, 1)).astype(np.int8)
    w = rng.integers(-128, 128, (3, 3, 8)).astype(np.int8)
    wzp = np.zeros(8, np.int8)
    c0, c1 = _consts(rng, 8, 9)
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(3), mf.TensorViewPadding(0), (2, 2))
    op = mf.ops.prepare_depthwise_conv_2d((64, 64, 1), w, wzp, -128, 0.0235294122, -128, opts, (c0, c1), (32, 32))
    assert ROUTING_SWITCHED or op.kernel == "dw3x3_stem_rt<8>", op.kernel
    xd = torch.as_tensor(x).cuda()
    got = op(xd).cpu().numpy()
    idx = [0, 1, 2047, 4096, 4098]
    want = np.stack([O.depthwise_conv_2d(x[i], w, wzp, -128, 0.0235294122, -128, 3, 0, (2, 2), (32, 32), c0, c1) for i in idx])
    assert np.array_equal(got[idx], want)
    op.set_generic(True)
    assert np.array_equal(op(xd).cpu().numpy(), got)


CONV_MM = [
    # H, W, C, N, KH, KW, sh, sw, pad, filter zero points, batch
    (16, 16, 16, 16, 3, 3, 1, 1, 0, False, 5),       # ResNet-8-like blocks
    (16, 16, 16, 32, 3, 3, 2, 2, 0, False, 5),
    (8, 8, 32, 64, 3, 3, 2, 2, 0, False, 9),
    (8, 8, 64, 64, 3, 3, 1, 1, 0, False, 9),         # K = 576: 9 k steps
    (32, 32, 16, 16, 3, 3, 1, 1, 0, False, 3),
    (12, 10, 16, 24, 5, 5, 1, 1, 0, False, 4),       # 5x5, N = 24
    (9, 7, 32, 20, 3, 5, 2, 1, 1, False, 4),         # rectangular filter, VALID, N = 20
    (10, 10, 48, 40, 3, 3, 1, 1, 0, False, 4),       # C = 48: taps straddle the 64-deep k steps
    (64, 64, 16, 16, 3, 3, 1, 1, 0, False, 2),       # row bands
    (6, 6, 128, 32, 3, 3, 1, 1, 0, False, 6),        # K = 1152: 18 k steps
    (16, 16, 16, 16, 3, 3, 1, 1, 0, True, 5),        # filter zero points
    (8, 8, 32, 72, 3, 3, 2, 2, 0, True, 6),          # ... and 5 tiles = two blocks
    (7, 7, 16, 128, 2, 2, 1, 1, 1, False, 5),        # even filter, VALID, N = 128 (two blocks)
]


@pytest.mark.parametrize("case", CONV_MM, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("u8", [False, True], ids=["i8", "u8"])
def test_conv_mm_vs_oracle(mf, O, case, u8):
    H, W, C, N, KH, KW, sh, sw, pad, wz, batch = case
    rng = np.random.default_rng(hash(case) % (2 ** 32) + int(u8))
    dt = np.uint8 if u8 else np.int8
    lo, hi = (0, 256) if u8 else (-128, 128)
    if pad == 0:
        OH, OW = -(-H // sh), -(-W // sw)
    else:
        OH, OW = (H - KH) // sh + 1, (W - KW) // sw + 1
    x = rng.integers(lo, hi, (batch, H, W, C)).astype(dt)
    x[0] = hi - 1
    f = rng.integers(lo, hi, (N, KH, KW, C)).astype(dt)
    zp = (rng.integers(-25, 25, N) + (128 if u8 else 0)).astype(dt) if wz else np.full(N, 128 if u8 else 0, dt)
    izp, oscale, ozp, act = int(rng.integers(lo, hi)), 0.0235294122, int(rng.integers(lo, lo + 100)), int(rng.choice([0, 1, 3]))
    c0, c1 = _consts(rng, N, KH * KW * C)
    opts = mf.ops.Conv2DOptions(mf.FusedActivation(act), mf.TensorViewPadding(pad), (sh, sw))
    op = mf.ops.prepare_conv_2d((H, W, C), f, zp, izp, oscale, ozp, opts, (c0, c1), (OH, OW))
    assert ROUTING_SWITCHED or op.kernel == "conv_mm_rt" + ("<wzp>" if wz else ""), op.kernel
    want = np.stack([O.conv_2d(x[i], f, zp, izp, oscale, ozp, act, pad, (sh, sw), (OH, OW), c0, c1) for i in range(batch)])
    got = op(x)
    assert np.array_equal(got, want), (op.kernel, np.argwhere(got != want)[:5])
    op.set_generic(True)
    assert np.array_equal(op(x), want)
