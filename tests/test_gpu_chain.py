"""The run-time-geometry fused chain kernel (k_chain.hip): consecutive DepthwiseConv2D 3x3 + Conv2D 1x1 pairs of any
height / width (C % 16 == 0, N % 16 == 0) in one launch, the tensors between them in LDS.  Parity: bit-exact against the
oracle (src/ops/depthwise_conv_2d.rs:28-105 + src/ops/conv_2d.rs:28-108 restated) on sampled images, and equal to the
layer-wise kernels on every image of ragged multi-step batches."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tflite_writer as tw  # noqa: E402
from tests.conftest import ROUTING_SWITCHED, model_path  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = [
    # side, width, element type, batch
    (64, 1.0, tw.INT8, 301),
    (128, 1.0, tw.INT8, 67),
    (96, 0.5, tw.INT8, 259),
    (96, 0.75, tw.INT8, 130),   # 12 / 24 / 48 / 96 / 192 channels: three- and six-group tensors, TB = 3 blocks
    (80, 1.0, tw.INT8, 77),     # 40x40, 20x20, 10x10, 5x5, 3x3: odd sizes (few divisors: small column grids, many images per step)
    (64, 1.0, tw.UINT8, 140),
    (112, 1.0, tw.INT8, 33),
    (128, 0.5, tw.INT8, 41),    # C = 4 and C = 8 stride 2 as superpixel pairs at 64x64
    (64, 0.5, tw.UINT8, 90),
    (96, 0.25, tw.INT8, 100),   # C = 2 (eight pixels per superpixel), C = 4 stride 2
    (64, 1.0, tw.INT8, 90, 40),     # small weights: the bit-pattern epilogues in the 256-deep layers (pair3_tail<2,2>, mode 1)
    (128, 1.0, tw.UINT8, 35, 40),   # ... pair3_tail<4,4>, u8
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d-w%s-%s-b%d%s" % (c[0], c[0], c[1], "u8" if c[2] == tw.UINT8 else "i8", c[3], "-wmax%d" % c[4] if len(c) > 4 else ""))
def test_generated_models_through_chains(O, case):
    import torch
    import microflow_rs_amd as mf
    side, width, elem, batch = case[:4]
    blob = tw.person_detect_like(np.random.default_rng(side + int(width * 100)), side, width, elem, wmax=case[4] if len(case) > 4 else None)
    m, om = mf.Model(blob), O.Model(blob)
    m.prepare(batch)
    rng = np.random.default_rng(side)
    lo, hi = (0, 256) if m.dtype == np.uint8 else (-128, 128)
    xq = rng.integers(lo, hi, (batch, m.input_elems)).astype(m.dtype)
    xq[0], xq[1] = hi - 1, lo
    x = torch.from_numpy(xq).cuda().reshape((batch,) + m.input_shape)
    fused_names = [m.op(i)["kernel"] for i in range(m.num_ops)]
    got = m.run_quantized(x).reshape(batch, -1).cpu().numpy()
    m.set_fusion(False)
    lw = m.run_quantized(x).reshape(batch, -1).cpu().numpy()
    m.set_fusion(True)
    assert np.array_equal(got, lw), (fused_names, np.argwhere(got != lw)[:4])
    idx = [0, 1, 2, batch // 2, batch - 1]
    assert np.array_equal(got[idx], om.run_quantized_batch(xq[idx]))
    # the tensor at the end of every fused group, every image (fused vs layer-wise kernels)
    ends = [i - 1 for i in range(1, m.num_ops) if m.op(i)["kernel"] and not m.op(i)["kernel"].startswith("(fused")]
    for last in ends:
        a = m.run_until(x, last)
        m.set_fusion(False)
        b = m.run_until(x, last)
        m.set_fusion(True)
        assert torch.equal(a, b), (last, fused_names[last], m.op(last)["name"])
    if not ROUTING_SWITCHED and side != 96:
        assert any(k.startswith("chain_rt<") for k in fused_names), fused_names
    if not ROUTING_SWITCHED and width in (1.0, 0.5) and side in (64, 96, 128):  # the last pair + the tail in one launch, 2x2 / 3x3 / 4x4
        assert "pair3_tail<%d,%d,%d,2>" % (side // 32, side // 32, int(256 * width)) in fused_names, fused_names
        if False:   # (whether pairs are chained is the planner's cost decision)
            assert any(k.startswith("chain_rt<") and "|" in k for k in fused_names), fused_names


def test_person_detect_through_chains_only():
    """MF_CHAIN_ALL=1: the chain kernel instead of the table pairs / quads / stage on person_detect.tflite itself (C >= 16
    pairs), against the default routing's output and the oracle, ragged batch -- in a child process (routing is decided at
    operator creation)."""
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import microflow_rs_amd as mf
from microflow_rs_amd.model import checksum_i8, synth_i8
from oracle import oracle as O
m = mf.model(%r); B = 1031
m.prepare(B)
x = synth_i8(77, 0, B * m.input_elems).reshape((B,) + m.input_shape)
y = m.run_quantized(x)
om = O.Model(%r)
idx = [0, 5, B - 1]
ok = np.array_equal(y.reshape(B, -1)[idx].cpu().numpy(), om.run_quantized_batch(x.reshape(B, -1)[idx].cpu().numpy()))
print("KERNELS", sorted({m.op(i)["kernel"] for i in range(m.num_ops)}))
print("RESULT %%016x %%d" %% (checksum_i8(y.reshape(-1)), int(ok)))
''' % (ROOT, model_path("person_detect"), model_path("person_detect"))
    outs = []
    for env_extra in ({}, {"MF_DEV": "1", "MF_CHAIN_ALL": "1"}):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
        assert line[2] == "1", r.stdout
        outs.append((line[1], r.stdout))
    assert outs[0][0] == outs[1][0]
    if not ROUTING_SWITCHED:  # (which kernels run by default is the default routing's business: scripts/switch_matrix.sh)
        assert "chain_rt<" in outs[1][1] and "chain_rt<" not in outs[0][1], outs[1][1]
