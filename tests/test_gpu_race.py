"""Race screen.  The fused kernels synchronise waves through barriers, reuse LDS tiles across phases
and steps, draw their steps from device counters and write packed bytes with hand-scheduled SDWA sequences
(k_common.hpp: a forwarding hazard is padded by hand).  A fault in any of that shows as a run-to-run difference under load,
so the whole fused step is repeated on a large ragged batch and every run's output checksum must equal the first run's and
the layer-wise kernels' (which share none of that machinery).  ~10 s on an MI355X."""
import os

import numpy as np
import pytest

from tests.conftest import model_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mf():
    import microflow_rs_amd as mf
    return mf


@pytest.mark.parametrize("name,batch,reps", [("person_detect", 32768 + 13, 30), ("speech", 65536 + 7, 30)])
def test_fused_step_is_stable_under_repetition(mf, name, batch, reps):
    import torch
    from microflow_rs_amd.model import checksum_i8, synth_i8
    m = mf.model(model_path(name))
    m.prepare(batch)
    x = synth_i8(99, 0, batch * m.input_elems).reshape((batch,) + m.input_shape)
    sums = []
    for _ in range(reps):  # back to back on one stream: launch k + 1's workgroups start while launch k's tail still runs
        sums.append(m.run_quantized(x))
    torch.cuda.synchronize()
    sums = [checksum_i8(y.reshape(-1)) for y in sums]
    assert len(set(sums)) == 1, [i for i, c in enumerate(sums) if c != sums[0]]
    m.set_fusion(False)
    ref = checksum_i8(m.run_quantized(x).reshape(-1))
    m.set_fusion(True)
    assert ref == sums[0]


def test_fused_groups_are_stable_at_their_boundaries(mf):
    """the tensor at the END of every fused group (a whole intermediate tensor, not the model's 2 output bytes)"""
    from microflow_rs_amd.model import checksum_i8, synth_i8
    m = mf.model(model_path("person_detect"))
    batch = 16384 + 5
    m.prepare(batch)
    x = synth_i8(98, 0, batch * m.input_elems).reshape((batch,) + m.input_shape)
    ends = [i - 1 for i in range(1, m.num_ops) if m.op(i)["kernel"] and not m.op(i)["kernel"].startswith("(fused")]
    for last in ends:
        cs = {checksum_i8(m.run_until(x, last).reshape(-1)) for _ in range(6)}
        assert len(cs) == 1, last
        m.set_fusion(False)
        ref = checksum_i8(m.run_until(x, last).reshape(-1))
        m.set_fusion(True)
        assert cs == {ref}, last


def test_one_operator_handle_on_two_streams(mf, O):
    """mf_op_run takes the caller's stream, so one handle may be in flight on two streams at once; every
    launch draws its steps from its own counter set (kernels.hpp DYNQ_RING), so neither launch may lose or repeat a step."""
    import torch
    rng = np.random.default_rng(5)
    H = W = 24
    C = 32
    w = rng.integers(-128, 128, (3, 3, C)).astype(np.int8)
    c0 = rng.uniform(-20, 20, C).astype(np.float32)
    c1 = (rng.uniform(0.5, 1.5, C) * 0.002).astype(np.float32)
    opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(3), mf.TensorViewPadding.SAME, (1, 1))
    op = mf.ops.prepare_depthwise_conv_2d((H, W, C), w, np.zeros(C, np.int8), -128, 0.0235294122, -128, opts, (c0, c1), (H, W))
    B = 20000
    xa = torch.randint(-128, 128, (B, H, W, C), dtype=torch.int8, device="cuda")
    xb = torch.randint(-128, 128, (B, H, W, C), dtype=torch.int8, device="cuda")
    ya, yb = op(xa).clone(), op(xb).clone()   # serial reference (checked against the oracle on two images)
    want = O.depthwise_conv_2d(xa[7].cpu().numpy(), w, np.zeros(C, np.int8), -128, 0.0235294122, -128, 3, 0, (1, 1), (H, W), c0, c1)
    assert np.array_equal(ya[7].cpu().numpy(), want)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(8):
        with torch.cuda.stream(sa):
            za = op(xa)
        with torch.cuda.stream(sb):
            zb = op(xb)
        torch.cuda.synchronize()
        assert torch.equal(za, ya) and torch.equal(zb, yb)
