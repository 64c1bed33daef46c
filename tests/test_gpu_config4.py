"""BASELINE.json config 4 -- person_detect.tflite, 524 288 images sharded over 8 GPUs -- on the hardware a test box has.

The reference's batch is 524 288 independent `predict()` calls (microflow-macros/src/lib.rs:188-201), so a shard is a
contiguous slice of the stream and nothing crosses GPUs on the data path.  What a one-GPU box CAN exercise of that
configuration is everything but the xGMI hop:

* the whole 524 288-image batch in ONE launch sequence on one device: 4.8 GB of input (images >= 466 034 start beyond
  byte 2^32 of the input tensor; the intermediate tensors pass 2^32 bytes from image 116 509 on), checked against the
  oracle on sampled images on both sides of those offsets and at the ends, and at the end tensors of the fused groups;
* the eight shards `shard_range(524288, r, 8)` run one by one give the same eight checksums as the slices of the big run.
"""
import numpy as np
import pytest

from tests.conftest import model_path

pytestmark = pytest.mark.gpu

TOTAL, WORLD = 524288, 8


@pytest.fixture(scope="module")
def big():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import microflow_rs_amd as mf
    from microflow_rs_amd.model import synth_i8
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2 ** 30:
        pytest.skip("needs 80 GB of free HBM (MI355X has 288)")
    m = mf.model(model_path("person_detect"))
    m.prepare(TOTAL)
    x = synth_i8(3 + 0x4D4643, 0, TOTAL * m.input_elems).reshape((TOTAL,) + m.input_shape)
    assert x.numel() > 2 ** 32
    y = m.run_quantized(x)
    torch.cuda.synchronize()
    yield mf, m, x, y
    del x, y, m
    torch.cuda.empty_cache()


def _sample_indices(elems_per_image):
    first_beyond = -(-2 ** 32 // elems_per_image)  # first image that STARTS at or beyond byte 2^32
    rng = np.random.default_rng(4)
    idx = set(range(8)) | set(range(TOTAL - 8, TOTAL))
    idx |= set(range(first_beyond - 3, first_beyond + 3))
    for mult in (1, 2, 4, 8):  # the larger tensors pass 2^32 (and 2^33, ...) earlier: 36 864-, 18 432-byte images
        b = -(-2 ** 32 * mult // 36864)
        if b + 2 < TOTAL:
            idx |= {b - 1, b, b + 1}
    idx |= set(int(i) for i in rng.integers(first_beyond, TOTAL, 24))
    idx |= set(int(i) for i in rng.integers(0, first_beyond, 8))
    return sorted(i for i in idx if 0 <= i < TOTAL)


def test_one_launch_sequence_over_524288_images_matches_the_oracle(big, O):
    mf, m, x, y = big
    idx = _sample_indices(m.input_elems)
    assert any(i * m.input_elems >= 2 ** 32 for i in idx)
    import torch
    sel = torch.tensor(idx, device=x.device)
    xs = x.reshape(TOTAL, -1).index_select(0, sel).cpu().numpy()
    got = y.reshape(TOTAL, -1).index_select(0, sel).cpu().numpy()
    om = O.Model(model_path("person_detect"))
    want = om.run_quantized_batch(xs)
    assert np.array_equal(got, want), [idx[i] for i in np.nonzero((got != want).any(axis=1))[0]]
    # the inputs really are the global stream's images (a wrong 64-bit offset in the generator would also "agree")
    from tests.synth import synth_i8 as host_synth
    for k in (0, len(idx) // 2, len(idx) - 1):
        assert np.array_equal(xs[k], host_synth(3, idx[k], 1, m.input_elems)[0]), idx[k]


def test_group_end_tensors_beyond_4GiB_match_the_oracle(big, O):
    """the int8 tensors the fused launches hand to each other (18 432 .. 9 216 bytes per image: up to 9.7 GB at this batch)"""
    mf, m, x, y = big
    import torch
    om = O.Model(model_path("person_detect"))
    ends = [i - 1 for i in range(1, m.num_ops) if m.op(i)["kernel"] and not m.op(i)["kernel"].startswith("(fused")]
    idx = [0, 1, 233016, 233017, 466033, 466034, 466035, TOTAL - 2, TOTAL - 1]
    sel = torch.tensor(idx, device=x.device)
    xs = x.reshape(TOTAL, -1).index_select(0, sel).cpu().numpy()
    layers = [om.run_quantized(xs[k], layers=True)[1] for k in range(len(idx))]
    for last in ends[:5]:
        t = m.run_until(x, last).reshape(TOTAL, -1)
        got = t.index_select(0, sel).cpu().numpy()
        del t
        for k in range(len(idx)):
            assert np.array_equal(got[k], layers[k][last].reshape(-1)), (last, idx[k])
    torch.cuda.empty_cache()


def test_eight_shards_equal_the_slices_of_the_one_big_run(big):
    mf, m, x, y = big
    import torch
    from microflow_rs_amd.model import checksum_i8
    from microflow_rs_amd.shard import shard_range
    ms = mf.model(model_path("person_detect"))
    ms.prepare(TOTAL // WORLD)
    seen = set()
    xf, yf = x.reshape(TOTAL, -1), y.reshape(TOTAL, -1)
    for r in range(WORLD):
        first, count = shard_range(TOTAL, r, WORLD)
        assert count == 65536  # BASELINE config 4 = eight times config 3
        ys = ms.run_quantized(xf[first:first + count].reshape((count,) + m.input_shape))
        a, b = checksum_i8(ys.reshape(-1)), checksum_i8(yf[first:first + count].reshape(-1))
        assert a == b, r
        seen.add(a)
    torch.cuda.synchronize()
    assert len(seen) == WORLD  # eight different slices of the stream
