"""Whole-model parity on the GPU through the C ABI: the reference's golden vectors,
the recorded sine outputs, the sample tensors, seeded batches against the CPU oracle
(with per-layer localisation), ragged batch sizes, and full-size properties."""
import csv
import importlib
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROUTING_SWITCHED, model_path
from tests.synth import SEED, layer_checksum, synth_i8

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def mf():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import microflow_rs_amd as m
    return m


@pytest.fixture(scope="module")
def models(mf):
    return {n: mf.model(model_path(n)) for n in ("sine", "speech", "person_detect")}


@pytest.mark.parametrize("name", ["sine", "speech", "person_detect"])
def test_reference_golden_vectors(models, kats, name):
    """tests/{sine,speech,person_detect}.rs: exact f32 outputs for a constant 0.5 input."""
    k = kats[name + "_model"]
    m = models[name]
    out = m.predict(np.full(m.input_shape, k["input_fill"], f32))
    assert out.shape == m.output_shape
    assert np.array_equal(out.reshape(-1), np.array(k["output"], f32)), out
    # and with the shape-generic kernels only
    m.set_generic(True)
    out = m.predict(np.full(m.input_shape, k["input_fill"], f32))
    m.set_generic(False)
    assert np.array_equal(out.reshape(-1), np.array(k["output"], f32)), out


def test_sine_500_recorded_outputs(models):
    rows = list(csv.reader(open(os.path.join(GOLDEN, "sine_microflow.csv"))))[1:]
    x = np.array([f32(r[0]) for r in rows], f32).reshape(-1, 1, 1)
    y = np.array([f32(r[1]) for r in rows], f32)
    out = models["sine"].predict(x)          # one batch of 500 independent predicts
    assert out.shape == (500, 1, 1)
    assert np.array_equal(out.reshape(-1), y)


def test_sample_tensors(models, samples, oracle_vectors):
    pm, sm = models["person_detect"], models["speech"]
    for name, m in (("PERSON", pm), ("NO_PERSON", pm), ("YES", sm), ("NO", sm)):
        q = m.run_quantized(samples[name])
        assert np.array_equal(q.reshape(-1), oracle_vectors["sample_" + name].reshape(-1)), name
        f = m.predict_quantized(samples[name])
        assert np.array_equal(f.reshape(-1), oracle_vectors["sample_" + name + "_f32"].reshape(-1)), name


@pytest.mark.parametrize("name,cfg", [("sine", 1), ("speech", 2), ("person_detect", 3)])
def test_committed_oracle_vectors(models, oracle_vectors, name, cfg):
    m = models[name]
    n = int(oracle_vectors[name + "_n"][0])
    x = synth_i8(cfg, 0, n, m.input_elems)
    out = m.run_quantized(x.reshape((n,) + m.input_shape))
    assert np.array_equal(out.reshape(n, -1), oracle_vectors[name + "_final"])
    # per-layer checksums localise a mismatch to the first wrong operator
    sums = oracle_vectors[name + "_layer_checksums"]
    for i in range(m.num_ops):
        lay = m.run_until(x.reshape((n,) + m.input_shape), i)
        got = [layer_checksum(lay[j]) for j in range(n)]
        assert got == sums[:, i].tolist(), (name, "first mismatching op", i, m.op(i)["name"], m.op(i)["kernel"])


@pytest.mark.parametrize("name,cfg,n", [("speech", 2, 96), ("person_detect", 3, 48)])
def test_seeded_batch_vs_oracle_live(models, O, name, cfg, n):
    """Fresh comparison against the oracle (not the committed fixtures), different images."""
    m = models[name]
    o = O.Model(model_path(name))
    x = synth_i8(cfg, 1000, n, m.input_elems)
    want = o.run_quantized_batch(x)
    got = m.run_quantized(x.reshape((n,) + m.input_shape)).reshape(n, -1)
    assert np.array_equal(got, want)
    # f32 entry point: identical quantized data -> identical outputs
    xf = (x.astype(f32) - f32(m.input_zero_point)) * m.input_scale
    gf = m.predict(xf.reshape((n,) + m.input_shape)).reshape(n, -1)
    wf = np.stack([o.predict(xf[i]) for i in range(8)]).reshape(8, -1)
    assert np.array_equal(gf[:8], wf)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 17, 129])
def test_ragged_batches(models, O, n):
    """Batch sizes that are not multiples of any tile (images per workgroup, MFMA chunks)."""
    m = models["person_detect"]
    o = O.Model(model_path("person_detect"))
    x = synth_i8(3, 5000, n, m.input_elems)
    got = m.run_quantized(x.reshape((n,) + m.input_shape)).reshape(n, -1)
    k = min(n, 4)
    want = o.run_quantized_batch(x[:k])
    assert np.array_equal(got[:k], want)
    # batch invariance: every image's result is independent of its position / batch size
    single = np.stack([m.run_quantized(x[i].reshape(m.input_shape)).reshape(-1) for i in (0, n - 1)])
    assert np.array_equal(single, got[[0, n - 1]])


def test_device_synth_matches_host(mf):
    from microflow_rs_amd.model import checksum_i8, synth_i8 as dsynth
    d = dsynth(SEED + 3, 9216 * 5, 9216 * 3 + 5)
    h = synth_i8(3, 0, 9, 9216).reshape(-1)[9216 * 5: 9216 * 8 + 5]
    assert np.array_equal(d.cpu().numpy(), h)
    assert checksum_i8(d) == int(layer_checksum(h))


def test_full_size_properties(mf, models, O):
    """BASELINE config 3 size (65536 images): size-independent properties.
    - fast kernels == shape-generic kernels over the whole batch (checksum of checksums),
    - a sampled subset equals the oracle,
    - shards recombine: running [0,B) equals running [0,B/2) and [B/2,B) separately."""
    import torch
    from microflow_rs_amd.model import checksum_i8, synth_i8 as dsynth
    m = models["person_detect"]
    B = 65536
    x = dsynth(SEED + 3, 0, B * m.input_elems).reshape((B,) + m.input_shape)
    # 28 structured images (levels, ramps, checkerboards, blobs: they exercise the late layers, which noise barely
    # does) planted across the batch, including the first / last image of a workgroup step and of the batch
    from tests.synth import structured_images
    st = structured_images(96)
    spots = [0, 1, 2, 3, 4, 7, 8, 15, 16, 17, 255, 256, 4095, 4096, 16383, 16384, 21845, 32767, 32768, 40000, 43690,
             49151, 49152, 60000, 65532, 65533, 65534, 65535][: len(st)]
    x[spots] = torch.from_numpy(st).cuda().reshape((len(st),) + m.input_shape)
    y = m.run_quantized(x)
    m.sync()
    full = checksum_i8(y)
    # sampled oracle check: the structured images + 68 noise images spread over the batch
    idx = sorted(set(spots + [5, 777, 1023, 1024, 30000, 65531] + list(range(11, B, B // 62))))
    assert len(idx) >= 96
    o = O.Model(model_path("person_detect"))
    xs = x[idx].cpu().numpy().reshape(len(idx), -1)
    want = o.run_quantized_batch(xs)
    assert np.array_equal(y[idx].cpu().numpy().reshape(len(idx), -1), want)
    assert len({tuple(r) for r in want.tolist()}) >= 20     # the sample really spreads over many different outputs
    # shard recombination (the multi-GPU decomposition on one device)
    y0 = m.run_quantized(x[: B // 2])
    y1 = m.run_quantized(x[B // 2:])
    m.sync()
    assert checksum_i8(torch.cat([y0, y1])) == full
    # generic kernels on a slice large enough to cross every tile boundary many times
    S = 4096
    m.set_generic(True)
    yg = m.run_quantized(x[:S])
    m.sync()
    m.set_generic(False)
    assert torch.equal(yg, y[:S])


def test_speech_batch_4096(mf, models, O):
    """BASELINE config 2: speech, batch 4096, bit-exact sample check + batch invariance."""
    from microflow_rs_amd.model import synth_i8 as dsynth
    m = models["speech"]
    B = 4096
    x = dsynth(SEED + 2, 0, B * m.input_elems).reshape((B,) + m.input_shape)
    y = m.run_quantized(x)
    o = O.Model(model_path("speech"))
    idx = list(range(0, B, 257))
    want = o.run_quantized_batch(x[idx].cpu().numpy().reshape(len(idx), -1))
    assert np.array_equal(y[idx].cpu().numpy().reshape(len(idx), -1), want)


def test_speech_batch_65536(mf, models, O):
    """speech at the throughput batch (16 steps per workgroup of the one-launch kernel, ragged tail included):
    sampled inferences against the oracle, the first 4096 against a batch-4096 run, one launch vs operator by operator."""
    from microflow_rs_amd.model import synth_i8 as dsynth
    m = models["speech"]
    B = 65536 + 7  # not a multiple of the 16 images of a workgroup step
    x = dsynth(SEED + 3, 0, B * m.input_elems).reshape((B,) + m.input_shape)
    y = m.run_quantized(x).reshape(B, -1)
    o = O.Model(model_path("speech"))
    idx = list(range(0, B, 2111)) + [B - 8, B - 7, B - 1]
    want = o.run_quantized_batch(x[idx].cpu().numpy().reshape(len(idx), -1))
    assert np.array_equal(y[idx].cpu().numpy(), want)
    assert np.array_equal(y[:4096].cpu().numpy(), m.run_quantized(x[:4096]).reshape(4096, -1).cpu().numpy())
    m.set_fusion(False)
    try:
        z = m.run_quantized(x).reshape(B, -1)
    finally:
        m.set_fusion(True)
    assert np.array_equal(y.cpu().numpy(), z.cpu().numpy())


_NO_QUAD_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import microflow_rs_amd as mf
from tests.synth import synth_i8
m = mf.model({root!r} + "/models/person_detect.tflite")
x = synth_i8(3, 0, {n}, m.input_elems)
m.prepare({n})
names = [m.op(i)["kernel"] for i in range(m.num_ops)]
assert not any(k.startswith(("quad_rr", "penta_rr", "quad_mm")) for k in names), names
assert sum(k.startswith("dwpw_rr") for k in names) == 4, names
np.save({out!r}, np.asarray(m.run_until(x, 8)))
np.save({out!r} + ".12.npy", np.asarray(m.run_until(x, 12)))
print("no-quad ok")
"""


def test_quads_equal_the_four_pair_launches(models, O, tmp_path):
    """Ops 1..8 of person_detect run as two quad kernels (k_quad.hip: two depthwise + 1x1 pairs per launch, the tensor between
    them in LDS).  The tensor after op 8 must be the one the four separate pair launches produce (MF_NO_QUAD=1, in a child
    process: the switch is read once) on a batch that is not a multiple of anything, and the oracle's on its first images."""
    import subprocess
    import sys
    m = models["person_detect"]
    n = 1031
    m.prepare(n)
    names = [m.op(i)["kernel"] for i in range(m.num_ops)]
    if ROUTING_SWITCHED or not any(k.startswith(("quad_rr", "penta_rr")) for k in names):
        pytest.skip("quads switched off / non-default routing")
    x = synth_i8(3, 0, n, m.input_elems)
    got = np.asarray(m.run_until(x, 8)).reshape(n, -1)
    om = O.Model(model_path("person_detect"))
    for b in (0, 1, n - 1):
        _, layers = om.run_quantized(x[b], layers=True)
        assert np.array_equal(got[b], layers[8].reshape(-1)), b
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "noquad.npy")
    env = dict(os.environ, MF_DEV="1", MF_NO_QUAD="1")
    r = subprocess.run([sys.executable, "-c", _NO_QUAD_SCRIPT.format(root=root, n=n, out=out)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "no-quad ok" in r.stdout, r.stdout + r.stderr
    assert np.array_equal(got, np.load(out).reshape(n, -1))
    # ops 9..12 as one launch (k_quad_mm.hip: both C = 64 pairs, three tensors through LDS) against the two pair launches of the child
    # process and the oracle; pieces that end inside it fall back to the pairs
    if any(k.startswith("quad_mm") for k in names):
        got12 = np.asarray(m.run_until(x, 12)).reshape(n, -1)
        assert np.array_equal(got12, np.load(out + ".12.npy").reshape(n, -1))
        for b in (0, 1, n - 1):
            _, layers = om.run_quantized(x[b], layers=True)
            assert np.array_equal(got12[b], layers[12].reshape(-1)), b
            for last in (9, 10, 11):
                assert np.array_equal(np.asarray(m.run_until(x[b:b + 1], last)).reshape(-1), layers[last].reshape(-1)), (b, last)
        for k in (1, 3):
            assert np.array_equal(np.asarray(m.run_until(x[:k], 12)).reshape(k, -1), got12[:k])
    # a single image and a batch smaller than the grid
    for k in (1, 5):
        assert np.array_equal(np.asarray(m.run_until(x[:k], 8)).reshape(k, -1), got[:k])
    # pieces that end inside the five-operator launch fall back to the quad / the pairs / the stem alone
    _, layers = om.run_quantized(x[2], layers=True)
    for last in (0, 2, 4):
        assert np.array_equal(np.asarray(m.run_until(x[2:3], last)).reshape(-1), layers[last].reshape(-1)), last


def test_routing_switches_need_the_master_switch():
    """A stray MF_NO_QUAD / MF_NO_TABLE / MF_NO_FMA_EPI in somebody's environment must not change which kernels the product runs:
    without MF_DEV=1 the library ignores every routing and tuning variable (csrc/switches.cpp); with it, they act."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import microflow_rs_amd as mf; m = mf.model(%r); m.prepare(1); "
            "print('KERNELS', [m.op(i)['kernel'] for i in (0, 5)], m.op_epilogue_mode(0))" % (root, model_path("person_detect")))
    base = {k: v for k, v in os.environ.items() if not k.startswith("MF_")}
    outs = {}
    for tag, extra in (("plain", {}), ("stray", {"MF_NO_QUAD": "1", "MF_NO_TABLE": "1", "MF_NO_FMA_EPI": "1"}),
                       ("dev", {"MF_DEV": "1", "MF_NO_QUAD": "1", "MF_NO_FMA_EPI": "1"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(base, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = [ln for ln in r.stdout.splitlines() if ln.startswith("KERNELS")][-1]
    assert outs["stray"] == outs["plain"] and "penta_rr" in outs["plain"] and "quad_rr" in outs["plain"] and outs["plain"].endswith(" 3")
    assert outs["dev"] != outs["plain"] and "quad_rr" not in outs["dev"] and "penta_rr" not in outs["dev"]


def test_kernel_routing(models):
    """The fast HIP kernels are the ones that run for person_detect."""
    if ROUTING_SWITCHED:
        pytest.skip("default routing only (switches set: %s)" % ", ".join(ROUTING_SWITCHED))
    m = models["person_detect"]
    m.prepare(1)
    names = [m.op(i)["kernel"] for i in range(m.num_ops)]
    penta = names[0].startswith("penta_rr")   # the stem + ops 1..4 in one launch (k_quad.hip STEM; MF_NO_PENTA=1: not)
    assert penta or names[0].startswith("dw3x3_stem8")
    # fused depthwise + 1x1 conv pairs: dwpw_rr / dwpw_mm (taps on the matrix pipe; MF_DWPW_IMPL=mm: dwpw_mm only);
    # the five 6x6x128 pairs (ops 13..22) are ONE persistent kernel (MF_NO_STAGE=1: not)
    # ops 1..4 and 5..8 are two "quads" (two pairs per launch, k_quad.hip; MF_NO_QUAD=1: four pair launches)
    quads = sum(n.startswith(("quad_rr", "penta_rr")) for n in names)
    quads_mm = sum(n.startswith("quad_mm") for n in names)   # ops 9..12 in one launch (k_quad_mm.hip; MF_NO_QUAD_MM=1 / MF_NO_QUAD=1: not)
    npairs = sum(n.startswith(("dwpw_rr", "dwpw_mm")) for n in names) + 2 * (quads + quads_mm)
    pair_tail = not os.environ.get("MF_NO_PAIRTAIL")
    front = names[23].startswith("pair_front_tail")   # ops 23..30 in one launch (k_tail3.hip FRONT; MF_NO_PAIR_FRONT=1 / MF_NO_PAIRTAIL=1: not)
    npairs += 2 * front
    if pair_tail and not any(os.environ.get(k) for k in ("MF_NO_PAIR_FRONT", "MF_NO_MAGIC", "MF_DWPW_IMPL", "MF_NO_TABLE")):
        assert front, names
    if not (os.environ.get("MF_NO_QUAD") or os.environ.get("MF_DWPW_IMPL")):
        assert quads == 2 and names[5].startswith("quad_rr<24,24,32"), names
        assert names[0].startswith("penta_rr<96,96,1,2,8|48,48,8") if penta else names[1].startswith("quad_rr<48,48,8"), names
        assert all(n.startswith("(fused") for n in names[2:5] + names[6:9]) and (not penta or names[1].startswith("(fused")), names
        if not os.environ.get("MF_NO_QUAD_MM"):
            assert quads_mm == 1 and names[9].startswith("quad_mm<12,12,64,1,64|12,12,64,2,128>") and all(n.startswith("(fused") for n in names[10:13]), names
        if not os.environ.get("MF_NO_PENTA"):
            assert penta, names
    if os.environ.get("MF_NO_STAGE"):
        assert npairs == (12 if pair_tail else 13), names
    else:
        assert npairs in (7, 8) and names[13].startswith("stage_6x6x128"), names
        assert all(n.startswith("(fused") for n in names[14:23]), names
        assert front or (names[23].startswith("dwpw") and names[25].startswith(("dwpw", "pair3_tail"))), names
    if not pair_tail:
        assert names[27] == "tail_pool_head_softmax<2>"               # pool + head conv + softmax
    else:  # the last pair (ops 25, 26) + the tail (27..30) in one launch
        assert names[23 if front else 25].startswith("pair_front_tail<6,6,128,2,256|3,3,256,2>" if front else "pair3_tail"), names
        assert all(n.startswith("(fused") or n == "" for n in names[(24 if front else 26):]), names
    assert names[28].startswith("(fused") and names[29] == "" and names[30].startswith("(fused")
    if not os.environ.get("MF_DWPW_IMPL"):
        assert sum(n.startswith("dwpw_rr") for n in names) + 2 * quads == 4
    m.set_fusion(False)
    names = [m.op(i)["kernel"] for i in range(m.num_ops)]
    m.set_fusion(True)
    assert sum(n.startswith(("dw3x3_mm", "dw3x3_nhwc")) for n in names) == 13  # matrix-pipe taps on the three large layers, v_dot4 on the rest
    assert sum(n.startswith("pw_mfma") for n in names) == 13
    assert names[28] == "conv1x1_rowwave" and names[27] == "avgpool_c4"
    assert names[29] == "" and names[30] == "softmax_table"
    # speech: [reshape] depthwise (one input channel) -> FullyConnected + Softmax in one launch
    sp = models["speech"]
    sp.prepare(1)
    names = [sp.op(i)["kernel"] for i in range(sp.num_ops)]
    if os.environ.get("MF_NO_DWFC"):
        assert "dw_c1_lds" in names and "fc_rowwave_softmax<4>" in names and names[-1].startswith("(fused"), names
    else:  # depthwise (matrix pipe) + FullyConnected + Softmax: the whole model in one launch
        assert names[1].startswith("dwc1_fc_softmax") and all(n.startswith("(fused") for n in names[2:]), names
    sp.set_fusion(False)
    names = [sp.op(i)["kernel"] for i in range(sp.num_ops)]
    sp.set_fusion(True)
    assert "fc_rowwave<4>" in names and names[-1] == "softmax_table", names


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 13, 64, 257])
def test_fused_equals_layerwise(models, n):
    """DW+PW fusion on/off must give identical int8 tensors at every operator boundary that
    still exists (i.e. after each pair), for batch sizes around every images-per-step value."""
    import torch
    m = models["person_detect"]
    x = synth_i8(3, 9000, n, m.input_elems).reshape((n,) + m.input_shape)
    for last in (2, 4, 6, 8, 10, 12, 14, 22, 24, 26, 27, 28, 30):
        m.set_fusion(True)
        a = m.run_until(x, last)
        m.set_fusion(False)
        b = m.run_until(x, last)
        m.set_fusion(True)
        assert np.array_equal(a, b), ("first mismatch after op", last, m.op(last - 1)["kernel"])
    # a depthwise op as the last op always runs unfused
    assert np.array_equal(m.run_until(x, 1), m.set_fusion(False).run_until(x, 1))
    m.set_fusion(True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("sine", 1), ("speech", 5), ("person_detect", 3)])
def test_graph_replay_equals_eager(name, batch):
    """mf_model_set_graph: the captured launch sequence gives the eager path's results, is only
    used for repeated (input, output, batch) triples, and is dropped when routing changes."""
    import torch
    mf = importlib.import_module("microflow_rs_amd")
    m = mf.model(model_path(name))
    rng = np.random.default_rng(5)
    x = torch.as_tensor(rng.integers(-128, 128, (batch, m.input_elems)).astype(np.int8)).cuda()
    want = m.run_quantized(x).clone()
    m.set_graph(True)
    out = torch.empty_like(want)
    for it in range(4):  # 1st call eager, 2nd captures + replays, then replays
        out.zero_()
        m.run_quantized(x, out=out)
        assert torch.equal(out, want), it
    assert m.graph_launches == 3
    # new input values in the SAME buffer: the replay reads the buffer, not a snapshot
    x2 = torch.as_tensor(rng.integers(-128, 128, (batch, m.input_elems)).astype(np.int8)).cuda()
    want2 = m.set_graph(False).run_quantized(x2).clone()
    m.set_graph(True)
    x.copy_(x2)
    for it in range(3):
        m.run_quantized(x, out=out)
        assert torch.equal(out, want2), it
    n = m.graph_launches
    # the f32 entry point is a different sequence (quantize + dequantize): its own capture
    xf = (x.float() - float(m.input_zero_point)) * float(m.input_scale)
    of = torch.empty((batch, m.output_elems), dtype=torch.float32, device="cuda")
    ref = m.set_graph(False).predict(xf).clone()
    m.set_graph(True)
    for it in range(3):
        m.predict(xf, out=of)
        assert torch.equal(of.reshape(ref.shape), ref), it
    assert m.graph_launches == n + 2
    # routing change invalidates the graph
    m.set_fusion(False)
    m.run_quantized(x, out=out)
    assert torch.equal(out, want2)


@pytest.mark.gpu
def test_graph_replay_of_a_fully_connected_with_a_weight_zero_point():
    """the row-sum pre-pass of a FullyConnected with wzp != 0 uses the operator's one scratch buffer, whose event
    handshake (ops.hip scratch_acquire / _release) must stay out of a stream capture: a captured wait on an event recorded outside
    the capture is not legal, and an event recorded into the graph would leave later eager waits looking at a stale record.
    Eager, captured + replayed, replayed again, then eager again on another stream -- all the same bytes."""
    import torch
    from tools.make_fc_model import synthetic_fc
    mf = importlib.import_module("microflow_rs_amd")
    M = 256
    m = mf.Model(synthetic_fc(M, 512, 256, wzp=-3, seed=9))
    rng = np.random.default_rng(6)
    x = torch.as_tensor(rng.integers(-128, 128, (1, m.input_elems)).astype(np.int8)).cuda()
    want = m.run_quantized(x).clone()
    assert "fc_mfma" in m.op(0)["kernel"] or "fc_" in m.op(0)["kernel"]
    m.set_graph(True)
    out = torch.empty_like(want)
    for it in range(4):
        out.zero_()
        m.run_quantized(x, out=out)
        assert torch.equal(out, want), it
    assert m.graph_launches == 3
    m.set_graph(False)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # an eager launch on another stream behind the replays: the handshake's event is a live one
        out2 = m.run_quantized(x).clone()
    side.synchronize()
    assert torch.equal(out2, want)


@pytest.mark.gpu
def test_host_fed_chunked_equals_device():
    """Host (numpy) batches above ~96 MB are cut into chunks whose H2D copies overlap compute;
    the results must equal the device-resident path's, for the int8 and the f32 entry points,
    including a ragged last chunk."""
    import torch
    mf = importlib.import_module("microflow_rs_amd")
    m = mf.model(model_path("person_detect"))
    n = 12000                                   # 110 MB of int8 input -> chunks of 7168 + 4832
    x = synth_i8(3, 0, n, m.input_elems)
    xd = torch.as_tensor(x).cuda()
    want = m.run_quantized(xd).cpu().numpy()
    got = m.run_quantized(x)
    assert isinstance(got, np.ndarray) and np.array_equal(got, want)
    nf = 3000                                   # 110 MB of f32 input -> chunks of 1792 + 1208
    xf = ((x[:nf].astype(np.float32) - np.float32(m.input_zero_point)) * m.input_scale).astype(np.float32)
    wantf = m.predict(torch.as_tensor(xf).cuda()).cpu().numpy()
    gotf = m.predict(xf)
    assert np.array_equal(gotf.view(np.uint32), wantf.view(np.uint32))
    assert np.array_equal(m.predict_quantized(x[:9000]).view(np.uint32),
                          m.predict_quantized(xd[:9000]).cpu().numpy().view(np.uint32))


@pytest.mark.gpu
def test_empty_batch_is_a_no_op(models):
    """Zero inferences: every entry point returns an empty result and launches nothing."""
    import torch
    m = models["speech"]
    e = m.input_elems
    assert m.run_quantized(np.zeros((0, e), np.int8)).shape == (0,) + tuple(m.output_shape)
    assert m.predict(np.zeros((0, e), np.float32)).shape == (0,) + tuple(m.output_shape)
    got = m.run_quantized(torch.zeros((0, e), dtype=torch.int8, device="cuda"))
    assert got.is_cuda and got.numel() == 0
    mf = importlib.import_module("microflow_rs_amd")
    op = mf.ops.prepare_softmax(1, 4, 0.1, 1 / 256, -128)
    assert op(np.zeros((0, 1, 4), np.int8)).shape == (0, 1, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,replicas", [("speech", 7, 3), ("person_detect", 5, 2), ("sine", 2, 4)])
def test_sharded_replicas_in_one_process(O, name, n, replicas):
    """mf_models_*: one host batch cut into contiguous shards over several prepared replicas, one
    host thread each (on an 8-GPU node: one replica per GPU; here all replicas share the one GPU,
    which exercises the same sharding, threading and error paths).  Ragged shards and more
    replicas than images included."""
    mf = importlib.import_module("microflow_rs_amd")
    reps = [mf.Model(model_path(name), device=0) for _ in range(replicas)]
    om = O.Model(model_path(name))
    rng = np.random.default_rng(n * replicas)
    xq = rng.integers(-128, 128, (n, reps[0].input_elems)).astype(np.int8)
    want = om.run_quantized_batch(xq)
    got = mf.run_sharded(reps, xq)
    assert np.array_equal(got.reshape(n, -1), want)
    pq = mf.run_sharded(reps, xq, "predict_quantized").reshape(n, -1)
    assert np.array_equal(pq.view(np.uint32), np.stack([om.predict_quantized(x).reshape(-1) for x in xq]).view(np.uint32))
    xf = ((xq.astype(np.float32) - np.float32(om.in_zp)) * om.in_scale).astype(np.float32)
    pf = mf.run_sharded(reps, xf, "predict").reshape(n, -1)
    assert np.array_equal(pf.view(np.uint32), np.stack([om.predict(x).reshape(-1) for x in xf]).view(np.uint32))
    # handles of different models are rejected
    other = mf.Model(model_path("sine" if name != "sine" else "speech"), device=0)
    other.prepare(1)
    with pytest.raises(mf.MicroflowError):
        mf._lib.check(mf.lib().mf_models_run_quantized(
            (__import__("ctypes").c_void_p * 2)(reps[0]._h, other._h), 2, xq.ctypes.data, n, got.ctypes.data))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 257, 4099])
def test_f32_predict_with_quantize_fused_into_the_stem(O, n):
    """M::predict on person_detect: with fusion on, the boundary quantize runs inside the first launch -- the five-operator
    kernel's f32 instance (k_quad.hip F32IN: the quantisation between two of its phases, in round-to-nearest inside a
    round-toward-zero kernel), or the stem kernel's staging under MF_NO_F32_GROUP / MF_NO_PENTA (scripts/switch_matrix.sh runs
    both); odd batches exercise the ragged last step, 4099 many steps per workgroup; with fusion off it
    is the separate quantize_f32 kernel.  All must equal the oracle's predict bit for bit,
    including values that quantize to ties and beyond the int8 range."""
    import torch
    mf = importlib.import_module("microflow_rs_amd")
    m = mf.model(model_path("person_detect"))
    om = O.Model(model_path("person_detect"))
    rng = np.random.default_rng(n)
    q = rng.integers(-140, 140, (n, m.input_elems)).astype(np.float32)       # some saturate
    xf = ((q - np.float32(om.in_zp)) * om.in_scale).astype(np.float32)
    xf[:, ::7] += np.float32(0.5) * om.in_scale                                # exact .5 ties in x / scale + zp
    xf[0, :4] = [np.nan, np.inf, -np.inf, 0.0]
    want = np.stack([om.predict(v).reshape(-1) for v in xf[: min(n, 8)]])
    got = m.predict(torch.as_tensor(xf).cuda()).cpu().numpy().reshape(n, -1)
    assert np.array_equal(got[: want.shape[0]].view(np.uint32), want.view(np.uint32))
    m.set_fusion(False)
    ref = m.predict(torch.as_tensor(xf).cuda()).cpu().numpy().reshape(n, -1)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    m.set_fusion(True)
    assert np.array_equal(m.predict(xf).reshape(n, -1).view(np.uint32), ref.view(np.uint32))   # host-fed path


@pytest.mark.parametrize("wzp", [0, -3], ids=["wzp0", "wzp-3"])
def test_fully_connected_4096_cubed_through_predict_inner(mf, O, wzp):
    """BASELINE config 5 at full size (M = K = N = 4096, the 256x256-tile staggered GEMM) through the
    model API, weight zero point 0 and != 0 (the x.1 / c3 terms of src/ops/fully_connected.rs:60-72).
    Rows 0..63 are crafted: row r is +127 where weight row r is positive and -128 elsewhere, so that
    accumulator (r, r) = 255 * sum of the positive weights ~ 3.3e7 > 2^24, where f32(acc) really rounds
    (v_cvt_f32_i32, to nearest even; src/ops/fully_connected.rs:68-72 `as f32`); bias[r] cancels that
    accumulator to within a few hundred and c1 = 1/8, so the rounding of the conversion is visible in the
    int8 output instead of disappearing in the saturation.  Rows 64..67 are all +127 / all -128 /
    alternating; the rest are uniform random.  128 rows are compared with the oracle."""
    import torch
    from tools.make_fc_model import fc_model
    M = K = N = 4096
    rng = np.random.default_rng(4096 + wzp)
    w = rng.integers(-128, 128, (N, K), dtype=np.int8)
    x = rng.integers(-128, 128, (M, K), dtype=np.int8)
    ncraft = 64
    x[:ncraft] = np.where(w[:ncraft] > wzp, 127, -128).astype(np.int8)
    x[64], x[65] = 127, -128
    x[66] = np.where(np.arange(K) % 2 == 0, 127, -128)
    x[67] = np.where(np.arange(K) % 2 == 0, -128, 127)
    izp = -128
    acc_diag = ((x[:ncraft].astype(np.int64) - izp) * (w[:ncraft].astype(np.int64) - wzp)).sum(axis=1)
    assert acc_diag.min() > (1 << 24)
    bias = rng.integers(-4096, 4096, N).astype(np.int64)
    bias[:ncraft] = -acc_diag + rng.integers(-300, 300, ncraft)
    in_s, w_s, out_s = 1.0 / 128, 1.0 / 128, 1.0 / 2048            # c1 = in_s * w_s / out_s = 1/8
    quants = ((in_s, izp), (w_s, wzp), (in_s * w_s, 0), (out_s, 3))
    m = mf.model(fc_model(M, K, N, w, bias.astype(np.int32), *quants))
    m.prepare(1)
    assert m.op(0)["kernel"] == "fc_mfma"
    xd = torch.from_numpy(x).cuda()
    y = m.run_quantized(xd.reshape((1, M, K))).reshape(M, N).cpu().numpy()
    rows = sorted(set(list(range(68)) + [int(r) for r in rng.integers(68, M, 60)]))
    om = O.Model(fc_model(len(rows), K, N, w, bias.astype(np.int32), *quants))
    want = om.run_quantized(x[rows]).reshape(len(rows), N)
    assert np.array_equal(y[rows], want), np.argwhere(y[rows] != want)[:5]
    # the crafted diagonal really sits in the unsaturated range (else the test would prove nothing)
    diag = y[np.arange(ncraft), np.arange(ncraft)]
    assert np.all(np.abs(diag.astype(int)) < 120) and len(np.unique(diag)) > 20
    # and truncating instead of rounding the int -> f32 conversion would change some of those outputs
    f_rne = acc_diag.astype(np.float32)
    f_trunc = (acc_diag - (acc_diag % 2)).astype(np.float32)       # ulp = 2 in [2^24, 2^25)
    if acc_diag.max() < (1 << 25):
        assert np.any(f_rne != f_trunc)


@pytest.mark.parametrize("wzp", [0, 5], ids=["wzp0", "wzp5"])
def test_fully_connected_mfma_every_row_at_256(mf, O, wzp):
    """fc_mfma (the 256 x 256 staggered-tile GEMM of BASELINE config 5) at M = 256, K = N = 4096: EVERY output row against the
    oracle (the 4096-row test above samples 128 rows; this one leaves none out, at a size the oracle finishes in seconds)."""
    import torch
    from tools.make_fc_model import fc_model
    M, K, N = 256, 4096, 4096
    rng = np.random.default_rng(77 + wzp)
    w = rng.integers(-128, 128, (N, K), dtype=np.int8)
    x = rng.integers(-128, 128, (M, K), dtype=np.int8)
    x[0], x[1] = 127, -128
    bias = rng.integers(-200000, 200000, N).astype(np.int32)
    quants = ((1.0 / 128, -128), (1.0 / 128, wzp), (1.0 / 16384, 0), (1.0 / 16, -5))   # c1 = 2^-10: outputs spread over the int8 range
    blob = fc_model(M, K, N, w, bias, *quants)
    m = mf.model(blob)
    m.prepare(1)
    assert m.op(0)["kernel"] == "fc_mfma"
    y = m.run_quantized(torch.from_numpy(x).cuda().reshape((1, M, K))).reshape(M, N).cpu().numpy()
    want = O.Model(blob).run_quantized(x).reshape(M, N)
    assert np.array_equal(y, want), np.argwhere(y != want)[:5]
    assert len(np.unique(want)) > 100                         # (not a saturated tensor)


def test_models_run_quantized_over_all_devices(mf, O):
    """Single-process sharding entry point (mf_models_run_quantized) over EVERY GPU of the box: one replica
    per device, contiguous shards, no collective.  Skips itself on a 1-GPU box (the multi-replica form on
    one device is covered by test_run_sharded_replicas)."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("one GPU visible")
    n = 64 * ndev + 5                                              # ragged over the devices
    base = mf.model(model_path("person_detect"))
    x = synth_i8(3, 4242, n, base.input_elems).reshape((n,) + base.input_shape)
    replicas = [mf.model(model_path("person_detect"), device=d) for d in range(ndev)]
    got = mf.run_sharded(replicas, x)
    want = O.Model(model_path("person_detect")).run_quantized_batch(x.reshape(n, -1))
    assert np.array_equal(got.reshape(n, -1), want)


def test_device_inputs_are_type_checked(mf):
    """A device tensor is passed by pointer, so a wrong dtype would be reinterpreted (or read out of
    bounds): predict wants float32, the quantized entry points the model's element type."""
    import torch
    m = mf.model(model_path("sine"))
    m.prepare(4)
    xf = torch.full((4, 1, 1), 0.5, device="cuda")
    assert m.predict(xf).shape[0] == 4
    for bad in (xf.half(), xf.double(), xf.to(torch.int32)):
        with pytest.raises(TypeError):
            m.predict(bad)
    xq = torch.zeros((4, 1, 1), dtype=torch.int8, device="cuda")
    assert m.run_quantized(xq).shape[0] == 4
    for bad in (xq.to(torch.int32), xq.float(), xq.to(torch.uint8)):
        with pytest.raises(TypeError):
            m.run_quantized(bad)
        with pytest.raises(TypeError):
            m.predict_quantized(bad)


@pytest.mark.parametrize("name,batch", [("sine", 3), ("speech", 1), ("speech", 21), ("person_detect", 1), ("person_detect", 5)])
def test_last_launch_writes_exactly_the_result(mf, name, batch):
    """mf_model_run_quantized with device buffers: the last launch writes the result straight into the caller's
    buffer (no copy).  Nothing outside [out, out + batch * output_elems) may be touched, whatever vector width the
    kernel stores with, and an unaligned output pointer must still work (copy path)."""
    import torch
    from microflow_rs_amd import _lib
    m = mf.Model(model_path(name), device=0)
    m.prepare(batch)
    L = _lib.lib()
    n_out = batch * m.output_elems
    x = torch.randint(-128, 128, (batch * m.input_elems,), dtype=torch.int8, device="cuda")
    want = m.run_quantized(x.reshape((batch,) + m.input_shape)).reshape(-1).cpu().numpy()
    for lead in (256, 3):  # 16-byte aligned (direct write) / unaligned (copy)
        buf = torch.full((lead + n_out + 256,), 0x5A, dtype=torch.int8, device="cuda")
        _lib.check(L.mf_model_run_quantized(m._h, x.data_ptr(), batch, buf.data_ptr() + lead, _lib.MF_MEM_DEVICE))
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        assert np.array_equal(got[lead:lead + n_out], want), (name, lead)
        assert (got[:lead] == 0x5A).all() and (got[lead + n_out:] == 0x5A).all(), (name, lead)


@pytest.mark.parametrize("name,batch", [("speech", 21), ("person_detect", 5)])
def test_unaligned_caller_input_pointer(mf, name, batch):
    """A device input pointer that is not 16-byte aligned (an offset into a larger allocation): the fast kernels read
    16-byte words, so the runtime realigns it instead of handing it to them."""
    import torch
    from microflow_rs_amd import _lib
    m = mf.Model(model_path(name), device=0)
    m.prepare(batch)
    L = _lib.lib()
    n_in = batch * m.input_elems
    x0 = torch.randint(-128, 128, (n_in,), dtype=torch.int8, device="cuda")
    big = torch.zeros(n_in + 64, dtype=torch.int8, device="cuda")
    want = m.run_quantized(x0.reshape((batch,) + m.input_shape)).reshape(-1)
    for lead in (3, 8):
        big[lead:lead + n_in] = x0
        out = torch.empty(batch * m.output_elems, dtype=torch.int8, device="cuda")
        _lib.check(L.mf_model_run_quantized(m._h, big.data_ptr() + lead, batch, out.data_ptr(), _lib.MF_MEM_DEVICE))
        torch.cuda.synchronize()
        assert torch.equal(out, want), (name, lead)


def test_single_launch_model_with_overlapping_buffers(mf):
    """A model that is ONE launch (a single FullyConnected GEMM) called with the output buffer ON TOP of the input
    buffer: the last launch normally writes straight into the caller's buffer, which here is still being read -- the
    runtime must detect the overlap and go through its own activation buffer."""
    import torch
    from microflow_rs_amd import _lib
    from tools.make_fc_model import synthetic_fc
    M = K = N = 512
    m = mf.model(synthetic_fc(M, K, N, wzp=0, seed=9))
    m.prepare(1)
    assert m.op(0)["kernel"] == "fc_mfma"
    L = _lib.lib()
    x = torch.randint(-128, 128, (M * K,), dtype=torch.int8, device="cuda")
    want = m.run_quantized(x.reshape(1, M, K)).reshape(-1).clone()
    for shift in (0, 4096):  # exactly in place / partially overlapping
        buf = torch.zeros(M * K + M * N, dtype=torch.int8, device="cuda")
        buf[shift:shift + M * K] = x
        _lib.check(L.mf_model_run_quantized(m._h, buf.data_ptr() + shift, 1, buf.data_ptr(), _lib.MF_MEM_DEVICE))
        torch.cuda.synchronize()
        assert torch.equal(buf[:M * N], want), shift


def test_boundary_quantisation_fast_division_is_exact(mf, O):
    """M::predict quantises x / scale + zp with a true division (src/quantize.rs:16-18).  The f32-input stem runs it
    as x * r + two fused corrections -- only after checking ALL 2^32 float inputs give the same byte.  Re-run that
    check for the three models' input parameters, show that it does catch a wrong reciprocal, and compare predict()
    with the oracle on inputs sitting on and next to every rounding boundary."""
    import ctypes as C
    from microflow_rs_amd import _lib
    L = _lib.lib()
    bad = C.c_uint64(123)
    for name in ("sine", "speech", "person_detect"):
        om = O.Model(model_path(name))
        scale, zp = np.float32(om.in_scale), int(om.in_zp)
        rcp = np.float32(1.0 / np.float64(scale))
        _lib.check(L.mf_verify_quant_div(0, C.c_float(scale), C.c_float(rcp), zp, 0, C.byref(bad)))
        assert bad.value == 0, (name, bad.value)
    _lib.check(L.mf_verify_quant_div(0, C.c_float(scale), C.c_float(rcp * np.float32(1.0001)), zp, 0, C.byref(bad)))
    assert bad.value > 0  # the checker is not vacuous
    # predict() of person_detect (f32 input, quantisation fused into the stem) on boundary inputs
    m = mf.Model(model_path("person_detect"), device=0)
    om = O.Model(model_path("person_detect"))
    scale, zp = np.float32(om.in_scale), np.float32(om.in_zp)
    ks = np.arange(-130, 131, dtype=np.float32) + np.float32(0.5)
    ties = ((ks - zp) * scale).astype(np.float32)
    cands = np.concatenate([ties, np.nextafter(ties, np.float32(np.inf)), np.nextafter(ties, np.float32(-np.inf)),
                            np.array([0.0, -0.0, 1e-40, -1e-40, 3e38, -3e38], dtype=np.float32)])
    rng = np.random.default_rng(5)
    x = rng.choice(cands, size=(4, m.input_elems)).astype(np.float32)
    got = m.predict(x.reshape((4,) + m.input_shape)).reshape(4, -1)
    want = np.stack([om.predict(v).reshape(-1) for v in x])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
