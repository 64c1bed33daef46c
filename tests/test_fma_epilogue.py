"""Epilogue mode 3 -- the single-fma requantisation (csrc/epi_fma.cpp, k_common.hpp).

The reference's tail  y = sat(roundf(fl(A + fl(S * f32(acc)))))  (src/ops/conv_2d.rs:93-98) is a staircase in acc; the library
replaces it by  v_cvt_pk_u8_f32(v_fma_f32(S', bits(acc + pivot), C'))  only where the two staircases are IDENTICAL on every
accumulator the channel can produce.  CPU tests: what the host search returns is checked here against an independent numpy
restatement of both forms (no code shared with the product).  GPU tests: the device's own instructions against the
two-rounding form, a negative control, and the models end to end.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT, ROUTING_SWITCHED, model_path

f32 = np.float32
M0 = 12582912  # 1.5 * 2^23 = the f32 whose bit pattern is 0x4B400000


def _lib():
    from microflow_rs_amd import _lib
    return _lib


def search(A, S, amin, amax, u8=False, patch=False):
    """(S', C', pivot) -- with patch=True (S', C', pivot, patch_acc, patch_delta) -- or None"""
    L = _lib().lib()
    s, c, d, found, pa, pd = C.c_float(), C.c_float(), C.c_int(), C.c_int(), C.c_longlong(), C.c_int()
    _lib().check(L.mf_fma_epilogue_search(C.c_float(A), C.c_float(S), int(u8), int(amin), int(amax), int(patch), C.byref(s), C.byref(c),
                                          C.byref(d), C.byref(pa), C.byref(pd), C.byref(found)))
    if not found.value:
        return None
    assert patch or pd.value == 0
    return (f32(s.value), f32(c.value), d.value, pa.value, pd.value) if patch else (f32(s.value), f32(c.value), d.value)


def check_host(A, S, amin, amax, s3, c3, d, pa=0, pd=0, u8=False):
    bad = C.c_uint64(1)
    _lib().check(_lib().lib().mf_fma_epilogue_check_host(C.c_float(A), C.c_float(S), int(u8), int(amin), int(amax), C.c_float(s3), C.c_float(c3),
                                                         int(d), int(pa), int(pd), C.byref(bad)))
    return bad.value


def check_device(A, S, amin, amax, s3, c3, d, pa=0, pd=0, u8=False):
    bad = C.c_uint64(1)
    _lib().check(_lib().lib().mf_selftest_fma_epilogue(0, C.c_float(A), C.c_float(S), int(u8), int(amin), int(amax), C.c_float(s3), C.c_float(c3),
                                                       int(d), int(pa), int(pd), C.byref(bad)))
    return bad.value


def ref_numpy(A, S, acc, off):
    """the reference's tail on an array of accumulators, in the u8 domain (individually rounded f32 operations; roundf = half away)"""
    x = f32(A) + f32(S) * acc.astype(np.float32)           # two roundings (numpy never fuses)
    r = np.sign(x) * np.floor(np.abs(x).astype(np.float64) + 0.5)
    lo, hi = (-128, 127) if off else (0, 255)               # i8 (then + 128) or u8
    return np.clip(r, lo, hi).astype(np.int64) + off


def fma_numpy(s3, c3, d, acc, pa=0, pd=0):
    """v_fma_f32 on the bit pattern + v_cvt_pk_u8_f32, both in round-toward-zero, emulated with exact integer arithmetic:
    S' = ms 2^es, C' = mc 2^ec, F an integer  ->  S' F + C' = (ms F 2^(es - e0) + mc 2^(ec - e0)) 2^e0 with e0 = min(es, ec);
    rounding that toward zero to f32 and truncating is the floor of the exact value wherever it is positive (integers below
    2^24 are f32 values, so the first rounding never crosses one); then saturation to [0, 255]"""
    def split(x):
        m, e = np.frexp(np.float64(x))                     # x = m 2^e, |m| in [0.5, 1)
        return int(m * (1 << 24)), int(e) - 24             # 24-bit integer mantissa (exact: x is an f32)
    ms, es = split(s3)
    mc, ec = split(c3)
    e0 = min(es, ec)
    assert es - e0 <= 40 and ec - e0 <= 40                 # python integers below, no overflow anyway
    if pd:                                                 # the one replaced accumulator
        acc = np.where(acc == pa, acc + pd, acc)
    F = acc.astype(object) + (M0 + d)
    num = F * (ms << (es - e0)) + (mc << (ec - e0))        # exact, as python ints
    if e0 >= 0:
        v = num * (1 << e0)
    else:
        v = np.array([int(n) >> (-e0) for n in num], dtype=object)   # floor division by 2^-e0 (python >> floors)
    return np.clip(np.array(v, dtype=object), 0, 255).astype(np.int64)


def full_range(A, S):
    """accumulators covering every output value plus a saturated margin on both sides"""
    lo = int(np.floor((-129.0 - float(A)) / float(S))) - 3
    hi = int(np.ceil((128.0 - float(A)) / float(S))) + 3
    lim = (1 << 22) - 2
    return max(lo, -lim), min(hi, lim)


def person_detect_channels(O, ops=(0, 1, 2, 4, 6, 8, 12, 13, 14, 20, 24, 26), per_op=6):
    om = O.Model(model_path("person_detect"))
    out = []
    for i in ops:
        c0, c1 = om.op_constants(i)[:2]
        ozp = om.ops[i]["out_zp"]
        n = len(c0)
        for c in sorted(set(int(round(t)) for t in np.linspace(0, n - 1, per_op))):
            out.append((i, c, f32(f32(ozp) + f32(c0[c])), f32(c1[c if c < len(c1) else 0])))
    return out


def test_search_results_hold_on_every_accumulator(O):
    """Whatever the search returns reproduces the reference on the WHOLE range it was asked for -- checked with numpy, not
    with the library's own evaluator -- and it finds a solution for most real channels even on the full output range
    (the library restricts the range to the accumulators the weights can produce, which only makes it easier)."""
    chans = person_detect_channels(O)
    found = 0
    for op, c, A, S in chans:
        amin, amax = full_range(A, S)
        r = search(A, S, amin, amax)
        if r is None:
            continue
        found += 1
        s3, c3, d = r
        acc = np.arange(amin, amax + 1, dtype=np.int64)
        want = ref_numpy(A, S, acc, 128)
        got = fma_numpy(s3, c3, d, acc)
        assert np.array_equal(want, got), (op, c, float(A), float(S), np.flatnonzero(want != got)[:4])
        assert abs(int(np.float32(s3).view(np.int32)) - int(np.float32(S).view(np.int32))) <= 64  # S' is S moved by ulps
    assert found >= 0.85 * len(chans), (found, len(chans))


def test_search_synthetic_constants():
    """ties on purpose (A and S dyadic), steep and shallow staircases, outputs that saturate on one side only.  "No solution"
    is a legitimate answer -- the operator then keeps the two-rounding form: dyadic constants produce exact ties, which the
    reference rounds away from zero on both sides of 0 and no single line does; other constants can have two near-ties the
    reference's roundings resolve in opposite directions -- but whatever IS returned must hold on the whole range."""
    found = 0
    cases = [(-127.5, 0.5), (-128.0, 0.25), (0.5, 1.0), (-3.25, 0.001953125), (-127.31, 0.0021), (-100.0, 0.31964308), (5.0, 0.007),
             (-64.03125, 0.015625), (-200.0, 0.004), (90.0, 0.02), (-131.7, 0.0123), (-150.2, 0.0031), (-128.9, 0.05), (-180.0, 0.0077)]
    for A, S in cases:
        amin, amax = full_range(f32(A), f32(S))
        r = search(A, S, amin, amax)
        if r is None:
            continue
        found += 1
        s3, c3, d = r
        acc = np.arange(amin, amax + 1, dtype=np.int64)
        assert np.array_equal(ref_numpy(f32(A), f32(S), acc, 128), fma_numpy(s3, c3, d, acc)), (A, S)
        assert check_host(A, S, amin, amax, s3, c3, d) == 0, (A, S)
    assert found >= 6, found


def test_search_u8_and_degenerate_inputs():
    amin, amax = -60000, 70000
    r = search(3.25, 0.0031, amin, amax, u8=True)      # u8: the reference's own domain, no +128
    assert r is not None
    acc = np.arange(amin, amax + 1, dtype=np.int64)
    assert np.array_equal(ref_numpy(f32(3.25), f32(0.0031), acc, 0), fma_numpy(*r, acc))
    assert search(1.0, 0.0, -100, 100) is None              # S must be positive and normal
    assert search(1.0, -0.01, -100, 100) is None
    assert search(float("nan"), 0.01, -100, 100) is None
    assert search(1.0, float("inf"), -100, 100) is None
    assert search(1.0, 0.01, -(1 << 22), 100) is None       # outside the bit-pattern accumulators
    # a narrow reachable range: few steps, a solution with a small pivot
    assert search(-20.0, 0.02, -50, 50) is not None


def test_a_perturbed_constant_is_not_accepted_by_the_host_check(O):
    """negative control of the host-side exhaustive check: nudging C' by a few f32 steps must move a step somewhere"""
    caught = tried = 0
    for op, c, A, S in person_detect_channels(O, ops=(2, 6, 12), per_op=4):
        amin, amax = full_range(A, S)
        r = search(A, S, amin, amax)
        if r is None:
            continue
        s3, c3, d = r
        for k in (-64, 64):
            c_bad = f32(np.int32(np.float32(c3).view(np.int32) + k).view(np.float32))
            tried += 1
            caught += check_host(A, S, amin, amax, s3, c_bad, d) > 0
    assert tried >= 8 and caught == tried, (caught, tried)


def test_one_patched_accumulator_rescues_the_channels_without_a_line(O):
    """Channels whose reference steps are not those of any line (two near-ties rounded apart) get a form with ONE accumulator
    replaced by its neighbour: over the sampled channels of person_detect every one then has a form, the patched ones are exactly
    those the plain search gave up on, and the patched form holds on the whole range (numpy, exact integers) -- while the same
    form WITHOUT its patch differs at exactly that one accumulator."""
    chans = person_detect_channels(O, ops=(10, 14, 16, 18, 20, 22, 24, 26), per_op=24)
    patched = 0
    for op, c, A, S in chans:
        amin, amax = full_range(A, S)
        plain = search(A, S, amin, amax)
        r = search(A, S, amin, amax, patch=True)
        assert r is not None, (op, c)
        s3, c3, d, pa, pd = r
        assert (pd != 0) == (plain is None), (op, c)
        acc = np.arange(amin, amax + 1, dtype=np.int64)
        want = ref_numpy(A, S, acc, 128)
        assert np.array_equal(want, fma_numpy(s3, c3, d, acc, pa, pd)), (op, c)
        assert check_host(A, S, amin, amax, s3, c3, d, pa, pd) == 0
        if pd:
            patched += 1
            assert amin <= pa <= amax and pd in (-1, 1)
            diff = np.flatnonzero(want != fma_numpy(s3, c3, d, acc))
            assert list(acc[diff]) == [pa], (op, c, acc[diff][:4], pa)
            assert check_host(A, S, amin, amax, s3, c3, d) == 1
    assert patched >= 3, patched


def test_search_property_random_constants():
    """Property test over random (A, S) and random reachable ranges: whatever the search returns -- with or without a patched
    accumulator -- equals the reference at every accumulator next to one of the reference's steps and at both ends of the range.
    Both maps are monotone staircases, so that is equality on the whole range (a step of the form inside a flat stretch of the
    reference would show at the stretch's last accumulator, which is checked).  The reference side is numpy f32, the form side exact
    python integers; neither uses the library's evaluators."""
    from hypothesis import given, settings, strategies as st

    stats = {"found": 0, "patched": 0, "none": 0}

    @settings(max_examples=120, deadline=None, derandomize=True)
    @given(a=st.floats(-300.0, 300.0, width=32), ls=st.floats(-13.0, 0.0), lo_frac=st.floats(0.0, 0.4), hi_frac=st.floats(0.6, 1.0),
           u8=st.booleans())
    def prop(a, ls, lo_frac, hi_frac, u8):
        A, S = f32(a), f32(2.0 ** ls)
        fmin, fmax = full_range(A, S) if not u8 else full_range(f32(A - 128.0), S)
        amin = int(fmin + lo_frac * (fmax - fmin))
        amax = int(fmin + hi_frac * (fmax - fmin))
        r = search(A, S, amin, amax, u8=u8, patch=True)
        if r is None:
            stats["none"] += 1
            return
        s3, c3, d, pa, pd = r
        stats["found"] += 1
        stats["patched"] += pd != 0
        acc = np.arange(amin, amax + 1, dtype=np.int64)
        want = ref_numpy(A, S, acc, 0 if u8 else 128)
        steps = np.flatnonzero(np.diff(want)) + 1                                   # first accumulator of every new output value
        pts = np.unique(np.clip(np.concatenate([[0, len(acc) - 1], steps - 1, steps, [max(pa - amin, 0)] if pd else []]).astype(np.int64),
                                0, len(acc) - 1))
        got = fma_numpy(s3, c3, d, acc[pts], pa, pd)
        assert np.array_equal(want[pts], got), (float(A), float(S), amin, amax, u8, acc[pts][want[pts] != got][:4])
        assert np.all(np.diff(want) >= 0)

    prop()
    assert stats["found"] >= 60, stats      # (most random constants have a form; "none" is legitimate, see above)


# ------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_cvt_pk_u8_f32_is_what_the_search_assumes():
    bad = C.c_uint64(1)
    _lib().check(_lib().lib().mf_selftest_cvt_pk(0, C.byref(bad)))
    # all 2^32 inputs, in both rounding modes: truncation when MODE.FP_ROUND = toward zero (the mode-3 kernels), round half to even in
    # the default mode; saturation to [0, 255], NaN -> 0, the other bytes of the destination kept
    assert bad.value == 0


@pytest.mark.gpu
def test_device_confirms_the_host_and_catches_a_wrong_constant(O):
    """the gate the library itself uses at mf_*_create: the kernels' own requant_pack4<3> on every accumulator of the range"""
    n = npatched = 0
    for op, c, A, S in person_detect_channels(O, ops=(0, 1, 2, 4, 6, 8, 12, 13, 14, 18, 20, 22, 24, 26), per_op=6):
        amin, amax = full_range(A, S)
        r = search(A, S, amin, amax, patch=True)
        assert r is not None
        s3, c3, d, pa, pd = r
        assert check_device(A, S, amin, amax, s3, c3, d, pa, pd) == 0, (op, c, float(A), float(S))
        if pd:  # the patch is load-bearing: without it the device sees exactly the one accumulator
            npatched += 1
            assert check_device(A, S, amin, amax, s3, c3, d) == 1
        c_bad = f32(np.int32(np.float32(c3).view(np.int32) + 48).view(np.float32))
        bad = check_device(A, S, amin, amax, s3, c_bad, d, pa, pd)
        assert bad > 0, (op, c)                               # negative control: a perturbed C' is caught
        # and the host's evaluator agrees with the device on how many accumulators the wrong constant moves
        assert check_host(A, S, amin, amax, s3, c_bad, d, pa, pd) == bad, (op, c)
        n += 1
    assert n >= 40 and npatched >= 1, (n, npatched)
    r = search(3.25, 0.0031, -60000, 70000, u8=True)
    assert check_device(3.25, 0.0031, -60000, 70000, r[0], r[1], r[2], u8=True) == 0


@pytest.mark.gpu
def test_person_detect_launch_modes_and_parity(O):
    """which launches of the fused step take the single-fma form, and that the step stays bit-exact at every layer"""
    import microflow_rs_amd as mf
    from tests.synth import synth_i8
    m = mf.Model(model_path("person_detect"))
    m.prepare(64)
    modes = {i: m.op_epilogue_mode(i) for i in range(m.num_ops)}
    if not ROUTING_SWITCHED:
        assert m.op(0)["kernel"].startswith("penta_rr") and modes[0] == 3      # ops 0..4: every channel of all five operators
        assert m.op(5)["kernel"].startswith("quad_rr") and modes[5] == 3       # ops 5..8
        # ops 9..12 in one launch (quad_mm): one channel of op 10 with a patched accumulator; as two dwpw_mm launches: both mode 3
        assert modes[9] == 3 and (m.op(9)["kernel"].startswith("quad_mm") or modes[11] == 3)
        assert m.op(13)["kernel"].startswith("stage_6x6x128") and modes[13] == 3   # ops 13..22: 14 patched channels in seven of the ten operators
        # op 24 / op 26 need more patches than the kernels' list holds: ops 23..30 in one launch (pair_front_tail) run the bit-pattern
        # form with the v_med3 clamp for all four convolutions; as two launches (MF_NO_PAIR_FRONT) ops 23..24 take the saturating pack
        assert (m.op(23)["kernel"].startswith("pair_front_tail") and modes[23] == 1) or (modes[23] == 2 and modes[25] <= 2)
    om = O.Model(model_path("person_detect"))
    x = synth_i8(3, 0, 64, om.in_elems)
    x[0] = -128
    x[1] = 127
    got = m.run_quantized(x)
    want = om.run_quantized_batch(x)
    assert np.array_equal(np.asarray(got).reshape(want.shape), want)
    for last in (0, 2, 4, 6, 8, 10, 12, 22, 24, 26):
        lay = m.run_until(x[:8], last)
        for b in range(8):
            _, outs = om.run_quantized(x[b], layers=True)
            assert np.array_equal(np.asarray(lay[b]).reshape(-1), outs[last].reshape(-1)), (last, b)
    m.set_fusion(False)                                     # layer-wise: dw3x3_mm / pw_mfma / the stem in mode 3 where they have it
    got = m.run_quantized(x)
    assert np.array_equal(np.asarray(got).reshape(want.shape), want)


@pytest.mark.gpu
def test_patched_accumulators_occur_and_are_replaced_in_the_fused_kernels():
    """The kernels' patch path (k_common.hpp epi_patch_apply: which tile, lane group and register hold the patched channel) only
    runs when a patched channel's accumulator takes its one special value -- about once in 10^5 values.  At batch 49 152 every
    patched channel of ops 10 .. 22 sees 1.7 .. 7 million accumulators, so the value occurs dozens of times per channel; the tensors
    behind the patched operators are compared in full, on the device, between the fused launches (dwpw_mm, the stage: single-fma
    form with patches) and the layer-wise kernels (two-rounding form for these operators, pinned to the oracle by the other tests).
    A patch applied to the wrong lanes, or not at all, shows as single bytes off by one (the test's teeth were checked by building
    with -DMF_EPI_PATCH_KO=1, which compiles the patch code out: it fails then)."""
    import torch
    import microflow_rs_amd as mf
    from microflow_rs_amd.model import synth_i8 as synth_dev
    from microflow_rs_amd.synth import SEED
    B = 49152
    m = mf.Model(model_path("person_detect"))
    m.prepare(B)
    if not ROUTING_SWITCHED:
        assert m.op_epilogue_mode(9) == 3 and m.op_epilogue_mode(13) == 3
    x = synth_dev(SEED + 3, 7 * m.input_elems, B * m.input_elems).reshape(B, -1)
    for last in (10, 14, 16, 17, 18, 19, 20, 22):
        m.set_fusion(True)
        a = m.run_until(x, last).clone()
        m.set_fusion(False)
        b = m.run_until(x, last)
        m.set_fusion(True)
        ndiff = int((a != b).sum().item())
        assert ndiff == 0, (last, ndiff)


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import microflow_rs_amd as mf
from tests.synth import synth_i8
for name, cfg, n in (("person_detect", 3, 4099), ("speech", 2, 257)):
    m = mf.Model({root!r} + "/models/" + name + ".tflite")
    m.prepare(n)
    x = synth_i8(cfg, 0, n, m.input_elems)
    y = np.asarray(m.run_quantized(x))
    from microflow_rs_amd.synth import layer_checksum
    print(name, m.op_epilogue_mode(0), int(layer_checksum(y.view(np.int8) if y.dtype != np.int8 else y)))
"""


@pytest.mark.gpu
def test_switching_the_form_off_changes_no_byte():
    """MF_NO_FMA_EPI=1 (every operator on the two-rounding forms) in a child process: the same outputs, different modes"""
    outs = []
    for env_extra in ({}, {"MF_DEV": "1", "MF_NO_FMA_EPI": "1"}):
        env = dict(os.environ)
        env.pop("MF_NO_FMA_EPI", None)
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l.split() for l in r.stdout.strip().splitlines()])
    a, b = outs
    assert [l[0] for l in a] == [l[0] for l in b] == ["person_detect", "speech"]
    assert [l[2] for l in a] == [l[2] for l in b]          # checksums
    if not ROUTING_SWITCHED:
        assert a[0][1] == "3" and b[0][1] in ("1", "2")
