#!/usr/bin/env python3
"""GPU box: M::predict (f32 in HBM -> f32 out) against predict_quantized on person_detect, batch 65 536, median of HIP-event timings.
usage: time_f32.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import microflow_rs_amd as mf  # noqa: E402
from microflow_rs_amd import _lib  # noqa: E402
from microflow_rs_amd.model import synth_i8  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 65536
m = mf.model(os.path.join(ROOT, "models", "person_detect.tflite"))
m.prepare(B, device=0)
L = _lib.lib()
x = synth_i8(0x4D4643 + 3, 0, B * m.input_elems)
y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
xf = (x.reshape(B, -1).float() - float(m.input_zero_point)) * float(m.input_scale)
yf = torch.empty((B, m.output_elems), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()
_lib.check(L.mf_model_set_stream(m._h, s.cuda_stream))


def med(fn):
    fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def wall(fn, n=20):
    import time
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ti = med(lambda: _lib.check(L.mf_model_run_quantized(m._h, x.data_ptr(), B, y.data_ptr(), _lib.MF_MEM_DEVICE)))
tf = med(lambda: _lib.check(L.mf_model_predict(m._h, xf.data_ptr(), B, yf.data_ptr(), _lib.MF_MEM_DEVICE)))
yq = (yf / float(m.output_scale) + float(m.output_zero_point)).round().to(torch.int8).reshape(-1)
wi = wall(lambda: _lib.check(L.mf_model_run_quantized(m._h, x.data_ptr(), B, y.data_ptr(), _lib.MF_MEM_DEVICE)))
wf = wall(lambda: _lib.check(L.mf_model_predict(m._h, xf.data_ptr(), B, yf.data_ptr(), _lib.MF_MEM_DEVICE)))
print("wall, 20 calls back to back: int8 %.4f ms | f32 %.4f ms" % (wi, wf))
print("int8 %.4f ms (%.2f M/s) | f32 %.4f ms (%.2f M/s) | same outputs: %s" % (ti, B / ti / 1e3, tf, B / tf / 1e3, bool(torch.equal(yq, y))))
