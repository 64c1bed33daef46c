#!/usr/bin/env python3
"""GPU box: speech.tflite step time (HIP events, median) at the given batch sizes + a parity spot check."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import microflow_rs_amd as mf  # noqa: E402
from microflow_rs_amd import _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402

path = os.path.join(ROOT, "models", "speech.tflite")
om = O.Model(path)
L = _lib.lib()
res = []
for B in [int(a) for a in sys.argv[1:]] or [4096, 65536]:
    m = mf.model(path)
    m.prepare(B, device=0)
    _lib.check(L.mf_model_set_stream(m._h, torch.cuda.current_stream().cuda_stream))
    x = torch.randint(-128, 128, (B * m.input_elems,), dtype=torch.int8, device="cuda")
    y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
    step = lambda: _lib.check(L.mf_model_run_quantized(m._h, x.data_ptr(), B, y.data_ptr(), _lib.MF_MEM_DEVICE))  # noqa: E731
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    n = 100
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    idx = list(range(0, B, B // 8 + 1))
    ok = np.array_equal(y.reshape(B, -1)[idx].cpu().numpy(), om.run_quantized_batch(x.reshape(B, -1)[idx].cpu().numpy()))
    res.append("B=%d %.2f us (min %.2f) %s" % (B, t[n // 2] * 1e3, t[0] * 1e3, "ok" if ok else "MISMATCH"))
print(" | ".join(res), m.op(1)["kernel"])
