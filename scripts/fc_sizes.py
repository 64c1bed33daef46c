#!/usr/bin/env python3
"""GPU box: FullyConnected M = K = N in {4096, 8192} through mf_model_run_quantized (fc_mfma), HIP-event timed.
usage: fc_sizes.py [sizes ...]   (the PMC wrapper scripts/prof_fc_sizes.sh runs it under rocprofv3)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import microflow_rs_amd as mf  # noqa: E402
from microflow_rs_amd import _lib  # noqa: E402
from make_fc_model import synthetic_fc  # noqa: E402

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 8192]
out = {}
for n in sizes:
    m = mf.model(synthetic_fc(n, n, n, wzp=0, seed=5))
    m.prepare(1, device=0)
    L = _lib.lib()
    _lib.check(L.mf_model_set_stream(m._h, torch.cuda.current_stream().cuda_stream))
    x = torch.randint(-128, 128, (n, n), dtype=torch.int8, device="cuda")
    y = torch.empty(n * n, dtype=torch.int8, device="cuda")
    step = lambda: _lib.check(L.mf_model_run_quantized(m._h, x.data_ptr(), 1, y.data_ptr(), _lib.MF_MEM_DEVICE))  # noqa: E731
    for _ in range(5):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    med = ts[len(ts) // 2]
    out[str(n)] = {"kernel": m.op(0)["kernel"], "ms": round(med, 4), "TOPs": round(2.0 * n ** 3 / (med * 1e-3) / 1e12, 1),
                   "frac_of_5033": round(2.0 * n ** 3 / (med * 1e-3) / 1e12 / 5033.0, 4), "tiles_256x256": (n // 256) ** 2}
    del m, x, y
print(json.dumps(out))
