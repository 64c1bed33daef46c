#!/usr/bin/env python3
"""GPU box: FullyConnected 4096^3 through mf_model_run_quantized: host enqueue time vs device time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import microflow_rs_amd as mf  # noqa: E402
from microflow_rs_amd import _lib  # noqa: E402
from make_fc_model import synthetic_fc  # noqa: E402

M = K = N = 4096
wzp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m = mf.model(synthetic_fc(M, K, N, wzp=wzp, seed=5))
m.prepare(1, device=0)
L = _lib.lib()
if os.environ.get("MF_OWN_STREAM"):
    own = torch.cuda.Stream()
    torch.cuda.set_stream(own)
_lib.check(L.mf_model_set_stream(m._h, torch.cuda.current_stream().cuda_stream))
print("stream handle", torch.cuda.current_stream().cuda_stream)
x = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda")
y = torch.empty(M * N, dtype=torch.int8, device="cuda")
step = lambda: L.mf_model_run_quantized(m._h, x.data_ptr(), 1, y.data_ptr(), _lib.MF_MEM_DEVICE)  # noqa: E731
for _ in range(20):
    step()
torch.cuda.synchronize()
for n in (50, 200):
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d host enqueue %.1f us/step, total %.1f us/step -> %.1f TOP/s  (%s)" % (
        n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, 2.0 * M * K * N / ((t2 - t0) / n) / 1e12, m.op(0)["kernel"]))
# the same steps inside ONE HIP event pair (round 6: does an event-bracketed region of n steps agree with the wall clock?)
for n in (20, 20, 20, 200, 1000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("region n=%d: events %.1f us/step, wall %.1f us/step" % (n, e0.elapsed_time(e1) * 1e3 / n, (t2 - t0) / n * 1e6))
