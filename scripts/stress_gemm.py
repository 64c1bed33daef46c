#!/usr/bin/env python3
"""Race screen for the staggered FullyConnected GEMM schedule: many launches of several shapes
(1..32 staging steps), every result compared bit for bit with the first one, and the first one
with the shape-generic kernel.  A staging race would show up as an unstable tile."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MF_FC_TILE", "256")
import torch  # noqa: E402
import microflow_rs_amd as mf  # noqa: E402

bad = 0
for (M, K, N, reps) in [(256, 128, 256, 200), (512, 256, 512, 200), (1024, 640, 768, 100), (2048, 2048, 2048, 60),
                        (4096, 4096, 4096, 40)]:
    rng = np.random.default_rng(M + K)
    x = torch.as_tensor(rng.integers(-128, 128, (M, K)).astype(np.int8)).cuda()
    w = rng.integers(-128, 128, (N, K)).astype(np.int8)
    c0 = rng.uniform(-3, 3, N).astype(np.float32)
    c1 = np.float32(127.0 / (74 * 74 * 3 * np.sqrt(K)))
    op = mf.ops.prepare_fully_connected(M, w, 0, 0.05, 3, mf.ops.FullyConnectedOptions(), (c0, c1, np.zeros(N, np.int32), 0))
    assert op.kernel == "fc_mfma"
    first = op(x).clone()
    for i in range(reps):
        y = op(x)
        if not torch.equal(y, first):
            bad += 1
            print("UNSTABLE", (M, K, N), "launch", i, int((y != first).sum()))
    op.set_generic(True)
    ok = torch.equal(op(x), first)
    print((M, K, N), reps, "launches stable;" if not bad else "", "equals generic kernel:", ok)
    bad += 0 if ok else 1
print("stress", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
