#!/usr/bin/env python3
"""Race screen for the run-time-geometry path (chain_rt, dw3x3_stem_rt, pair3_tail<H,W,C,2>; the partition is measured at prepare()):
generated person_detect-shaped models, a ragged device-resident batch through each many times, every output checksum compared with
the first and with the layer-wise kernels'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import microflow_rs_amd as mf  # noqa: E402
import tflite_writer as tw  # noqa: E402
from microflow_rs_amd.model import checksum_i8, synth_i8  # noqa: E402

bad = 0
for side, width, batch, reps in ((128, 1.0, 9001, 60), (64, 1.0, 40003, 60), (96, 0.5, 20011, 60), (96, 0.25, 10007, 40), (80, 1.0, 7001, 40)):
    m = mf.model(tw.person_detect_like(np.random.default_rng(side), side, width))
    m.prepare(batch)
    x = synth_i8(4321, 0, batch * m.input_elems).reshape(batch, -1)
    out = torch.empty((batch, m.output_elems), dtype=torch.int8, device="cuda")
    sums = {}
    for fusion in (True, False):
        m.set_fusion(fusion)
        m.run_quantized(x, out=out)
        first = checksum_i8(out.reshape(-1))
        sums[fusion] = first
        for i in range(reps if fusion else 5):
            m.run_quantized(x, out=out)
            if checksum_i8(out.reshape(-1)) != first:
                bad += 1
                print("UNSTABLE", side, width, batch, "fusion", fusion, "run", i)
    if sums[True] != sums[False]:
        bad += 1
        print("MISMATCH fused vs layer-wise", side, width)
    m.set_fusion(True)
    print("%dx%d width %s batch %d: %d fused runs, checksum %016x (= layer-wise), kernels %s" % (
        side, side, width, batch, reps, sums[True], sorted({m.op(i)["kernel"].split("<")[0] for i in range(m.num_ops) if m.op(i)["kernel"] and not m.op(i)["kernel"].startswith("(")})))
print("stress", "FAILED" if bad else "ok")
