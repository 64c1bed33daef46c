#!/usr/bin/env python3
"""Per-launch times of generated person_detect-shaped models (tools/tflite_writer.person_detect_like), fused and layer-wise.
    python scripts/time_generated.py [side:width ...]      default: 128:1.0 64:1.0 96:0.5"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import microflow_rs_amd as mf  # noqa: E402
import tflite_writer as tw  # noqa: E402
from microflow_rs_amd.model import synth_i8  # noqa: E402

cases = [a for a in sys.argv[1:] if ":" in a] or ["128:1.0", "64:1.0", "96:0.5"]
for c in cases:
    side, width = int(c.split(":")[0]), float(c.split(":")[1])
    blob = tw.person_detect_like(np.random.default_rng(side), side, width)
    m = mf.Model(blob, autotune=(os.environ.get("MF_TIME_GENERATED_AUTOTUNE", "1") != "0"))
    B = int(65536 * 96 * 96 / (side * side))
    m.prepare(B)
    x = synth_i8(9, 0, B * m.input_elems)
    y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
    for fusion in (True, False):
        m.set_fusion(fusion)
        total, per = m.time_device(x, y, B, warmup=2, iters=10)
        print("== %dx%d width %s batch %d %s: %.4f ms per step, %.2f M inf/s" % (side, side, width, B, "fused" if fusion else "layer-wise", total, B / total / 1e3))
        for i in range(m.num_ops):
            d = m.op(i)
            if d["kernel"] and not d["kernel"].startswith("(fused") and per[i] > 0:
                print("   %2d %-20s %-70s %8.4f ms  in %s" % (i, d["name"], d["kernel"][:70], per[i], "x".join(map(str, d["in_shape"]))))
