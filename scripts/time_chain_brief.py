#!/usr/bin/env python3
"""one line per generated model: fused ms, sum of the chain launches, layer-wise ms (A/B tool for scripts/variants.py)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, microflow_rs_amd as mf, tflite_writer as tw
from microflow_rs_amd.model import synth_i8
for side, width in ((128, 1.0), (64, 1.0), (96, 0.5)):
    m = mf.model(tw.person_detect_like(np.random.default_rng(side), side, width))
    B = int(65536 * 96 * 96 / (side * side)); m.prepare(B)
    x = synth_i8(9, 0, B * m.input_elems); y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
    tot, per = m.time_device(x, y, B, warmup=2, iters=10)
    ch = [(m.op(i)["kernel"].split(";")[0][9:], per[i]) for i in range(m.num_ops) if m.op(i)["kernel"].startswith("chain_rt")]
    print("%dx%d w%s fused %.3f ms chains %.3f :" % (side, side, width, tot, sum(t for _, t in ch)), " ".join("%s=%.3f" % c for c in ch))
