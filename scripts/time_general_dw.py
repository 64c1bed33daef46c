#!/usr/bin/env python3
"""GPU box: bench.py's general_depthwise record alone (DepthwiseConv2D beyond 3x3 SAME: conv_mm_rt's depthwise mode)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import microflow_rs_amd as mf  # noqa: E402

from oracle import oracle as O  # noqa: E402

rec = bench.general_depthwise_record({"mf": mf, "torch": torch, "checker": O})
if "--line" in sys.argv:  # one line per run: the A/B form (scripts/variants.py run "python scripts/time_general_dw.py --line")
    print(" | ".join("%s %s %.4f ms %.3f of HBM %s" % (k, v["kernel"], v["ms"], v["hbm_frac"], "exact" if v["bit_exact_vs_oracle"] else "WRONG")
                     for k, v in rec.items()))
else:
    print(json.dumps(rec, indent=1))
