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

print(json.dumps(bench.general_depthwise_record({"mf": mf, "torch": torch, "checker": O}), indent=1))
