#!/usr/bin/env python3
"""GPU box: per-launch times of the person_detect step (HIP events per launch, median of `iters` passes) -- the quick
A/B tool behind scripts/variants.py.  usage: time_kernels.py [iters] [layerwise]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import microflow_rs_amd as mf  # noqa: E402
from microflow_rs_amd.model import synth_i8  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
B = int(os.environ.get("MF_TIME_BATCH", "65536"))   # (per-launch times at another batch, e.g. to see what the 256 MB memory-side cache does)
m = mf.model(os.path.join(ROOT, "models", "person_detect.tflite"))
m.prepare(B, device=0)
if "layerwise" in sys.argv:
    m.set_fusion(False)
x = synth_i8(0x4D4643 + 3, 0, B * m.input_elems)
y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
avg, per = m.time_device(x, y, B, warmup=3, iters=iters)
ks = [(i, m.op(i)["kernel"], per[i]) for i in range(m.num_ops) if m.op(i)["kernel"] and not m.op(i)["kernel"].startswith("(fused")]
if "--json" in sys.argv:
    import json
    print(json.dumps({"ms_per_step": avg, "kernels": [{"op": i, "kernel": k, "name": m.op(i)["name"], "ms": t} for i, k, t in ks]}))
    sys.exit(0)
print("%.4f ms (sum of launches %.4f) | " % (avg, sum(k[2] for k in ks)) + " ".join("%d:%.4f" % (i, t) for i, _, t in ks))
