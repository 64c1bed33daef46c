#!/usr/bin/env python3
"""Small-batch latency of predict_inner on device-resident buffers: eager launches (one per
operator / fused group) vs hipGraph replay (mf_model_set_graph).  Prints one JSON line.

    python scripts/latency.py [--iters 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    args = ap.parse_args()
    import torch
    import microflow_rs_amd as mf
    rows = []
    for name, batches in (("sine", [1]), ("speech", [1, 16, 256]), ("person_detect", [1, 8, 64, 512])):
        m = mf.model(os.path.join(ROOT, "models", name + ".tflite"))
        for b in batches:
            x = torch.randint(-128, 128, (b, m.input_elems), dtype=torch.int8, device="cuda")
            out = torch.empty((b, m.output_elems), dtype=torch.int8, device="cuda")
            res = {}
            for mode in ("eager", "graph"):
                m.set_graph(mode == "graph")
                for _ in range(5):
                    m.run_quantized(x, out=out)
                torch.cuda.synchronize()
                # (a) back-to-back enqueue throughput; (b) per-call latency with a sync after each call
                t0 = time.perf_counter()
                for _ in range(args.iters):
                    m.run_quantized(x, out=out)
                torch.cuda.synchronize()
                thr = (time.perf_counter() - t0) / args.iters
                lat = []
                for _ in range(args.iters):
                    t0 = time.perf_counter()
                    m.run_quantized(x, out=out)
                    torch.cuda.synchronize()
                    lat.append(time.perf_counter() - t0)
                res[mode] = {"us_per_call_pipelined": round(thr * 1e6, 1),
                             "us_latency_median": round(float(np.median(lat)) * 1e6, 1)}
            launches = sum(1 for i in range(m.num_ops) if m.op(i)["kernel"] and not m.op(i)["kernel"].startswith("(fused"))
            rows.append({"model": name, "batch": b, "launches_per_predict": launches, **res})
            m.set_graph(False)
    print(json.dumps({"latency": rows}))


if __name__ == "__main__":
    main()
