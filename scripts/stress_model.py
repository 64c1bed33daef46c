#!/usr/bin/env python3
"""Race screen for the LDS-DMA staged kernels: the same device-resident batch through person_detect
and speech many times (fused and layer-wise), every output checksum compared with the first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import microflow_rs_amd as mf  # noqa: E402
from microflow_rs_amd.model import checksum_i8, synth_i8  # noqa: E402

bad = 0
for name, batch, reps in (("person_detect", 8191, 150), ("person_detect", 65536, 30), ("speech", 4099, 300)):
    m = mf.model(os.path.join(ROOT, "models", name + ".tflite"))
    m.prepare(batch)
    x = synth_i8(1234, 0, batch * m.input_elems).reshape(batch, -1)
    out = torch.empty((batch, m.output_elems), dtype=torch.int8, device="cuda")
    for fusion in (True, False):
        m.set_fusion(fusion)
        m.run_quantized(x, out=out)
        first = checksum_i8(out.reshape(-1))
        for i in range(reps):
            m.run_quantized(x, out=out)
            c = checksum_i8(out.reshape(-1))
            if c != first:
                bad += 1
                print("UNSTABLE", name, batch, "fusion", fusion, "run", i)
        print(name, batch, "fusion" if fusion else "layer-wise", reps, "runs, checksum %016x" % first)
print("stress", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
