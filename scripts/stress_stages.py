#!/usr/bin/env python3
"""Race screen for the multi-phase kernels (barriers, LDS buffers reused across phases and steps): the same
device-resident batch is run many times up to the END of every fused group -- where a whole intermediate tensor is
visible, not just the 2 output bytes -- and every checksum is compared with the first run's.  Batches are ragged
(not a multiple of any step size) and large enough for every workgroup to run many steps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import microflow_rs_amd as mf  # noqa: E402
from microflow_rs_amd.model import checksum_i8, synth_i8  # noqa: E402

bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
PD = (0, 2, 4, 6, 8, 10, 12, 22, 24, 30)
# (several batch sizes: the step count per workgroup sets the step-queue configuration and the timing of the kernels' tails -- round 6's
# missing wait in front of a step-top barrier showed at 16 389 images once in ~20 launches and never at 32 781)
for name, batch, lasts in (("person_detect", 32768 + 13, PD), ("person_detect", 16384 + 5, PD), ("person_detect", 4096 + 3, PD), ("speech", 65536 + 5, (3,))):
    m = mf.model(os.path.join(ROOT, "models", name + ".tflite"))
    m.prepare(batch)
    x = synth_i8(99, 0, batch * m.input_elems).reshape((batch,) + m.input_shape)
    for last in lasts:
        first = None
        for i in range(reps):
            y = m.run_until(x, last)
            c = checksum_i8(y.reshape(-1))
            if first is None:
                first = c
            elif c != first:
                bad += 1
                print("UNSTABLE", name, batch, "until op", last, "run", i)
        m.set_fusion(False)  # ... and the operator-by-operator kernels must produce the same tensor
        ref = checksum_i8(m.run_until(x, last).reshape(-1))
        m.set_fusion(True)
        if ref != first:
            bad += 1
            print("MISMATCH fused vs layer-wise", name, "until op", last)
        print(name, batch, "until op", last, reps, "runs, checksum %016x" % first, "(= layer-wise)" if ref == first else "")
print("stress", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
