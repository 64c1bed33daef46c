#!/usr/bin/env python3
"""GPU box: coordinate descent over the dynamic-step-queue configuration (k_common.hpp DynSteps) of each of the
queue launches of the person_detect step (four with the default routing -- ops 0-4, 5-8, 9-12, 13-22; the ops 23-30 launch walks statically -- six when the sweep of profiles/r06/u_tune_dq_resweep.txt was made: 9-10, 11-12 and 23-24 were launches of their own), scored by the WHOLE step's time (per-launch times move with the chip's
power state, so a launch is only ever judged inside the real mix).  All candidates of a round are timed interleaved."""
import os
import sys

os.environ["MF_DEV"] = "1"  # (the library ignores tuning switches without it)
os.environ["MF_DQ_TUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import microflow_rs_amd as mf  # noqa: E402
from microflow_rs_amd.model import synth_i8  # noqa: E402

CANDS = ["0x100", "0x101", "0x801", "0x102", "0x802", "0x104", "0x804"]
B = 65536
m = mf.model(os.path.join(ROOT, "models", "person_detect.tflite"))
m.prepare(B, device=0)
x = synth_i8(0x4D4643 + 3, 0, B * m.input_elems)
y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
NQ = int(os.environ.get("TUNE_DQ_LAUNCHES", "6"))  # queue launches per pass (dq_config calls): ops 0 5 9 11 13 23


def score(cfgs, reps=3):
    os.environ["MF_DQ_CFGS"] = ",".join(cfgs)
    ts = []
    for _ in range(reps):
        # 1 warm-up + 18 timed passes; both sweeps of time_device run whole passes, so the launch counter stays aligned
        avg, _ = m.time_device(x, y, B, warmup=1, iters=18, per_op=False)
        ts.append(avg)
    return float(np.median(ts))


cur = [sys.argv[1]] * NQ if len(sys.argv) > 1 else ["0"] * NQ
print("start", cur, "%.4f" % score(cur))
for sweep in range(2):
    for k in range(NQ):
        res = {}
        for rnd in range(3):
            for c in (CANDS if rnd % 2 == 0 else CANDS[::-1]):
                t = cur[:k] + [c] + cur[k + 1:]
                res.setdefault(c, []).append(score(t, reps=1))
        med = {c: float(np.median(v)) for c, v in res.items()}
        best = min(med, key=med.get)
        print("launch %d: " % k + " ".join("%s=%.4f" % (c, med[c]) for c in CANDS) + " -> " + best)
        cur[k] = best
    print("after sweep", sweep, cur, "%.4f" % score(cur))
