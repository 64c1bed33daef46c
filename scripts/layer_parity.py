#!/usr/bin/env python3
"""GPU box: where does the fused person_detect step first differ from the oracle?  The end tensor of every fused launch for 16 images,
mismatch counts and the first few differing (index, got, want); with --where the mismatching output positions (y, x, channel)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microflow_rs_amd as mf  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.synth import synth_i8  # noqa: E402

O.build()
path = os.path.join(ROOT, "models", "person_detect.tflite")
m = mf.Model(path)
m.prepare(64)
om = O.Model(path)
x = synth_i8(3, 0, 16, om.in_elems)
outs = [om.run_quantized(x[b], layers=True)[1] for b in range(16)]
ends = [i - 1 for i in range(1, m.num_ops) if m.op(i)["kernel"] and not m.op(i)["kernel"].startswith("(fused")] + [m.num_ops - 1]
for last in ends:
    lay = np.asarray(m.run_until(x, last))
    bad, first = 0, None
    for b in range(16):
        got, want = lay[b].reshape(-1), outs[b][last].reshape(-1)
        d = np.flatnonzero(got != want)
        bad += d.size
        if d.size and first is None:
            shp = outs[b][last].shape
            pos = [tuple(int(v) for v in np.unravel_index(i, shp)) for i in d[:8]] if "--where" in sys.argv else d[:6].tolist()
            first = (b, pos, got[d[:8]].tolist(), want[d[:8]].tolist(), "of %d" % want.size)
    print(last, m.op(last)["kernel"][:40] or "(inside a fused launch)", "mismatches", bad, first)
