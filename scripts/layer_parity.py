import sys, numpy as np
sys.path.insert(0, '/root/repo')
import microflow_rs_amd as mf
from oracle import oracle as O
from tests.synth import synth_i8
O.build()
m = mf.Model('/root/repo/models/person_detect.tflite'); m.prepare(64)
om = O.Model('/root/repo/models/person_detect.tflite')
x = synth_i8(3, 0, 16, om.in_elems)
outs = [om.run_quantized(x[b], layers=True)[1] for b in range(16)]
for last in (4, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26):
    lay = np.asarray(m.run_until(x, last))
    bad = 0; first = None
    for b in range(16):
        d = np.flatnonzero(lay[b].reshape(-1) != outs[b][last].reshape(-1))
        bad += d.size
        if d.size and first is None: first = (b, d[:6], lay[b].reshape(-1)[d[:6]], outs[b][last].reshape(-1)[d[:6]])
    print(last, m.op(last)['kernel'][:30], 'mismatches', bad, first)
