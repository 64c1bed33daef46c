#!/usr/bin/env python3
"""GPU box: person_detect-shaped models with random weights / scales / zero points (tools/tflite_writer.person_detect_like,
96x96, width 1.0: the shapes of the penta / quad / stage / tail kernels) against the oracle, every layer, several seeds and
batch sizes.  usage: fuzz_pd.py [seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import microflow_rs_amd as mf  # noqa: E402
import tflite_writer as tw  # noqa: E402
from oracle import oracle as O  # noqa: E402

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for seed in range(nseeds):
    elem = tw.UINT8 if seed % 4 == 3 else tw.INT8
    blob = tw.person_detect_like(np.random.default_rng(1000 + seed), 96, 1.0, elem, False, 5)
    m, om = mf.Model(blob), O.Model(blob)
    rng = np.random.default_rng(seed)
    lo, hi = (0, 256) if m.dtype == np.uint8 else (-128, 128)
    n = [1, 3, 17, 70][seed % 4]
    xq = rng.integers(lo, hi, (n, m.input_elems)).astype(m.dtype)
    xq[0] = hi - 1
    if n > 1:
        xq[1] = lo
    got = m.run_quantized(xq).reshape(n, -1)
    want = om.run_quantized_batch(xq)
    names = [m.op(i)["kernel"] for i in range(m.num_ops)]
    ok = np.array_equal(got, want)
    # the f32 entry (M::predict: the boundary quantisation inside the first launch) on the same images, bit for bit
    xf = ((xq.astype(np.float32) - np.float32(om.in_zp)) * om.in_scale).astype(np.float32)
    pf = np.asarray(m.predict(xf.reshape((n,) + tuple(m.input_shape)))).reshape(n, -1)
    wf = np.stack([om.predict(v).reshape(-1) for v in xf[: min(n, 6)]])
    ok = ok and np.array_equal(pf[: wf.shape[0]].view(np.uint32), wf.view(np.uint32))
    modes = sorted(set(m.op_epilogue_mode(i) for i in range(m.num_ops) if names[i] and not names[i].startswith("(")))
    if not ok:  # localise
        _, layers = om.run_quantized(xq[0], layers=True)
        for i, lay in enumerate(layers):
            g = np.asarray(m.run_until(xq[0:1], i)).reshape(-1)
            if not np.array_equal(g, lay.reshape(-1)):
                print("seed", seed, "first differing layer", i, names[i])
                break
        bad += 1
    print("seed %d %s batch %d modes %s: %s (%s, %s)" % (seed, "u8" if elem == tw.UINT8 else "i8", n, modes, "ok" if ok else "MISMATCH", names[0].split("<")[0], names[5].split("<")[0]))
print("fuzz", "ok" if not bad else "FAILED %d" % bad)
sys.exit(1 if bad else 0)
