#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/.

Run in the build container only (it reads DATA files of the mounted reference;
/root/reference does not exist on the GPU box, and nothing in tests/ reads it
at run time).  Three kinds of fixture are produced:

1. reference-held data, copied/transcribed as data (inputs + expected outputs):
   - sine_microflow.csv      <- analysis/accuracy/data/sine-microflow.csv (500 recorded
                                reference predict() outputs)
   - samples.npz             <- the numeric literals of samples/features/{person_detect,speech}.rs
                                (PERSON, NO_PERSON 96x96x1 i8; YES, NO 1x1960 i8)
   reference_kats.json is hand-transcribed from the reference's #[test] constants
   (each entry cites file:line) and is NOT produced by this script.

2. oracle-generated vectors (labelled as such; the oracle itself is pinned by (1)
   and reference_kats.json in tests/test_oracle_golden.py):
   - oracle_vectors.npz      per-model: seeded inputs' final int8 outputs, per-layer
                                checksums, and outputs for the four samples.

Usage:  python tests/golden/make_fixtures.py [--reference /root/reference]
"""
import argparse
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.synth import synth_i8, layer_checksum  # noqa: E402


def parse_rs_const(text, name):
    """All integer literals of `pub const NAME: ... = [matrix![ ... ]]` in row-major order."""
    m = re.search(r"pub const %s\b[^=]*=\s*\[?\s*matrix!\[" % re.escape(name), text)
    if not m:
        raise KeyError(name)
    start = m.end()
    depth, end = 1, start
    while depth:  # find the bracket closing `matrix![`
        ch = text[end]
        depth += (ch == "[") - (ch == "]")
        end += 1
    body = text[start:end - 1]
    vals = [int(v) for v in re.findall(r"-?\d+", body)]
    return np.array(vals, dtype=np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref = args.reference

    # 1a. the 500 recorded sine outputs (data file held by the reference)
    shutil.copyfile(os.path.join(ref, "analysis/accuracy/data/sine-microflow.csv"),
                    os.path.join(HERE, "sine_microflow.csv"))

    # 1b. the sample tensors
    pd = open(os.path.join(ref, "samples/features/person_detect.rs")).read()
    sp = open(os.path.join(ref, "samples/features/speech.rs")).read()
    samples = {}
    for name in ("PERSON", "NO_PERSON"):
        v = parse_rs_const(pd, name)
        assert v.size == 96 * 96, (name, v.size)
        assert v.min() >= -128 and v.max() <= 127
        samples[name] = v.astype(np.int8).reshape(96, 96, 1)
    for name in ("YES", "NO"):
        v = parse_rs_const(sp, name)
        assert v.size == 1960, (name, v.size)
        assert v.min() >= -128 and v.max() <= 127
        samples[name] = v.astype(np.int8).reshape(1, 1960)
    np.savez_compressed(os.path.join(HERE, "samples.npz"), **samples)

    # 2. oracle-generated vectors
    from oracle import oracle as O
    out = {}
    cfg = {"sine": (64, 1), "speech": (32, 2), "person_detect": (16, 3)}
    for model, (n, cfg_idx) in cfg.items():
        m = O.Model(os.path.join(ROOT, "models", model + ".tflite"))
        x = synth_i8(cfg_idx, 0, n, m.in_elems)
        finals = np.empty((n, m.out_elems), np.int8)
        sums = np.zeros((n, m.num_ops), np.uint64)
        for i in range(n):
            o, layers = m.run_quantized(x[i], layers=True)
            finals[i] = o
            for k, lay in enumerate(layers):
                sums[i, k] = layer_checksum(lay)
        out[model + "_n"] = np.array([n, cfg_idx])
        out[model + "_final"] = finals
        out[model + "_layer_checksums"] = sums
    pm = O.Model(os.path.join(ROOT, "models", "person_detect.tflite"))
    sm = O.Model(os.path.join(ROOT, "models", "speech.tflite"))
    for name in ("PERSON", "NO_PERSON"):
        out["sample_" + name] = pm.run_quantized(samples[name])
        out["sample_" + name + "_f32"] = pm.predict_quantized(samples[name])
    for name in ("YES", "NO"):
        out["sample_" + name] = sm.run_quantized(samples[name])
        out["sample_" + name + "_f32"] = sm.predict_quantized(samples[name])
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    for k in sorted(out):
        if k.startswith("sample_"):
            print(k, out[k].reshape(-1))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
