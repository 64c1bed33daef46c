#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> layer-wise step time and the pointwise kernels' times."""
import json
import sys

r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r["layerwise"]["ms_per_step"], r["layerwise"]["conv_2d"]["frac"], " ".join("%s=%.4f" % (k["kernel"][7:], k["ms"]) for k in r["layerwise"]["kernels"] if k["kernel"].startswith("pw_mfma") and k["op"] not in (16, 18, 20, 22)))
