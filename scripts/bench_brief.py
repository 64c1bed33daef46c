#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> one short line (step time, parity, per-kernel ms)."""
import json
import sys

r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%.4f ms %s | %s" % (r["ms_per_step"], "ok" if r["parity"]["bit_exact_vs_oracle"] else "MISMATCH",
                           " ".join("%s=%.4f" % (k["kernel"].split("<")[0][-4:] + k["kernel"].split("<")[1][:9], k["ms"]) for k in r["kernels"])))
