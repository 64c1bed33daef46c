#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> step time and the stage kernel's time."""
import json
import sys

r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r["ms_per_step"], "ok" if r["parity"]["bit_exact_vs_oracle"] else "MISMATCH", [(k["kernel"][:12], k["ms"]) for k in r["kernels"] if "stage" in k["kernel"] or "pair3" in k["kernel"]])
