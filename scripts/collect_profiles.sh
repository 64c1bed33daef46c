#!/bin/bash
# After `STEPS="test smoke bench prof pmc sqpmc fcprof" bash scripts/gpu_check.sh` (through gpurun): copy what is judged from the
# scratch directory gpurun_out/ into profiles/<round>/ under a prefix, and refresh the two files bench.py replays fields from.
#   scripts/collect_profiles.sh r05 b
set -e
cd "$(dirname "$0")/.."
R=${1:?round directory, e.g. r05}; P=${2:?file prefix, e.g. b}
D=profiles/$R; mkdir -p $D
cp "$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)" $D/${P}_kernel_stats.csv
cp "$(find gpurun_out/prof_fc -name '*kernel_stats.csv' | head -1)" $D/${P}_fc4096_kernel_stats.csv
cp gpurun_out/pmc_traffic.json $D/${P}_pmc_traffic.json
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic_latest.json
cp gpurun_out/sq_latest.json $D/${P}_sq_latest.json
cp gpurun_out/sq_latest.json profiles/sq_latest.json
cp gpurun_out/sq_summary.csv $D/${P}_sq_counters_summary.csv
cp gpurun_out/bench_details_default.json $D/${P}_bench_details.json
tail -1 gpurun_out/bench.json > $D/${P}_bench_line.json
cp gpurun_out/smoke.log $D/${P}_smoke.log
cp gpurun_out/pytest_gpu.log $D/${P}_pytest_gpu.log
python3 - <<PY
import json, sys
sys.path.insert(0, ".")
from benchlib import roofline as bench
c = json.loads(open("$D/${P}_bench_line.json").read())
print("value %.1f %s, %.4f ms/step (event median %.4f); dominant %s %.4f ms, frac %.4f" % (
    c["value"], c["unit"], c["ms_per_step"], c["whole_step"]["ms"], c["roofline"]["kernel"], c["roofline"]["ms"], c["roofline"]["frac"]))
sha = bench.source_sha16()
for f in ("profiles/pmc_traffic_latest.json", "profiles/sq_latest.json"):
    s = json.load(open(f)).get("source_sha16")
    print(f, "source_sha16", s, "(current sources)" if s == sha else "(STALE: sources are %s)" % sha)
PY
