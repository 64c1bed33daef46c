#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel stats.
# Everything is written under gpurun_out/ (merged back by gpurun).
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
lscpu | egrep "Model name|^CPU\(s\)" >> $OUT/gpu.txt
STEPS=${STEPS:-all}

if [[ $STEPS == all || $STEPS == *test* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -25 $OUT/pytest_gpu.log
fi
if [[ $STEPS == all || $STEPS == *smoke* ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke exit $?" >> $OUT/smoke.log
  tail -3 $OUT/smoke.log
fi
if [[ $STEPS == all || $STEPS == *bench* ]]; then
  timeout 600 python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
  echo "bench exit $?"
  cp $OUT/bench_details.json $OUT/bench_details_default.json   # (the profiling passes below run bench.py again and overwrite it)
  tail -5 $OUT/bench.err
  python - <<'PY'
import json
try:
    line = open("gpurun_out/bench.json").read().strip().splitlines()[-1]
    c = json.loads(line)
    print("compact line: %d bytes; keys %s" % (len(line), sorted(c)))
    r = json.load(open("gpurun_out/bench_details_default.json"))
    print("value", r["value"], r["unit"], "ms/step", r["ms_per_step"], "roofline", c["roofline"]); print("whole_step", c["whole_step"]); print("ceiling", r["requant_ceiling"]); print("fc4096", c.get("fc4096"), c.get("fc4096_wzp")); print("speech", c.get("speech"))
    print("layerwise ms", r["layerwise"]["ms_per_step"], "depthwise", r["depthwise"], "conv", r["conv_2d"])
    for k in r["kernels"]:
        print("%2d %-18s %-28s %8.4f ms %8.1f GB/s hbm %.3f valu %.3f" % (k["op"], k["kind"], k["kernel"], k["ms"], k["GBps"], k["frac"], k["valu_frac"]))
    print("cpu", c["cpu_baseline"]); print("parity", c["parity"])
except Exception as e:
    print("bench parse failed", e)
PY
fi
# (--no-extra: the timed workload, its layer-wise table and the parity legs only -- the other sub-records of the default
# line launch their own kernels hundreds of times and would top the stats)
if [[ $STEPS == all || $STEPS == *prof* ]]; then
  rm -rf $OUT/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -- \
      python "$OLDPWD/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-host-fed --no-extra > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err")
  echo "rocprof exit $?"
  find $OUT/prof -name "*kernel_stats*" | head
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f"
fi
if [[ $STEPS == *pmc* && $STEPS != sqpmc ]]; then
  # HBM traffic counters, one --pmc pass per counter (FETCH_SIZE needs 3 of the 4 TCC slots,
  # WRITE_SIZE 2), with --kernel-trace only (MI355X_MICROARCH.md, HBM section)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_$c
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_$c" -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed > /dev/null 2> "$OLDPWD/$OUT/pmc_$c.err")
    echo "pmc $c exit $?"
    find $OUT/pmc_$c -name "*.csv" | head -5
  done
  f=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && head -3 "$f"
  python scripts/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/bench_details.json > $OUT/pmc_traffic.json
  python - <<'PY'
import json
for k in json.load(open("gpurun_out/pmc_traffic.json"))["kernels"]:
    if "traffic_over_algorithmic" in k: print("%-60s traffic %12d  x%.3f of algorithmic" % (k["kernel"][:60], k["traffic_bytes"], k["traffic_over_algorithmic"]))
PY
fi
if [[ $STEPS == *sqpmc* ]]; then
  # SQ counters (8 slots per pass) for the issue/stall breakdown of every kernel
  rm -rf $OUT/pmc_sq
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_sq" -- \
      python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed > /dev/null 2> "$OLDPWD/$OUT/pmc_sq.err")
  echo "sq pmc exit $?"
  rm -rf $OUT/pmc_sq2
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_sq2" -- \
      python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed > /dev/null 2> "$OLDPWD/$OUT/pmc_sq2.err")
  echo "sq2 pmc exit $?"
  # third pass: how busy the matrix pipes are (what the launches whose matrix-pipe and VALU time add up are bounded by)
  rm -rf $OUT/pmc_sq3
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_sq3" -- \
      python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra > /dev/null 2> "$OLDPWD/$OUT/pmc_sq3.err")
  echo "sq3 pmc exit $?"
  python - <<'PY'
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in ("gpurun_out/pmc_sq", "gpurun_out/pmc_sq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mf::k::" in r["Kernel_Name"]:
                m = re.search(r"mf::k::([A-Za-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
                acc[m.group(1) + (m.group(2) or "").replace(" ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_INSTS_SALU","SQ_INSTS_VMEM","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_VALU_CVT","GRBM_GUI_ACTIVE"]
with open("gpurun_out/sq_summary.csv", "w") as o:
    o.write("kernel," + ",".join(names) + "\n")
    for k in sorted(acc):
        o.write('"' + k + '",' + ",".join("%.0f" % (sum(acc[k][n]) / max(len(acc[k][n]), 1)) for n in names) + "\n")
print(open("gpurun_out/sq_summary.csv").read())
PY
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  python scripts/pmc_summary.py --sq $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_sq3 $f > $OUT/sq_latest.json
  head -c 600 $OUT/sq_latest.json
fi
if [[ $STEPS == *fcprof* ]]; then
  # kernel stats of the 4096^3 FullyConnected workload alone (BASELINE config 5)
  rm -rf $OUT/prof_fc
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_fc" -- \
      python "$OLDPWD/bench.py" --workload fc4096 --steps 50 --warmup 5 > "$OLDPWD/$OUT/prof_fc_bench.json" 2> "$OLDPWD/$OUT/prof_fc.err")
  echo "fc rocprof exit $?"
  f=$(find $OUT/prof_fc -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -i "fc_mfma\|fc_rowsum\|Name" "$f" | cut -c1-200
fi
