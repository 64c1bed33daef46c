#!/bin/bash
# GPU box: parity tests with the matrix-pipe fused kernels (default), then an A/B of the two
# depthwise implementations (MF_DWPW_IMPL=valu is r01's) and a rocprofv3 kernel-stats pass.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02_ab
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
summ() {
python - "$1" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", r["value"], "ms/step", r["ms_per_step"], "parity", r["parity"]["bit_exact_vs_oracle"])
    for k in r["kernels"]:
        print("  %2d %-34s %8.4f ms %8.1f GB/s %.3f" % (k["op"], k["kernel"], k["ms"], k["GBps"], k["frac"]))
except Exception as e:
    print("bench parse failed", sys.argv[1], e)
PY
}
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed > $OUT/bench_mm.json 2> $OUT/bench_mm.err
echo "bench mm exit $?"; tail -3 $OUT/bench_mm.err; summ $OUT/bench_mm.json
MF_DWPW_IMPL=valu timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed > $OUT/bench_valu.json 2> $OUT/bench_valu.err
echo "bench valu exit $?"; tail -3 $OUT/bench_valu.err; summ $OUT/bench_valu.json
if [[ "$1" == *prof* ]]; then
  rm -rf $OUT/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -- \
      python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-host-fed > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err")
  echo "rocprof exit $?"
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -30 "$f"
fi
