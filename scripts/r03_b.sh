#!/bin/bash
# GPU box, round 3 call B: new epilogue ubench, stage kernel cycle stamps, SQ counters of the current build
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT/r03_b
export TMPDIR=/tmp
./scripts/ubench/epi_rate > $OUT/r03_b/epi_rate.txt 2>&1; cat $OUT/r03_b/epi_rate.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed --no-extra"
python scripts/variants.py run "$B 2>&1 >/dev/null | grep 'stage trace'" > $OUT/r03_b/stage_trace.txt 2>&1
cat $OUT/r03_b/stage_trace.txt
STEPS=sqpmc bash scripts/gpu_check.sh > $OUT/r03_b/sq.log 2>&1
tail -45 $OUT/r03_b/sq.log
