#!/bin/bash
# round 5, run F: the f32 entry inside the five-operator launch (penta_rr F32IN)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_u8.py tests/test_fma_epilogue.py -x -q -m gpu > $OUT/t.log 2>&1; tail -6 $OUT/t.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-fed > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | python -c "
import json,sys
c=json.loads(sys.stdin.read()); print('int8', c['value'], c['ms_per_step'], 'f32', c.get('predict_f32'))"
