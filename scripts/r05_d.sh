#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fma_epilogue.py tests/test_gpu_models.py tests/test_gpu_race.py -x -q -m gpu > $OUT/t.log 2>&1; tail -6 $OUT/t.log
MF_DEBUG_EPI=1 timeout 300 python -c "
import microflow_rs_amd as mf
m = mf.Model('models/person_detect.tflite'); m.prepare(64)
print([ (i, m.op_epilogue_mode(i)) for i in range(m.num_ops) if not m.op(i)['kernel'].startswith('(') and m.op(i)['kernel']])
" 2>&1 | grep -v "^\[epi\] \(conv\|depthwise\)" | tail -32
timeout 600 python scripts/time_kernels.py 30
timeout 600 python scripts/time_kernels.py 30
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-fed --no-extra > $OUT/bench.log 2> $OUT/bench.err; tail -c 1200 $OUT/bench.log
