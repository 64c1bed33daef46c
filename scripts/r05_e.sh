#!/bin/bash
# round 5, run E: stage kernel with detect-and-redo patches; switches struct
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fma_epilogue.py tests/test_gpu_models.py tests/test_gpu_race.py tests/test_gpu_rt.py -x -q -m gpu > $OUT/t.log 2>&1; tail -6 $OUT/t.log
timeout 600 python scripts/time_kernels.py 30
timeout 600 python scripts/time_kernels.py 30
