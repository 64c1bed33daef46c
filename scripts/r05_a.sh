#!/bin/bash
# round 5, first GPU pass: the single-fma epilogue -- selftests, parity, microbenchmark, bench
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fma_epilogue.py -x -q -m gpu > $OUT/t_fma.log 2>&1; echo "fma tests rc $?" >> $OUT/t_fma.log
tail -5 $OUT/t_fma.log
MF_DEBUG_EPI=1 timeout 300 python -c "
import microflow_rs_amd as mf
m = mf.Model('models/person_detect.tflite'); m.prepare(64)
for i in range(m.num_ops): print(i, m.op(i)['kernel'], m.op_epilogue_mode(i))
" > $OUT/modes.log 2>&1
tail -40 $OUT/modes.log
(cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 epi_rate.hip -o epi_rate 2>/dev/null; ./epi_rate) > $OUT/epi_rate.txt 2>&1
cat $OUT/epi_rate.txt
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc $?"
tail -c 3000 $OUT/bench.log
cp bench_details.json $OUT/ 2>/dev/null
MF_NO_FMA_EPI=1 timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-fed --no-extra > $OUT/bench_nofma.log 2> $OUT/bench_nofma.err
tail -c 600 $OUT/bench_nofma.log
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_race.py -x -q -m gpu > $OUT/t_models.log 2>&1; tail -5 $OUT/t_models.log
