#!/bin/bash
# round 6, GPU call A: the quad kernels' LDS-store / stem-operand / counted-wait variants on one box, parity of the new default,
# and the v_cvt_pknorm_i16_f32 probe.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_race.py tests/test_fma_epilogue.py -m gpu -x -q > $OUT/a_pytest.log 2>&1; echo "pytest exit $?" >> $OUT/a_pytest.log; tail -5 $OUT/a_pytest.log
for r in 1 2; do timeout 900 python scripts/variants.py run "python scripts/time_kernels.py 40"; done > $OUT/a_variants.txt 2>&1
cat $OUT/a_variants.txt
timeout 300 python scripts/variants.py run "python scripts/time_f32.py 20" > $OUT/a_variants_f32.txt 2>&1; cat $OUT/a_variants_f32.txt
timeout 300 scripts/ubench/pknorm_probe > $OUT/a_pknorm_probe.txt 2>&1; cat $OUT/a_pknorm_probe.txt
