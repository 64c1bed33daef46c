#!/bin/bash
# GPU box, round 3 call A: epilogue microbenchmark, the GPU tests, and an A/B of the epilogue forms
# (default = modes 1/2 of k_common.hpp; MF_NO_SAT_PACK=1 = mode 1 only; lib_r02epi = round 2's form).
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03_a
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
./scripts/ubench/epi_rate > $OUT/epi_rate.txt 2>&1; cat $OUT/epi_rate.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed --no-extra"
python scripts/variants.py run "$B 2>/dev/null | python scripts/bench_brief.py" > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
MF_NO_SAT_PACK=1 $B 2>/dev/null | python scripts/bench_brief.py > $OUT/nosat.txt 2>&1
cat $OUT/nosat.txt
