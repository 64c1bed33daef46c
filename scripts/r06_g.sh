#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/g_pytest.log 2>&1; echo "pytest exit $?" >> $OUT/g_pytest.log; tail -4 $OUT/g_pytest.log
for r in 1 2 3; do timeout 900 python scripts/variants.py run "python scripts/time_kernels.py 40"; done > $OUT/g_variants.txt 2>&1
cat $OUT/g_variants.txt
