#!/bin/bash
# round 6, GPU call B: full -m gpu suite on the new default (asm tile-B stores + LDS stem operands in the five-operator launch, MF_DEV
# master switch, config-4 and eight-rank tests), then second-round variants (counted waits, raw barriers) on the same box.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/b_pytest.log 2>&1; echo "pytest exit $?" >> $OUT/b_pytest.log; tail -25 $OUT/b_pytest.log
for r in 1 2; do timeout 900 python scripts/variants.py run "python scripts/time_kernels.py 40"; done > $OUT/b_variants.txt 2>&1
cat $OUT/b_variants.txt
