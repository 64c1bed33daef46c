#!/bin/bash
# round 6, GPU call C: the -m gpu suite on the new sources (late loads in the stage kernel, counted wait in the quads, per-operator
# launch counters, capture-aware scratch handshake), then same-box A/B against the old forms.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/c_pytest.log 2>&1; echo "pytest exit $?" >> $OUT/c_pytest.log; tail -12 $OUT/c_pytest.log
for r in 1 2 3; do timeout 900 python scripts/variants.py run "python scripts/time_kernels.py 40"; done > $OUT/c_variants.txt 2>&1
cat $OUT/c_variants.txt
