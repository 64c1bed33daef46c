#!/bin/bash
# round 6: full -m gpu suite, then the A/B variants of microflow_rs_amd/variants on the same box; `bash scripts/r06_g.sh <tag>`
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
T=${1:-g}
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/${T}_pytest.log 2>&1; echo "pytest exit $?" >> $OUT/${T}_pytest.log; tail -6 $OUT/${T}_pytest.log
for r in 1 2 3; do timeout 900 python scripts/variants.py run "python scripts/time_kernels.py 40"; done > $OUT/${T}_variants.txt 2>&1
cat $OUT/${T}_variants.txt
