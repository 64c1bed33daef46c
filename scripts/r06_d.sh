#!/bin/bash
# round 6: knock-out timing of the long launches (WRONG results on purpose: timing only); `bash scripts/r06_d.sh <out-file>`
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp MF_ALLOW_DIAG_BUILD=1
F=$OUT/${1:-d_knockouts.txt}
for r in 1 2; do timeout 1200 python scripts/variants.py run "python scripts/time_kernels.py 30"; done > $F 2>&1
cat $F
