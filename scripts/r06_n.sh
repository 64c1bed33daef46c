#!/bin/bash
# round 6: dw_mm_rt layout A/B (planar tile / NHWC tile / knock-outs); `bash scripts/r06_n.sh <out-file>`
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp MF_ALLOW_DIAG_BUILD=1
F=$OUT/${1:-n_dwmm_variants.txt}
for r in 1 2; do timeout 1500 python scripts/variants.py run "python scripts/time_general_dw.py --line"; done > $F 2>&1
grep -v "^\[" $F | grep -v amdgpu.ids
