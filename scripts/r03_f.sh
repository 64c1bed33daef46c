#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo -n "base      "; python scripts/time_kernels.py 30 2>&1 | grep -v amdgpu
for i in 0 1 2 3 4 5 6 7 8; do echo -n "dwmm alt $i "; MF_DWMM_ALT=$i python scripts/time_kernels.py 30 2>&1 | grep -v amdgpu; done
for i in 0 1; do echo -n "dwrr alt $i "; MF_DWRR_ALT=$i python scripts/time_kernels.py 30 2>&1 | grep -v amdgpu; done
echo -n "base      "; python scripts/time_kernels.py 30 2>&1 | grep -v amdgpu
