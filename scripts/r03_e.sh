#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 800 python -m pytest tests/test_gpu_rt.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -12
echo "stem dw_c1:"; MF_NO_TABLE=1 python scripts/time_kernels.py 20 layerwise 2>&1 | grep -v amdgpu | cut -c1-70
echo "stem rows:"; MF_DW_C1=rows MF_NO_TABLE=1 python scripts/time_kernels.py 20 layerwise 2>&1 | grep -v amdgpu | cut -c1-70
