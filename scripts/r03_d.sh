#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "fused:"; python scripts/time_kernels.py 30
echo "layerwise table:"; python scripts/time_kernels.py 30 layerwise
echo "layerwise rt (MF_NO_TABLE):"; MF_NO_TABLE=1 python scripts/time_kernels.py 30 layerwise
echo "layerwise generic (MF_NO_TABLE MF_NO_RT):"; MF_NO_TABLE=1 MF_NO_RT=1 python scripts/time_kernels.py 5 layerwise
