#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do
python scripts/variants.py run "python scripts/time_kernels.py 40"
for c in 0x101 0x801 0x102 0x802 0x104 0x804; do echo -n "cfg $c     "; MF_DQ_CFG=$c python scripts/time_kernels.py 40; done
done
