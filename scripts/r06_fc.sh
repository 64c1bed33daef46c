#!/bin/bash
# round 6: FullyConnected 4096^3 with a weight zero point -- row sums in the GEMM's own prologue against the pre-pass launch and the in-loop fold
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "fc or fully or gemm or zero_point" 2>&1 | tail -4
for r in 1 2 3; do
  echo "wzp 0:";                python scripts/time_fc.py 0  2>&1 | grep "n=200"
  echo "wzp -3 prologue:";      python scripts/time_fc.py -3 2>&1 | grep "n=200"
  echo "wzp -3 pre-pass:";      MF_DEV=1 MF_FC_ROWSUM_PREPASS=1 python scripts/time_fc.py -3 2>&1 | grep "n=200"
  echo "wzp -3 in-loop fold:";  MF_DEV=1 MF_FC_ROWSUM_FOLD=1 python scripts/time_fc.py -3 2>&1 | grep "n=200"
done
