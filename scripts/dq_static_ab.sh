#!/bin/bash
# the whole person_detect step (and the generated models' chains) under different static prefixes of the step queue (MF_DQ_STATIC =
# fraction of every workgroup's share that is a stride walk; 0 = round 3's two static steps), one box, two rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for round in 1 2; do
  for f in 0 0.3 0.6 0.8 0.9; do
    echo "MF_DQ_STATIC=$f  $(MF_DQ_STATIC=$f python scripts/time_kernels.py 30 2>/dev/null)"
  done
done
for f in 0 0.6 0.8; do
  echo "== chains MF_DQ_STATIC=$f"; MF_DQ_STATIC=$f MF_CHAIN_DQ_AUTO=1 python scripts/time_chain_brief.py 2>/dev/null | cut -c1-400
done
