#!/bin/bash
# Race screens under timing perturbation (on the GPU box, through gpurun).  Variant libraries built here beforehand:
#   python scripts/variants.py build ALL jit2="-DMF_JITTER_DIAG=2" jit9="-DMF_JITTER_DIAG=9"
#   VARIANTS_KEEP=1 python scripts/variants.py build ALL ko="-DMF_SYNC_KO=1" kojit2="-DMF_SYNC_KO=1 -DMF_JITTER_DIAG=2"
# jit*: every wave sleeps a pseudo-random time around every barrier and in front of every LDS-DMA (k_common.hpp mf_jitter) -- results
# must not change.  ko*: the positive control -- wg_sync() without its explicit lgkmcnt wait, i.e. the build that was wrong once in
# ~20 quad_mm launches; the screen has to catch it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MF_ALLOW_DIAG_BUILD=1
python scripts/variants.py run "python scripts/stress_stages.py ${REPS:-150} > /tmp/ss.txt 2>&1; echo unstable runs \$(grep -c UNSTABLE /tmp/ss.txt) mismatches \$(grep -c MISMATCH /tmp/ss.txt); tail -1 /tmp/ss.txt; python scripts/stress_model.py 2>&1 | tail -1; python scripts/stress_generated.py 2>&1 | tail -1; for i in 1 2 3 4; do python -m pytest tests/test_gpu_race.py -q 2>&1 | tail -1; done" 2>&1 | grep -v "amdgpu.ids"
