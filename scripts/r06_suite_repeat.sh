#!/bin/bash
# GPU box: the whole -m gpu suite several times on the shipped library, then once per variant library (scripts/variants.py: e.g. the
# MF_JITTER_DIAG builds) -- a flake screen for the suite itself.   usage: REPS=4 bash scripts/r06_suite_repeat.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MF_ALLOW_DIAG_BUILD=1
for i in $(seq 1 ${REPS:-4}); do echo "default run $i: $(python -m pytest tests -m gpu -q 2>&1 | tail -1)"; done
python scripts/variants.py run "python -m pytest tests -m gpu -q 2>&1 | tail -1" 2>&1 | grep -v amdgpu.ids
