#!/bin/bash
# round 6, GPU call F: LDS-DMA issued from inline asm in every kernel, two staging buffers in quad_rr<Quad57>: parity, then A/B
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/f_pytest.log 2>&1; echo "pytest exit $?" >> $OUT/f_pytest.log; tail -8 $OUT/f_pytest.log
for r in 1 2 3; do timeout 900 python scripts/variants.py run "python scripts/time_kernels.py 40"; done > $OUT/f_variants.txt 2>&1
cat $OUT/f_variants.txt
for r in 1 2; do timeout 900 python scripts/variants.py run "python scripts/time_kernels.py 20 layerwise"; done > $OUT/f_variants_layerwise.txt 2>&1
cat $OUT/f_variants_layerwise.txt
timeout 600 python scripts/variants.py run "python scripts/time_speech.py" > $OUT/f_variants_speech.txt 2>&1; cat $OUT/f_variants_speech.txt | tail -12
