#!/bin/bash
# hipcc -S of every kernel file, then scripts/asm_barrier_waits.py over the listings: prints the kernels with a barrier that a wave
# can reach with LDS operations of its own outstanding.  (k_gemm.hip's fc_mfma is expected there: its main loop runs a counted-wait
# schedule with bare s_barrier, argued in the comment block above the loop.)   usage: scripts/scan_barriers.sh [outdir]
cd "$(dirname "$0")/.." || exit 1
OUT=${1:-/tmp/mf_asm}
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form=1 -S --cuda-device-only -x hip"
FILES="k_generic k_depthwise k_pointwise k_fused_mm k_stage k_dwfc k_tail3 k_gemm k_rt k_quad k_quad_mm k_chain"
for f in $FILES; do
    hipcc $FLAGS microflow_rs_amd/csrc/$f.hip -o "$OUT/$f.s" 2>/dev/null &
done
wait
rc=0
for f in $FILES; do
    python scripts/asm_barrier_waits.py "$OUT/$f.s" > "$OUT/$f.scan"
    n=$(grep -c "pending [1-9]" "$OUT/$f.scan")
    k=$(grep -c "barriers" "$OUT/$f.scan")
    echo "$f: $k kernels, $n with a barrier reached with LDS operations pending"
    grep "pending [1-9]" "$OUT/$f.scan" | sed 's/  */ /g' | cut -c1-150 | head -8
    if [ "$n" != 0 ] && [ "$f" != k_gemm ]; then rc=1; fi
done
exit $rc
