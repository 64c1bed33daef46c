#!/bin/bash
# GPU box: kernel durations (rocprofv3 --kernel-trace --stats) of the GEMM ubench variants and of the product GEMM
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/fc2_a $R/gpurun_out/fc2_b
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fc2_a -- $R/scripts/ubench/gemm_i8 > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fc2_b -- python $R/scripts/time_fc.py > /dev/null 2>&1
for d in fc2_a fc2_b; do
  f=$(find $R/gpurun_out/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d"; grep -i "gemm\|fc_mfma" "$f" | cut -d, -f1-7 | sed 's/signed char const\*.*)"/..."/' | cut -c1-160
done
