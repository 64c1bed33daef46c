#!/bin/bash
# GPU box: rebuild k_stage.hip with each MF_STAGE_* setting given as arguments ("-DMF_STAGE_SB=2" ...) and time it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp microflow_rs_amd/libmicroflow_amd.so /tmp/lib_good.so
for flags in "$@"; do
  MF_EXTRA_HIPCC_FLAGS="$flags" python microflow_rs_amd/build.py --force > /tmp/build.log 2>&1 || { echo "build failed: $flags"; tail -3 /tmp/build.log; continue; }
  for i in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed --no-extra 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$flags', r['ms_per_step'], r['parity']['bit_exact_vs_oracle'], ' '.join('%s=%.4f' % (k['kernel'][:10], k['ms']) for k in r['kernels'] if 'stage' in k['kernel'] or 'mm<12,12,64,1' in k['kernel']))
"
  done
done
cp /tmp/lib_good.so microflow_rs_amd/libmicroflow_amd.so
