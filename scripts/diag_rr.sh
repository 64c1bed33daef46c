#!/bin/bash
# GPU box: build the three diagnostics variants of dwpw_rr (MF_RR_DIAG) and time each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp microflow_rs_amd/libmicroflow_amd.so /tmp/lib_good.so
for d in 1 2 3; do
  MF_EXTRA_HIPCC_FLAGS="-DMF_RR_DIAG=$d" python microflow_rs_amd/build.py --force > /tmp/build_$d.log 2>&1 || { echo build $d failed; tail -5 /tmp/build_$d.log; continue; }
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('diag $d', ' '.join('%s=%.4f' % (k['kernel'][:22], k['ms']) for k in r['kernels'] if 'dwpw_rr' in k['kernel']))
"
done
cp /tmp/lib_good.so microflow_rs_amd/libmicroflow_amd.so
