#!/bin/bash
# A/B the fused-pair tuning candidates on the GPU box: one bench.py run per candidate of
#   MF_DWRR_ALT_SHAPES (VAR=MF_DWRR_ALT) or MF_DWMM_ALT_SHAPES (VAR=MF_DWMM_ALT) in kernels.hpp.
# usage: tune_fused.sh <VAR> <count> [extra env assignments...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=$1; N=$2; shift 2
for e in "$@"; do export "$e"; done
for i in -1 $(seq 0 $((N-1))); do
  if [ $i -ge 0 ]; then export $VAR=$i; else unset $VAR; fi
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read().strip().splitlines()[-1])
    ks = [(k['kernel'], k['ms']) for k in r['kernels'] if 'dwpw' in k['kernel']]
    print('alt', '$i', 'ms/step', r['ms_per_step'], 'ok' if r['parity']['bit_exact_vs_oracle'] else 'MISMATCH',
          ' '.join('%s=%.4f' % (k.split('<')[0][5:] + '<' + ','.join(k.split('<')[1].split(',')[:3]) + '..' + ','.join(k.rstrip('>').split(',')[5:]), v) for k, v in ks[:5]))
except Exception as e:
    print('alt', '$i', 'FAILED', e)
"
done
