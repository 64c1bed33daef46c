#!/bin/bash
# A/B the fused depthwise+pointwise tuning candidates (MF_DWPW_ALT_SHAPES in kernels.hpp):
# one bench.py run per candidate, printing the per-kernel times of the fused groups.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-15}
for i in -1 $(seq 0 $((N-1))); do
  if [ $i -ge 0 ]; then export MF_DWPW_ALT=$i; else unset MF_DWPW_ALT; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
ks = {k['kernel']: k['ms'] for k in r['kernels']}
print('alt', '$i', 'ms/step', r['ms_per_step'], 'ok' if r['parity']['bit_exact_vs_oracle'] else 'MISMATCH',
      ' '.join('%s=%.4f' % (k.replace('dwpw3x3', ''), v) for k, v in ks.items() if 'dwpw' in k and True))
"
done
