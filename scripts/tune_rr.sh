#!/bin/bash
# GPU box: time the MF_DWRR_ALT candidates (kernels.hpp: MF_DWRR_ALT_SHAPES), two runs each; "-" = the shipped table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for alt in - "$@"; do
  for i in 1 2; do
    if [ "$alt" = "-" ]; then unset MF_DWRR_ALT; else export MF_DWRR_ALT=$alt; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed --no-extra 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('alt $alt', r['ms_per_step'], r['parity']['bit_exact_vs_oracle'], ' '.join('%s=%.4f' % (k['kernel'][:22], k['ms']) for k in r['kernels'] if 'dwpw_rr' in k['kernel']))
"
  done
done
