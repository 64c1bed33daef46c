#!/bin/bash
# A/B the layer-wise depthwise tuning candidates (MF_DW_ALT_SHAPES in kernels.hpp): one bench.py
# run per candidate, printing the layer-wise depthwise kernel times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-25}
for i in -1 $(seq 0 $((N-1))); do
  if [ $i -ge 0 ]; then export MF_DW_ALT=$i; else unset MF_DW_ALT; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
ks = {}
for k in r['layerwise']['kernels']:
    if k['kind'] == 'depthwise_conv_2d': ks[k['kernel']] = k['ms']
print('alt', '$i', 'dw_ms', r['depthwise']['ms'], 'ok' if r['parity']['bit_exact_vs_oracle'] else 'MISMATCH',
      ' '.join('%s=%.4f' % (k.replace('dw3x3_nhwc', ''), v) for k, v in ks.items()))
"
done
