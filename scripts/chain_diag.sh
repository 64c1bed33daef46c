#!/bin/bash
# shader-clock stamps of one workgroup's 3rd step (waves 0 and 3) of every chain launch of a generated model, with the dynamic step
# queue and with static striding.  Needs the diag variant: python scripts/variants.py build k_chain.hip diag="-DMF_CHAIN_DIAG=1"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
cp microflow_rs_amd/libmicroflow_amd.so /tmp/lib.good; cp microflow_rs_amd/variants/lib_diag.so microflow_rs_amd/libmicroflow_amd.so; touch microflow_rs_amd/libmicroflow_amd.so
for cfg in "" 0x100; do
  echo "== MF_DQ_CFG=$cfg"
  if [ -z "$cfg" ]; then python scripts/chain_diag_run.py ${1:-128} ${2:-1.0} 2>&1 | grep -A2 "chain trace"; else MF_DQ_CFG=$cfg python scripts/chain_diag_run.py ${1:-128} ${2:-1.0} 2>&1 | grep -A2 "chain trace"; fi
done
cp /tmp/lib.good microflow_rs_amd/libmicroflow_amd.so
