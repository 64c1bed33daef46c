#!/bin/bash
# GPU box: the model-level parity tests under every A/B switch of the library (each switch is read once per process).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MF_DEV=1   # (the library ignores every routing / tuning switch without it)
for sw in "" MF_NO_PENTA=1 MF_NO_F32_GROUP=1 MF_NO_QUAD=1 MF_NO_QUAD_MM=1 MF_NO_PAIR_FRONT=1 MF_QUADS=1 MF_QUADS=2 MF_NO_STAGE=1 MF_NO_PAIRTAIL=1 MF_DWPW_IMPL=mm MF_NO_FMA_EPI=1 "MF_NO_FMA_EPI=1 MF_NO_SAT_PACK=1" MF_NO_SAT_PACK=1 MF_NO_MAGIC=1 MF_NO_TABLE=1 MF_NO_RT=1 MF_NO_DWFC=1 MF_DQ_CFG=0 MF_DQ_CFG=0x100 MF_DQ_STATIC=0 MF_NO_CHAIN=1 MF_CHAIN_ALL=1 MF_CHAIN_FORCE=1 MF_CHAIN_AUTOTUNE=1 MF_CHAIN_OPCOST=0 MF_CHAIN_TUNE_G=0 MF_CHAIN_NO_SP=1 MF_CHAIN_NO_RES=1 MF_NO_STEM_RT=1 MF_FC_ROWSUM_FOLD=1 MF_FC_ROWSUM_PREPASS=1 MF_CONV_MM_256=1; do
  echo -n "[$sw] "
  env $sw timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -1
done
