export TMPDIR=/tmp
for cfg in "" 0x100 0x101 0x102 0x104 0x801 0x804; do
  echo "== MF_DQ_CFG=$cfg"
  if [ -z "$cfg" ]; then python scripts/time_chain_brief.py 2>/dev/null; else MF_DQ_CFG=$cfg python scripts/time_chain_brief.py 2>/dev/null; fi
done
