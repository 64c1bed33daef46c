export TMPDIR=/tmp
for oc in 1.0 0.0; do
  echo "== opcost $oc"
  MF_CHAIN_OPCOST=$oc MF_CHAIN_VERBOSE=1 python scripts/time_generated.py 128:1.0 64:1.0 96:0.5 2>&1 | grep -v "layer-wise\|^ .*\(pw_\|dw3x3_rt\|avg\|conv1x1\|softmax\|tail\|stem\)" | grep "est\|chain_rt\|==" 
done
