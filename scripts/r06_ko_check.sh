cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MF_ALLOW_DIAG_BUILD=1
cp microflow_rs_amd/libmicroflow_amd.so /tmp/good.so
for v in ko default; do
  if [ $v = ko ]; then cp microflow_rs_amd/variants/lib_ko.so microflow_rs_amd/libmicroflow_amd.so; else cp /tmp/good.so microflow_rs_amd/libmicroflow_amd.so; fi
  touch microflow_rs_amd/libmicroflow_amd.so
  for i in 1 2 3 4 5 6; do echo "$v race-only $i: $(python -m pytest tests/test_gpu_race.py -x -q 2>&1 | tail -1)"; done
  for i in 1 2 3; do echo "$v full $i: $(python -m pytest tests -m gpu -x -q 2>&1 | tail -1)"; done
done
cp /tmp/good.so microflow_rs_amd/libmicroflow_amd.so
