#!/bin/bash
# GPU box: effective shader clock and SQ wait breakdown of the fused kernels, normal build vs the
# MF_RR_DIAG=1 (no stores) build: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / kernel duration.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/diag_clock
mkdir -p $OUT
cp microflow_rs_amd/libmicroflow_amd.so /tmp/lib_good.so
for d in ${DIAGS:-0 1}; do
  if [ $d != 0 ]; then MF_EXTRA_HIPCC_FLAGS="-DMF_RR_DIAG=$d" python microflow_rs_amd/build.py --force > /tmp/build_$d.log 2>&1; fi
  rm -rf $OUT/pmc_$d
  (cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_$d -- \
      python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra > /dev/null 2> $OUT/pmc_$d.err)
  python - $OUT/pmc_$d $d <<'PY'
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"mf::k::(dwpw_[a-z]+<[^>]*>|dw3x3_stem8<[^>]*>|stage_[a-z0-9_]*<[^>]*>)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1).replace(" ", "")[:28]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(acc):
    a = {n: sum(v) / len(v) for n, v in acc[k].items()}
    d = sum(dur[k]) / len(dur[k])
    wc = a["SQ_WAVE_CYCLES"]
    print("diag %s %-28s %7.1f us  clock %.2f GHz  wait_any %.2f wait_inst %.2f active %.2f  valu/simd/cyc %.3f" % (
        sys.argv[2], k, d / 1e3, a["GRBM_GUI_ACTIVE"] / 8 / d, a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc,
        a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_INSTS_VALU"] / 1024 / (a["GRBM_GUI_ACTIVE"] / 8)))
PY
done
cp /tmp/lib_good.so microflow_rs_amd/libmicroflow_amd.so
