#!/bin/bash
# SQ counters of the chain kernels on the generated 64x64 model (two passes, 8 counters each) -> gpurun_out/chain_sq.csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT
cat > /tmp/one_model.py <<'PY'
import os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, microflow_rs_amd as mf, tflite_writer as tw
from microflow_rs_amd.model import synth_i8
side = int(os.environ.get("SIDE", "64"))
m = mf.model(tw.person_detect_like(np.random.default_rng(side), side, 1.0))
B = int(65536 * 96 * 96 / (side * side)); m.prepare(B)
x = synth_i8(9, 0, B * m.input_elems); y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
for _ in range(3): m.time_device(x, y, B, warmup=0, iters=1, per_op=False)
PY
rm -rf $OUT/csq1 $OUT/csq2
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/csq1 -- python /tmp/one_model.py > /dev/null 2> $OUT/csq1.err)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/csq2 -- python /tmp/one_model.py > /dev/null 2> $OUT/csq2.err)
python - <<'PY'
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in ("gpurun_out/csq1", "gpurun_out/csq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "chain_rt" in r["Kernel_Name"]:
                key = "%s grid%s lds%s" % (re.search(r"chain_rt<[^>]*>", r["Kernel_Name"]).group(0), r.get("Grid_Size", "?"), r.get("LDS_Block_Size", "?"))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_INSTS_SALU","SQ_INSTS_VMEM","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_SMEM","GRBM_GUI_ACTIVE"]
with open("gpurun_out/chain_sq.csv", "w") as o:
    o.write("kernel," + ",".join(names) + "\n")
    for k in sorted(acc):
        o.write('"' + k + '",' + ",".join("%.0f" % (sum(acc[k][n]) / max(len(acc[k][n]), 1)) for n in names) + "\n")
print(open("gpurun_out/chain_sq.csv").read())
PY
