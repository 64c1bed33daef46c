#!/bin/bash
# MFMA utilisation of the FullyConnected GEMM from PMC counters (one --pmc pass, kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/fc_pmc
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $R/gpurun_out/fc_pmc -- python $R/bench.py --workload fc4096 --steps 20 > $R/gpurun_out/fc_pmc.log 2>&1
echo rc $?
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/fc_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
dur = []
for r in csv.DictReader(open(f)):
    if "fc_mfma" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
n_simd = 256 * 4
out = dict(m)
out["launches"] = len(dur)
out["avg_duration_ns_under_pmc"] = sum(dur) / max(len(dur), 1)
out["mfma_instructions"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0
# GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (968 k for a 65 us kernel = 8 x 121 k cycles)
cycles = m["GRBM_GUI_ACTIVE"] / 8.0
out["kernel_cycles_per_xcd"] = cycles
out["effective_clock_ghz"] = cycles / out["avg_duration_ns_under_pmc"]
out["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * n_simd)
out["mfma_util_at_nominal_2p4ghz"] = out["mfma_util"] * out["effective_clock_ghz"] / 2.4
out["note"] = ("v_mfma_i32_32x32x32_i8 holds its SIMD's matrix pipe 32 cycles; mfma_util = busy cycles / "
               "(kernel cycles x 1024 SIMDs) = the fraction of the kernel's own cycles the matrix pipes were busy; "
               "the effective clock shows the power management under random int8 operands")
import json
print(json.dumps(out, indent=1))
open("$R/gpurun_out/fc_mfma_counters.json", "w").write(json.dumps(out, indent=1))
PY
