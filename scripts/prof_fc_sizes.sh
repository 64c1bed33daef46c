#!/bin/bash
# What bounds fc_mfma: matrix-pipe busy cycles, the effective shader clock and the wave count of the GEMM at 4096^3
# and 8192^3 from ONE rocprofv3 --pmc pass (kernel-trace only), next to the un-profiled event timing of the same sizes.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fc_sizes; rm -rf $O; mkdir -p $O
python $R/scripts/fc_sizes.py 4096 8192 2>/dev/null | tail -1 > $O/events.json
cat $O/events.json
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d $O/pmc -- python $R/scripts/fc_sizes.py 4096 8192 > $O/pmc.log 2>&1
echo rc $?
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fc_mfma" in r["Kernel_Name"]:
            size = int(r["Grid_Size"]) if "Grid_Size" in r else 0
            acc[size][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[size]["_dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
ev = json.load(open("$O/events.json"))
out = {"events_unprofiled": ev, "by_grid_size": {}}
for size in sorted(acc):
    m = {k: sum(v) / len(v) for k, v in acc[size].items()}
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
    e = {"launches": len(acc[size]["GRBM_GUI_ACTIVE"]), "avg_duration_ns_under_pmc": m["_dur"], "kernel_cycles_per_xcd": cycles,
         "effective_clock_ghz": cycles / m["_dur"], "SQ_WAVES": m.get("SQ_WAVES"),
         "mfma_busy_frac_of_kernel_cycles": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024),
         "wait_any_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
         "valu_inst_per_clk_per_simd": m.get("SQ_INSTS_VALU", 0) / cycles / 1024}
    e["mfma_busy_frac_at_nominal_2p4ghz"] = e["mfma_busy_frac_of_kernel_cycles"] * e["effective_clock_ghz"] / 2.4
    out["by_grid_size"][str(size)] = e
out["note"] = ("v_mfma_i32_32x32x32_i8 holds its SIMD's matrix pipe for 32 cycles (16 passes); mfma_busy_frac_of_kernel_cycles = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of the kernel's own cycles its matrix "
               "pipes were busy; the effective clock (cycles per XCD / duration) shows the power management under random int8 operands")
print(json.dumps(out, indent=1))
open("$R/gpurun_out/fc_counters.json", "w").write(json.dumps(out, indent=1))
PY
