#!/usr/bin/env python3
"""gpurun_out/sq_summary.csv (scripts/gpu_check.sh, STEPS=sqpmc) -> per-kernel derived ratios.
    python scripts/sq_derived.py gpurun_out/sq_summary.csv > profiles/rNN/x_sq_derived.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
if "--json" in sys.argv:
    # python scripts/sq_derived.py gpurun_out/sq_summary.csv --json <kernel_stats.csv> > profiles/sq_latest.json
    # (kernel_stats.csv of the same run supplies each kernel's average duration -> effective shader clock)
    import json
    import re
    dur = {}
    ks = [a for a in sys.argv[2:] if a.endswith(".csv")]
    if ks:
        for r in csv.DictReader(open(ks[0])):
            m = re.search(r"mf::k::([A-Za-z0-9_]+)(<.*>)?", r["Name"])
            if m:
                dur[m.group(1) + (m.group(2) or "").replace(" ", "")] = float(r["AverageNs"])
    out = {"note": "rocprofv3 --pmc SQ passes of bench.py (scripts/gpu_check.sh STEPS=sqpmc); valu_busy = SQ_INSTS_VALU / (cycles per XCD x 1024 SIMDs) / 0.5",
           "kernels": {}}
    for r in rows:
        f = lambda k: float(r[k] or 0)
        cyc = f("GRBM_GUI_ACTIVE") / 8.0
        if cyc <= 0:
            continue
        v = f("SQ_INSTS_VALU") / cyc / 1024.0
        name = r["kernel"]
        # the library's kernel names (mf_model_get_op) drop spaces and some template arguments: key by what bench.py looks up
        out["kernels"][name] = {"valu_inst_per_clk_per_simd": round(v, 4), "valu_busy": round(v / 0.5, 4),
                                "lds_bank_conflict_ratio": round(f("SQ_LDS_BANK_CONFLICT") / f("SQ_LDS_IDX_ACTIVE"), 4) if f("SQ_LDS_IDX_ACTIVE") > 0 else 0.0,
                                "wait_any_frac": round(f("SQ_WAIT_ANY") / f("SQ_WAVE_CYCLES"), 4) if f("SQ_WAVE_CYCLES") > 0 else None,
                                "cycles_per_xcd": round(cyc),
                                "shader_clock_GHz": round(cyc / dur[name], 3) if name in dur and dur[name] > 0 else None}
    print(json.dumps(out, indent=1))
    sys.exit(0)
print("# GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles per XCD = /8); 1024 SIMDs per chip; counters are per launch (mean over launches)")
print("kernel,lds_bank_conflict_over_lds_active,valu_inst_per_clk_per_simd,salu_inst_per_clk_per_simd,lds_inst_per_clk_per_simd,cycles_per_xcd")
for r in rows:
    f = lambda k: float(r[k] or 0)
    cyc = f("GRBM_GUI_ACTIVE") / 8.0
    if cyc <= 0:
        continue
    per = lambda k: f(k) / cyc / 1024.0
    conf = f("SQ_LDS_BANK_CONFLICT") / f("SQ_LDS_IDX_ACTIVE") if f("SQ_LDS_IDX_ACTIVE") > 0 else 0.0
    print('"%s",%.3f,%.3f,%.3f,%.4f,%.0f' % (r["kernel"], conf, per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), cyc))
