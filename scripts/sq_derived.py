#!/usr/bin/env python3
"""gpurun_out/sq_summary.csv (scripts/gpu_check.sh, STEPS=sqpmc) -> per-kernel derived ratios.
    python scripts/sq_derived.py gpurun_out/sq_summary.csv > profiles/rNN/x_sq_derived.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
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
