#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic.

    python scripts/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE [bench.json] > profiles/rNN/pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB
as derived from the L2's memory-side request counters; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled here.
WRITE_SIZE is reported as is (uncalibrated).  Values are per launch (median over launches).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def source_sha16():
    """the tag bench.py checks a committed profile against (bench.py source_sha16): sha256 over the kernel sources"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "microflow_rs_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load(d, counter):
    # one file per profiled process (bench.py runs its vendor-GEMM cross-check in a child): take them all
    acc = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    m = re.search(r"mf::k::([A-Za-z0-9_]+)(<[^>]*>)?", name.replace("mf::k::Quad", "Quad"))
    if not m:
        return name[:40]
    # rocprofv3 prints bool template arguments as true/false, the library's names use 1/0; the
    # trailing variant arguments (MG = bit-pattern int->f32, XR4 = element type, F32IN) are not part
    # of the library's names
    base, args = m.group(1), (m.group(2) or "").replace(" ", "").replace("true", "1").replace("false", "0")
    keep = {"dw3x3_nhwc": 6, "pw_mfma": 2, "dwpw3x3": 8, "dwpw_rr": 8, "dwpw_mm": 8, "stage_6x6x128": 3,
            "tail_pool_head_softmax": 1, "fc_rowwave": 1, "fc_rowwave_softmax": 1}
    if base == "dwpw_mm" and args.endswith(",1>") and args.count(",") >= 16:  # DWONLY instance = the layer-wise depthwise op
        a = args.strip("<>").split(",")
        return "dw3x3_mm<%s>" % ",".join(a[:4] + a[5:7])
    if base in keep:
        args = "<" + ",".join(args.strip("<>").split(",")[: keep[base]]) + ">"
    if base == "dw_c1_lds":
        args = ""
    if base == "quad_mm_12x12x64":  # (k_quad_mm.hip: template arguments = epilogue mode, element type)
        return "quad_mm<12,12,64,1,64|12,12,64,2,128>"
    if base == "quad_rr":  # quad_rr<mf::k::Quad13, MG, XR4>: the library's name carries the two pair shapes
        which = "Quad13" if "Quad13" in name else "Quad57"
        if which == "Quad13" and re.search(r"Quad13,\s*(true|1)\b", name):  # the STEM instance: ops 0..4 in one launch
            return "penta_rr<96,96,1,2,8|48,48,8,1,16|48,48,16,2,32>"
        return {"Quad13": "quad_rr<48,48,8,1,16|48,48,16,2,32>", "Quad57": "quad_rr<24,24,32,1,32|24,24,32,2,64>"}[which]
    if base == "pair3_tail":
        if re.search(r",\s*(true|1)>$", args):  # the FRONT instance: ops 23..30 in one launch
            return "pair_front_tail<6,6,128,2,256|3,3,256,2>"
        args = "<3,3,256,2>"
    if base == "dwc1_fc_softmax":
        args = "<49,40,10,8,2>"
    if base == "dw3x3_stem8_mm":  # matrix-pipe taps: the library calls both stems dw3x3_stem8<96,96,2>
        base, args = "dw3x3_stem8", "<96,96,2>" + (" (f32 input)" if args.endswith(",1>") else "")
    elif base == "dw3x3_stem8":   # VALU-tap stem: what remains in use is its f32-input variant (M::predict)
        args = "<96,96,2> (VALU taps%s)" % (", f32 input" if args.endswith(",1>") else "")
    return base + args


def sq_main(dirs, stats_csv):
    """python scripts/pmc_summary.py --sq gpurun_out/pmc_sq gpurun_out/pmc_sq2 [kernel_stats.csv] > profiles/sq_latest.json
    Per kernel (library names), median over launches: VALU wave-instructions per SIMD clock (/ 0.5 = valu_busy: a wave64
    instruction occupies a SIMD-32 for at least two clocks), LDS bank-conflict ratio, share of wave cycles parked, effective shader
    clock (cycles per XCD / average duration from the kernel-trace stats of the same command)."""
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                if "mf::k::" in r["Kernel_Name"]:
                    name = r["Counter_Name"]
                    if name == "GRBM_GUI_ACTIVE" and d.rstrip("/").endswith("sq3"):
                        name = "GRBM_GUI_ACTIVE_sq3"
                    acc[short(r["Kernel_Name"])][name].append(float(r["Counter_Value"]))
    dur = {}
    if stats_csv:
        for r in csv.DictReader(open(stats_csv)):
            if "mf::k::" in r["Name"]:
                dur[short(r["Name"])] = float(r["AverageNs"])
    med = lambda v: sorted(v)[len(v) // 2] if v else 0.0  # noqa: E731
    out = {"note": "rocprofv3 --pmc SQ passes of bench.py (scripts/gpu_check.sh STEPS=sqpmc), median over launches; "
                   "valu_inst_per_clk_per_simd = SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
                   "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): the share of the kernel's own cycles its matrix pipes were busy",
           "source_sha16": source_sha16(), "kernels": {}}
    for k in sorted(acc):
        c = {n: med(v) for n, v in acc[k].items()}
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc <= 0:
            continue
        v = c.get("SQ_INSTS_VALU", 0.0) / cyc / 1024.0
        e = {"valu_inst_per_clk_per_simd": round(v, 4),
             "lds_bank_conflict_ratio": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4) if c.get("SQ_LDS_IDX_ACTIVE") else 0.0,
             "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
             "cycles_per_xcd": round(cyc)}
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):  # third pass (its own GRBM_GUI_ACTIVE: the passes are separate runs)
            cyc3 = med(acc[k].get("GRBM_GUI_ACTIVE_sq3", [])) / 8.0 or cyc
            e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc3 / 1024.0, 4)
            if c.get("SQ_INSTS_MFMA"):
                e["mfma_busy_cycles_per_inst"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"], 2)
        if k in dur and dur[k] > 0:  # (profiled cycles over the UNprofiled duration of the same command: indicative only)
            e["cycles_per_xcd_over_trace_duration_GHz"] = round(cyc / dur[k], 3)
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--sq":
        rest = sys.argv[2:]
        return sq_main([a for a in rest if not a.endswith(".csv")], next((a for a in rest if a.endswith(".csv")), None))
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    bench = None
    if len(sys.argv) > 3:
        text = open(sys.argv[3]).read().strip()
        try:
            bench = json.loads(text)                      # bench_details.json (the full record)
        except ValueError:
            bench = json.loads(text.splitlines()[-1])     # a one-line record
    alg = {}
    if bench:
        for k in bench["kernels"] + bench.get("layerwise", {}).get("kernels", []):
            alg.setdefault(k["kernel"], k["bytes"])
    out = {"per_gpu_batch": bench["config"]["per_gpu_batch"] if bench else None,
           "unit": "bytes per launch", "fetch_correction": "FETCH_SIZE x 2 (gfx950 wide-read calibration)",
           "source_sha16": source_sha16(), "kernels": []}
    for name in sorted(set(fetch) | set(write)):
        if "mf::k::" not in name:
            continue
        # median over launches: the bench's parity checks launch the same kernels on a few images as well
        med = lambda v: sorted(v)[len(v) // 2] if v else 0.0  # noqa: E731
        f = med(fetch.get(name, [])) * 1024 * 2
        w = med(write.get(name, [])) * 1024
        s = short(name)
        e = {"kernel": s, "launches": len(fetch.get(name, [])), "fetch_bytes": int(f), "write_bytes": int(w),
             "traffic_bytes": int(f + w)}
        if s in alg:
            e["algorithmic_bytes"] = alg[s]
            e["traffic_over_algorithmic"] = round((f + w) / alg[s], 3)
        out["kernels"].append(e)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
