#!/usr/bin/env python3
"""Register / occupancy / instruction-mix summary of the kernels in a hipcc -S device listing.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 \
          --cuda-device-only -S -o /tmp/k.s microflow_rs_amd/csrc/k_fused_mm.hip
    python scripts/asm_stats.py /tmp/k.s [substring-filter] [--mix]"""
import collections
import re
import subprocess
import sys


def main():
    path = sys.argv[1]
    filt = [a for a in sys.argv[2:] if not a.startswith("--")]
    mix = "--mix" in sys.argv
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[A-Za-z0-9_]+:", l)]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
    for (i, _), name in zip(starts, names):
        name = re.sub(r"void mf::k::|\(signed char.*", "", name)
        if filt and not all(f in name for f in filt):
            continue
        end = next((j for j in range(i + 1, len(lines)) if lines[j].startswith("; Occupancy")), None)
        if end is None:
            continue
        body = lines[i:end + 1]
        get = lambda key: next((l.split(":")[1].strip() for l in body if l.startswith("; " + key + ":")), "?")
        ops = collections.Counter(l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";")))
        valu = sum(v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        print("%-70s vgpr %s sgpr %s scratch %s occ %s | insts %d valu %d mfma %d ds %d vmem %d barrier %d" % (
            name[:70], get("NumVgprs"), get("NumSgprs"), get("ScratchSize"), get("Occupancy"), sum(ops.values()), valu,
            sum(v for k, v in ops.items() if k.startswith("v_mfma")), sum(v for k, v in ops.items() if k.startswith("ds_")),
            sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_"))), ops.get("s_barrier", 0)))
        if mix:
            print("     " + ", ".join("%s %d" % kv for kv in ops.most_common(28)))


if __name__ == "__main__":
    main()
