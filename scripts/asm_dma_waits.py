#!/usr/bin/env python3
"""Finds compiler-inserted `s_waitcnt vmcnt(..)` in front of DS instructions in a hipcc -S listing.

hipcc (SIInsertWaitcnts) orders every LDS *store* behind all outstanding LDS-DMA loads (`global_load_lds_*`) of the wave,
because it cannot prove that the two LDS ranges differ: a kernel that issues the next step's staging DMA and then writes
an intermediate tile with `ds_write` waits for the HBM round trip right there.  This lists, per kernel, the vmcnt waits the
compiler added (those outside ;;#ASMSTART .. ;;#ASMEND) and what follows them.
    python scripts/asm_dma_waits.py /tmp/k.s [substring-filter ...]"""
import re
import subprocess
import sys


def main():
    path = sys.argv[1]
    filt = sys.argv[2:]
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[A-Za-z0-9_]+:", l)]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
    for (i, _), name in zip(starts, names):
        name = re.sub(r"void mf::k::|\(signed char.*|\(mf::k::.*", "", name)
        if filt and not all(f in name for f in filt):
            continue
        end = next((j for j in range(i + 1, len(lines)) if lines[j].startswith("; Occupancy")), None)
        if end is None:
            continue
        body = lines[i:end]
        in_asm = False
        dma = sum(1 for l in body if "global_load_lds" in l or ("buffer_load" in l and " lds" in l))
        found = []
        for j, l in enumerate(body):
            s = l.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
            elif s.startswith(";;#ASMEND"):
                in_asm = False
            elif not in_asm and s.startswith("s_waitcnt") and "vmcnt" in s:
                nxt = next((b.strip().split()[0] for b in body[j + 1:j + 6] if b.startswith("\t") and not b.strip().startswith((";", "s_waitcnt", "s_nop"))), "?")
                found.append((j, s, nxt))
        ds = [f for f in found if f[2].startswith("ds_")]
        print("%-90s dma %2d  compiler vmcnt waits %2d, in front of DS ops %2d %s" % (
            name[:90], dma, len(found), len(ds), sorted({f[2] for f in ds})))


if __name__ == "__main__":
    main()
