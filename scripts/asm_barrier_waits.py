#!/usr/bin/env python3
"""Finds `s_barrier` instructions a wave can reach with LDS operations of its own still outstanding, in a hipcc -S listing.

`__syncthreads()` is fence(release, workgroup) + s_barrier + fence(acquire): the `s_waitcnt lgkmcnt(0)` in front of the barrier
comes from the release fence, as a SOFT wait that SIInsertWaitcnts deletes when its scoreboard shows nothing pending.  At a loop
header the pass visits the block first with the pre-header's state only; a soft wait deleted on that visit is not re-created when
the back edge later brings pending ds_writes (gfx950 backs off barriers, so the pass itself never forces a wait in front of one).
Seen in round 6: quad_mm's barrier at the top of a step had no lgkmcnt wait although the previous iteration ends in ds_write -- the
kernel was wrong once in ~20 launches.  A kernel whose loop ends in LDS stores waits for them itself (`lds_settle()`), and this
scan is the gate: a forward data-flow over the listing's basic blocks, "an LDS instruction was issued and no `s_waitcnt` with
lgkmcnt(0) has followed", reported at every s_barrier where it holds.

    python scripts/asm_barrier_waits.py /tmp/k.s [substring-filter ...]      exit status 1 if anything is reported"""
import re
import subprocess
import sys

LDS_OP = re.compile(r"^(ds_(?!bpermute|permute|swizzle|nop)|buffer_load\S* .* lds)")  # (the cross-lane ds_* never touch LDS memory)


def kernels(lines):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[A-Za-z0-9_]+:", l)]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
    for (i, _), name in zip(starts, names):
        end = next((j for j in range(i + 1, len(lines)) if lines[j].startswith(".Lfunc_end")), None)
        if end is not None:
            yield re.sub(r"void mf::k::|\(signed char.*|\(mf::k::.*|\(float.*|\(void.*", "", name), lines[i + 1:end]


def scan(body):
    """-> list of (line offset, text) of s_barrier reached with LDS operations pending"""
    # basic blocks: a label starts one, a branch ends one
    blocks, cur, label_of = [], {"label": None, "ins": []}, {}
    for off, raw in enumerate(body):
        s = raw.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", raw)
        if m:
            if cur["ins"] or cur["label"]:
                blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        if not raw.startswith("\t") or not s or s.startswith((";", ".")):
            continue
        cur["ins"].append((off, s))
        if s.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"] or cur["label"]:
        blocks.append(cur)
    for k, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = k
    succ = []
    for k, b in enumerate(blocks):
        last = b["ins"][-1][1] if b["ins"] else ""
        out = []
        if last.startswith(("s_cbranch", "s_branch")):
            t = last.split()[-1]
            if t in label_of:
                out.append(label_of[t])
        if not last.startswith(("s_branch", "s_endpgm")) and k + 1 < len(blocks):
            out.append(k + 1)
        succ.append(out)
    # state = an upper bound of the wave's outstanding lgkm operations whose oldest part may be an LDS operation (LDS operations
    # return in order, so `lgkmcnt(n)` leaves at most n of them; scalar loads share the counter: they only make a counted wait
    # stricter).  0 = nothing of LDS outstanding.
    CAP = 32
    pend_in = [0] * len(blocks)
    hits = {}
    work = list(range(len(blocks)))
    while work:
        k = work.pop(0)
        p = pend_in[k]
        for off, s in blocks[k]["ins"]:
            op = s.split()[0]
            m = re.search(r"lgkmcnt\((\d+)\)", s) if op == "s_waitcnt" else None
            if m:
                p = min(p, int(m.group(1)))
            elif op == "s_waitcnt" and re.fullmatch(r"s_waitcnt\s+0(x0)?", s):
                p = 0
            elif LDS_OP.match(s):  # (global_load_lds -- an LDS-DMA -- is a vmcnt matter: asm_dma_waits.py)
                p = min(CAP, p + 1)
            elif op in ("s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16") and p:
                p = min(CAP, p + 1)
            elif op == "s_barrier" and p:
                hits[off] = s
        for t in succ[k]:
            if p > pend_in[t]:
                pend_in[t] = p
                work.append(t)
    return sorted(hits.items())


def main():
    path, filt = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    bad = 0
    for name, body in kernels(lines):
        if filt and not all(f in name for f in filt):
            continue
        nb = sum(1 for l in body if l.strip().startswith("s_barrier"))
        hits = scan(body)
        bad += len(hits)
        print("%-100s barriers %2d, reached with LDS operations pending %d" % (name[:100], nb, len(hits)))
        for off, _ in hits:
            ctx = [b.strip() for b in body[max(0, off - 6):off + 1] if b.startswith("\t")]
            print("      line +%d: ... %s" % (off, " | ".join(ctx[-5:])))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
