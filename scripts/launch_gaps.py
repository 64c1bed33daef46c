#!/usr/bin/env python3
"""Gaps between the launches of the fused person_detect step, from a rocprofv3 --kernel-trace CSV:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/gaps -- python $REPO/bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline --no-host-fed
    python scripts/launch_gaps.py gpurun_out/gaps
A step is recognised as the five (or however many) fused launches in order, the first being the launch that starts with penta_rr's kernel;
only full-batch steps (the longest ones) are kept.  Prints per boundary the median gap end(k) -> start(k + 1), and the step's
sum of kernel times against its first-start-to-last-end span."""
import csv
import glob
import statistics
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.split("(")[0].replace("void mf::k::", "")[:48]  # noqa: E731
# a step: Quad13-with-stem launch followed by the launches up to (not including) the next one of the same name
starts = [i for i, r in enumerate(rows) if "Quad13, true" in r[2]]
steps = []
for a, b in zip(starts, starts[1:]):
    seq = [r for r in rows[a:b] if "mf::k::" in r[2]]
    if 3 <= len(seq) <= 8:
        steps.append(seq)
if not steps:
    sys.exit("no steps found")
n = statistics.mode(len(s) for s in steps)
steps = [s for s in steps if len(s) == n]
long = sorted(steps, key=lambda s: s[-1][1] - s[0][0])[len(steps) // 2:]  # the full-batch half
print("%d steps of %d launches (of %d candidates)" % (len(long), n, len(steps)))
for k in range(n):
    dur = statistics.median(s[k][1] - s[k][0] for s in long) / 1e3
    gap = statistics.median(s[k + 1][0] - s[k][1] for s in long) / 1e3 if k + 1 < n else float("nan")
    print("  %-50s %8.1f us   gap to the next launch %6.1f us" % (short(long[0][k][2]), dur, gap))
span = statistics.median(s[-1][1] - s[0][0] for s in long) / 1e3
ksum = statistics.median(sum(e - b for b, e, _ in s) for s in long) / 1e3
print("  sum of launches %.1f us, first start -> last end %.1f us, gaps %.1f us (%.1f %%)" % (ksum, span, span - ksum, 100 * (span - ksum) / span))
