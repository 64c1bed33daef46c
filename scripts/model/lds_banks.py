#!/usr/bin/env python3
"""LDS bank-conflict model for gfx950 (MI355X_MICROARCH.md, LDS table) used to choose the tile / MID
swizzles of the matrix-pipe depthwise phase (k_fused_mm.hip).  Pure arithmetic, no GPU.

cost(addresses of one wave instruction) = LDS-array cycles = sum over lane groups of the largest
number of DISTINCT addresses that fall on one bank (identical addresses broadcast)."""
from collections import defaultdict

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               [32 + x for x in list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))],
               [32 + x for x in list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]]
HALF_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def cycles(addrs, kind):
    """addrs: 64 byte addresses (None = inactive lane)"""
    if kind == "read_b128":
        groups, width, banks = B128_GROUPS, 4, 64
    elif kind == "read_b64":
        groups, width, banks = HALF_GROUPS, 2, 64
    elif kind in ("read_b32", "write_b32"):
        groups, width, banks = HALF_GROUPS, 1, 32
    else:
        raise ValueError(kind)
    total = 0
    for g in groups:
        per_bank = defaultdict(set)
        for l in g:
            a = addrs[l]
            if a is None:
                continue
            for d in range(width):
                per_bank[(a // 4 + d) % banks].add(a // 4 + d)
        total += max((len(s) for s in per_bank.values()), default=0)
    return total


def ideal(kind):
    return {"read_b128": 4, "read_b64": 2, "read_b32": 2, "write_b32": 2}[kind]
