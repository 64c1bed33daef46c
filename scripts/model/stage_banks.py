#!/usr/bin/env python3
"""LDS bank-conflict evaluation of the late-stage kernel's (k_stage.hip) four dominant access patterns --
depthwise tap reads, depthwise MID writes, pointwise MID reads, pointwise tile writes of the 6x6x128 pairs --
over the free layout parameters: tile row padding, MID plane padding, x / y swizzles of the 16-byte group index.
Prints LDS-array cycles per wave instruction (ideal 4 for b128 reads, 2 for b32 writes)."""
import itertools
from lds_banks import cycles

G, PIX6, LP6 = 4, 36, 128


def evaluate(rowpad, planepad, sx, sy, verbose=False):
    """sx: swizzle from x (function), sy: swizzle from tile row y"""
    ROW6 = 128 + 768 + 128 + rowpad
    TILE6 = 8 * ROW6
    PLANE6 = G * PIX6 * 16 + planepad
    res = {}
    # P1 tap reads: columns = (cy = col&1, cg = (col>>1)&3, cx = col>>3); wave q
    tot = cnt = 0
    for q in range(8):
        for uy in range(3):
            for ux in range(3):
                for ty in range(3):
                    ad = []
                    for lane in range(64):
                        col, g = lane & 15, lane >> 4
                        cy, cg, cx = col & 1, (col >> 1) & 3, col >> 3
                        xin = cx + 2 * ux + g - 1
                        yt = cy + 2 * uy + ty
                        ad.append(cg * TILE6 + yt * ROW6 + LP6 + xin * 128 + 16 * (q ^ sx(xin) ^ sy(yt)))
                    tot += cycles(ad, "read_b128")
                    cnt += 1
    res["tap_read"] = tot / cnt
    # P2 dw MID writes
    tot = cnt = 0
    for q in range(8):
        for uy in range(3):
            for ux in range(3):
                ad = []
                for lane in range(64):
                    col, g = lane & 15, lane >> 4
                    cy, cg, cx = col & 1, (col >> 1) & 3, col >> 3
                    pix = cg * PIX6 + (cy + 2 * uy) * 6 + cx + 2 * ux
                    ad.append(q * PLANE6 + pix * 16 + 4 * g)
                tot += cycles(ad, "write_b32")
                cnt += 1
    res["mid_write"] = tot / cnt
    # P3 pw MID reads
    tot = cnt = 0
    for c in range(9):
        for ks in range(2):
            ad = []
            for lane in range(64):
                pcol, pg = lane & 15, lane >> 4
                ad.append((pg + 4 * ks) * PLANE6 + (c * 16 + pcol) * 16)
            tot += cycles(ad, "read_b128")
            cnt += 1
    res["mid_read"] = tot / cnt
    # P4 pw tile writes
    tot = cnt = 0
    for w in range(8):
        for c in range(9):
            ad = []
            for lane in range(64):
                pcol, pg = lane & 15, lane >> 4
                pix = c * 16 + pcol
                img, r = divmod(pix, PIX6)
                y, x = divmod(r, 6)
                ad.append(img * TILE6 + (y + 1) * ROW6 + LP6 + x * 128 + 16 * (w ^ sx(x) ^ sy(y + 1)) + 4 * pg)
            tot += cycles(ad, "write_b32")
            cnt += 1
    res["tile_write"] = tot / cnt
    # weighted LDS cycles per wave per pair: 27 tap reads, 9 MID writes, 18 MID reads, 9 tile writes
    res["per_pair"] = 27 * res["tap_read"] + 9 * res["mid_write"] + 18 * res["mid_read"] + 9 * res["tile_write"]
    return res


def swz(bits_out):
    """bits_out: tuple of input-bit indices (or None) per output bit 0..2"""
    def f(v):
        r = 0
        for i, b in enumerate(bits_out):
            if b is not None:
                r |= ((v >> b) & 1) << i
        return r
    return f


if __name__ == "__main__":
    cur = evaluate(16, 16, swz((0, None, 0)), swz((None, None, None)))
    print("current (rowpad 16, planepad 16, TS 0x101):", {k: round(v, 2) for k, v in cur.items()})
    best = []
    opts = [None, 0]  # x has only bit 0 lane-constant (units step by 2 in x); tile row y: bit 0 (units step by 2 rows)
    for rowpad in range(0, 256, 16):
        for planepad in (0, 16, 32, 64, 128):
            for bx in itertools.product(opts, repeat=3):
                for by in itertools.product(opts, repeat=3):
                    r = evaluate(rowpad, planepad, swz(bx), swz(by))
                    best.append((r["per_pair"], rowpad, planepad, bx, by, r))
    best.sort(key=lambda t: t[0])
    for b in best[:8]:
        print(round(b[0], 1), "rowpad", b[1], "planepad", b[2], "x-swz", b[3], "y-swz", b[4], {k: round(v, 2) for k, v in b[5].items()})
    print("ideal per pair:", 27 * 4 + 9 * 2 + 18 * 4 + 9 * 2)
