#!/usr/bin/env python3
"""Brute-force search of the LDS layouts of the matrix-pipe depthwise phase (dwpw_mm, k_fused_mm.hip):
   * tile-side chunk swizzle (applied on the DMA source address) for the tap reads (ds_read_b128),
   * MID-side chunk swizzle for the depthwise result writes (ds_write_b32) and the pointwise
     operand reads (ds_read_b128),
   * the order of the 16 columns of a unit over (image, row, x).
Prints LDS-array cycles per wave instruction (ideal: 4 for b128 reads, 2 for b32 writes)."""
import itertools
import sys
from lds_banks import cycles

SHAPES = [  # H, W, C, S, N, G
    (48, 48, 8, 1, 16, 1), (48, 48, 16, 2, 32, 1), (24, 24, 32, 1, 32, 1), (24, 24, 32, 2, 64, 2),
    (12, 12, 64, 1, 64, 4), (12, 12, 64, 2, 128, 4), (6, 6, 128, 1, 128, 8), (6, 6, 128, 2, 256, 8),
    (3, 3, 256, 1, 256, 8)]


def col_grids(G, OH, OWc):
    """factorisations CG*CY*CX = 16 with CG | G ... (CY, CX may exceed/pad the image: allow padding on x only)"""
    out = []
    for CG in (1, 2, 4, 8, 16):
        if G % CG:
            continue
        for CY in (1, 2, 4, 8, 16):
            if CG * CY > 16 or OH % CY:
                continue
            CX = 16 // (CG * CY)
            UX = -(-OWc // CX)
            eff = OWc / (UX * CX)
            out.append((CG, CY, CX, eff))
    return out


def swz_funcs(nbits, maxbit=6):
    """candidate swizzles built from bits < maxbit of x: output bit i = XOR of a subset of input bits"""
    fs = [("0", lambda x: 0)]
    if nbits == 0 or maxbit == 0:
        return fs
    inbits = list(range(maxbit))
    # each output bit = one input bit or XOR of two
    opts = [(a,) for a in inbits] + [(a, b) for a in inbits for b in inbits if a < b] + [()]
    import itertools as it
    combos = list(it.product(opts, repeat=nbits))
    if len(combos) > 600:
        combos = [c for c in combos if all(len(o) <= 1 for o in c)] + combos[::max(1, len(combos) // 300)]
    for c in combos:
        if all(len(o) == 0 for o in c):
            continue
        def f(x, c=c):
            r = 0
            for i, o in enumerate(c):
                bit = 0
                for b in o:
                    bit ^= (x >> b) & 1
                r |= bit << i
            return r
        fs.append(("/".join("^".join("b%d" % b for b in o) or "0" for o in c), f))
    return fs


def analyse(shape, verbose=False):
    H, W, C, S, N, G = shape
    OH, OW = (H + S - 1) // S, (W + S - 1) // S
    pair = C == 8
    OWc = OW // 2 if pair else OW
    NQ = max(C // 16, 1)
    LP = max(C, 16)
    XSTEP = 16 if pair else C
    best = []
    for CG, CY, CX, eff in col_grids(G, OH, OWc):
        if eff < 0.7:
            continue
        UX = -(-OWc // CX)
        for rowpad in range(0, 256, 16):
            ROW = LP + W * C + LP + rowpad
            TILE = (H + 2) * ROW
            for order in itertools.permutations("gyx"):
                # column index bits: order[0] fastest
                dims = {"g": CG, "y": CY, "x": CX}
                def col_coords(col):
                    c = {}
                    r = col
                    for d in order:
                        c[d] = r % dims[d]
                        r //= dims[d]
                    return c["g"], c["y"], c["x"]
                for tname, t in swz_funcs(min(NQ.bit_length() - 1, 3)):
                    worst, tot, cnt = 0, 0, 0
                    for uy in range(min(OH // CY, 3)):
                        for ux in range(UX):
                            for q in range(min(NQ, 2)):
                                addrs = []
                                for lane in range(64):
                                    col, g = lane & 15, lane >> 4
                                    cg, cy, cx = col_coords(col)
                                    x = min(cx + ux * CX, OWc - 1)
                                    if pair:
                                        a = cg * TILE + ((cy + uy * CY) * S) * ROW + LP + (2 * x - 2) * 8 + g * 16
                                    else:
                                        xin = x * S + g - 1
                                        a = (cg * TILE + ((cy + uy * CY) * S) * ROW + LP + xin * C +
                                             16 * (q ^ (t(xin & 0xff) if NQ > 1 else 0)))
                                    addrs.append(a)
                                c = cycles(addrs, "read_b128")
                                worst = max(worst, c)
                                tot += c
                                cnt += 1
                    best.append((tot / cnt, worst, CG, CY, CX, "".join(order), rowpad, tname, eff))
    best.sort()
    return best


def mid_analyse(shape, grid):
    """MID [pixel][C] with chunk swizzle s(pix): depthwise writes (ds_write_b32) and pointwise reads."""
    H, W, C, S, N, G = shape
    OH, OW = (H + S - 1) // S, (W + S - 1) // S
    if C == 8:
        return []
    NQ = C // 16
    CG, CY, CX, order = grid
    dims = {"g": CG, "y": CY, "x": CX}
    def col_coords(col):
        c = {}
        r = col
        for d in order:
            c[d] = r % dims[d]
            r //= dims[d]
        return c["g"], c["y"], c["x"]
    OPIX = OH * OW
    res = []
    for sname, s in swz_funcs(min(NQ.bit_length() - 1, 3)):
        # writes
        wt, wc = 0, 0
        for ug in range(G // CG):
            for uy in range(OH // CY):
                for ux in range(-(-OW // CX)):
                    for q in range(NQ):
                        for half in range(1):
                            addrs = []
                            for lane in range(64):
                                col, g = lane & 15, lane >> 4
                                cg, cy, cx = col_coords(col)
                                x = cx + ux * CX
                                if x >= OW:
                                    addrs.append(None)
                                    continue
                                pix = (cg + ug * CG) * OPIX + (cy + uy * CY) * OW + x
                                addrs.append(pix * C + 16 * (q ^ s(pix)) + 4 * g)
                            wt += cycles(addrs, "write_b32")
                            wc += 1
        # pointwise reads
        npix = G * OPIX
        rt, rc = 0, 0
        K = C
        if K >= 64:
            for chunk in range(-(-npix // 16)):
                for ks in range(K // 64):
                    addrs = []
                    for lane in range(64):
                        pcol, pg = lane & 15, lane >> 4
                        pix = min(chunk * 16 + pcol, npix - 1)
                        addrs.append(pix * K + 16 * ((pg + 4 * ks) ^ s(pix)))
                    rt += cycles(addrs, "read_b128")
                    rc += 1
        elif K == 32:
            for chunk in range(-(-npix // 32)):
                addrs = []
                for lane in range(64):
                    pcol, pg = lane & 15, lane >> 4
                    pix = min(chunk * 32 + (pg >> 1) * 16 + pcol, npix - 1)
                    addrs.append(pix * 32 + 16 * ((pg & 1) ^ s(pix)))
                rt += cycles(addrs, "read_b128")
                rc += 1
        elif K == 16:
            for chunk in range(-(-npix // 64)):
                addrs = []
                for lane in range(64):
                    pcol, pg = lane & 15, lane >> 4
                    pix = min(chunk * 64 + pg * 16 + pcol, npix - 1)
                    addrs.append(pix * 16)
                rt += cycles(addrs, "read_b128")
                rc += 1
        res.append((wt / wc + rt / rc, wt / wc, rt / rc, sname))
    res.sort()
    return res


if __name__ == "__main__":
    for shp in SHAPES:
        b = analyse(shp)
        print(shp, "tap reads (avg, worst, CG, CY, CX, order, rowpad, tile swizzle, x-efficiency):")
        seen = set()
        shown = 0
        for r in b:
            key = (r[2], r[3], r[4])
            if key in seen:
                continue
            seen.add(key)
            print("    %.2f %d  grid %dx%dx%d order %s rowpad %d swz %s eff %.2f" % r)
            m = mid_analyse(shp, (r[2], r[3], r[4], r[5]))
            for mm in m[:2]:
                print("        MID: write %.2f (ideal 2)  pw read %.2f (ideal 4)  swz %s" % (mm[1], mm[2], mm[3]))
            shown += 1
            if shown >= 4:
                break
