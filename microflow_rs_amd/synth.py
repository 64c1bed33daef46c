"""Deterministic synthetic inputs and checksums shared by tests, smoke() and bench.py.

Counter-based generator: byte g of the stream for BASELINE config
`cfg` is byte (g & 7) of splitmix64(SEED + cfg + (g >> 3)), where g is the GLOBAL
byte index n * elems + i -- so any rank can regenerate exactly its shard of the
batch.  The same function exists on the device (mf_synth_i8 in the C ABI) and the
two are compared bit-for-bit in the GPU tests.
"""
import numpy as np

SEED = 0x4D4643  # "MFC"
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser over uint64 arrays (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_i8(cfg, first, n, elems):
    """int8 [n, elems]: images first .. first+n-1 of config `cfg`'s stream."""
    g0 = np.uint64(first) * np.uint64(elems)
    total = int(n) * int(elems)
    g = g0 + np.arange(total, dtype=np.uint64)
    with np.errstate(over="ignore"):
        w = splitmix64(np.uint64(SEED + cfg) + (g >> np.uint64(3)))
    b = (w >> ((g & np.uint64(7)) * np.uint64(8))) & np.uint64(0xFF)
    return b.astype(np.uint8).view(np.int8).reshape(n, elems)


def layer_checksum(a):
    """Position-sensitive 64-bit checksum of an int8 tensor:
    sum_i (u8(a_i) + 1) * splitmix64(i)  (mod 2^64)."""
    u = np.ascontiguousarray(a).reshape(-1).view(np.uint8).astype(np.uint64)
    with np.errstate(over="ignore"):
        w = splitmix64(np.arange(u.size, dtype=np.uint64))
        return np.uint64(((u + np.uint64(1)) * w).sum(dtype=np.uint64))


def structured_images(side=96):
    """int8 [28, side*side]: constant levels, ramps, checkerboards, Gaussian blobs and noisy versions of them.
    Uniform-noise inputs drive person_detect into nearly the same output for every image, which makes the final bytes a
    weak witness of the late layers; these spread the outputs over the whole range (bench.py and the full-size test)."""
    yy, xx = np.mgrid[0:side, 0:side]
    k = (side - 1) / 255.0 * 0 + 255.0 / (side - 1)
    imgs = [np.full((side, side), v) for v in (-128, -90, -40, -1, 0, 37, 90, 127)]
    imgs += [xx * k - 128, yy * k - 128, (xx + yy) * k / 2 - 128, 127 - xx * k]
    imgs += [np.where(((xx // c) + (yy // c)) % 2 == 0, 100, -100) for c in (1, 4, 16)]
    imgs += [120 * np.exp(-((xx - cx * side) ** 2 + (yy - cy * side) ** 2) / (2.0 * (sg * side) ** 2)) - 100
             for cx, cy, sg in ((0.5, 0.5, 0.1), (0.2, 0.73, 0.19), (0.73, 0.31, 0.31), (0.5, 0.5, 0.42), (0.1, 0.1, 0.06))]
    rng = np.random.default_rng(11)
    imgs += [np.clip(im + rng.normal(0, 12, (side, side)), -128, 127) for im in imgs[8:16]]
    return np.clip(np.round(np.stack(imgs)), -128, 127).astype(np.int8).reshape(len(imgs), -1)
