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
