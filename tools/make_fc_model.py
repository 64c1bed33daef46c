#!/usr/bin/env python3
"""Writes a single-operator FullyConnected .tflite model (BASELINE config 5: the synthetic
int8 GEMM run through predict()).  A tiny forward FlatBuffers writer: every table is
emitted before the objects it references, so all uoffsets point forward as the format
requires.  Only the fields the MicroFlow front end reads are written
(microflow-macros/src/lib.rs:46-151, microflow-macros/src/ops/fully_connected.rs:66-98).

    python tools/make_fc_model.py out.tflite --m 4096 --k 4096 --n 4096 [--wzp 0] [--seed 5]
"""
import argparse
import struct

import numpy as np


class FB:
    """Forward FlatBuffers writer.  Objects are described as Python values:
    Table(fields), bytes/np arrays for [ubyte], IntVec, FloatVec, LongVec, TableVec."""

    def __init__(self):
        self.buf = bytearray()

    def align(self, n):
        while len(self.buf) % n:
            self.buf.append(0)

    def u32_at(self, pos, v):
        struct.pack_into("<I", self.buf, pos, v)


class Table:
    def __init__(self, fields):
        """fields: {id: ('i8'|'u8'|'i32'|'u32'|'f32', value) | ('ref', obj)}"""
        self.fields = fields


class Vec:
    def __init__(self, fmt, values, align=4):
        self.fmt, self.values, self.align_to = fmt, values, align


class TableVec:
    def __init__(self, tables):
        self.tables = tables


SCALAR = {"i8": ("<b", 1), "u8": ("<B", 1), "i32": ("<i", 4), "u32": ("<I", 4), "f32": ("<f", 4)}


def write_table(fb, t):
    """Emit vtable + table at the current end of the buffer; return (table_pos, [(slot_pos, obj)])."""
    ids = sorted(t.fields)
    nslots = (max(ids) + 1) if ids else 0
    # layout of the inline part: soffset(4) then fields, biggest first for alignment
    order = sorted(ids, key=lambda i: -(4 if t.fields[i][0] == "ref" else SCALAR[t.fields[i][0]][1]))
    offs, cur = {}, 4
    for i in order:
        size = 4 if t.fields[i][0] == "ref" else SCALAR[t.fields[i][0]][1]
        cur = (cur + size - 1) // size * size
        offs[i] = cur
        cur += size
    tsize = (cur + 3) // 4 * 4
    vsize = 4 + 2 * nslots
    fb.align(2)
    # make the table itself 4-aligned: pad so that (len + vsize) % 4 == 0
    while (len(fb.buf) + vsize) % 4:
        fb.buf.append(0)
    vpos = len(fb.buf)
    fb.buf += struct.pack("<HH", vsize, tsize)
    for i in range(nslots):
        fb.buf += struct.pack("<H", offs.get(i, 0))
    tpos = len(fb.buf)
    fb.buf += bytes(tsize)
    struct.pack_into("<i", fb.buf, tpos, tpos - vpos)
    refs = []
    for i in ids:
        kind, val = t.fields[i]
        if kind == "ref":
            refs.append((tpos + offs[i], val))
        else:
            struct.pack_into(SCALAR[kind][0], fb.buf, tpos + offs[i], val)
    return tpos, refs


def emit(fb, obj):
    """Emit obj (and, after it, everything it references); return its position."""
    if isinstance(obj, Table):
        pos, refs = write_table(fb, obj)
        for slot, child in refs:
            cpos = emit(fb, child)
            fb.u32_at(slot, cpos - slot)
        return pos
    if isinstance(obj, TableVec):
        fb.align(4)
        pos = len(fb.buf)
        fb.buf += struct.pack("<I", len(obj.tables)) + bytes(4 * len(obj.tables))
        for i, t in enumerate(obj.tables):
            slot = pos + 4 + 4 * i
            cpos = emit(fb, t)
            fb.u32_at(slot, cpos - slot)
        return pos
    if isinstance(obj, Vec):
        # the length word sits right before the (aligned) payload
        while (len(fb.buf) + 4) % obj.align_to:
            fb.buf.append(0)
        pos = len(fb.buf)
        payload = obj.values if isinstance(obj.values, (bytes, bytearray)) else \
            b"".join(struct.pack(obj.fmt, v) for v in obj.values)
        n = len(obj.values) if not isinstance(obj.values, (bytes, bytearray)) else len(payload)
        fb.buf += struct.pack("<I", n) + payload
        return pos
    raise TypeError(obj)


def quant(scale, zp):
    # QuantizationParameters { scale:2 zero_point:3 }
    return Table({2: ("ref", Vec("<f", [scale])), 3: ("ref", Vec("<q", [zp], align=8))})


def tensor(shape, ttype, buffer, scale, zp):
    # Tensor { shape:0 type:1 buffer:2 quantization:4 }
    return Table({0: ("ref", Vec("<i", list(shape))), 1: ("i8", ttype), 2: ("u32", buffer),
                  4: ("ref", quant(scale, zp))})


def fc_model(M, K, N, weights, bias, in_q, w_q, b_q, out_q, activation=0):
    """weights [N][K] int8, bias [N] int32; *_q = (scale, zero_point)."""
    INT32, INT8 = 2, 9
    tensors = TableVec([
        tensor((M, K), INT8, 0, *in_q),
        tensor((N, K), INT8, 1, *w_q),
        tensor((N,), INT32, 2, *b_q),
        tensor((M, N), INT8, 0, *out_q),
    ])
    # Operator { opcode_index:0 inputs:1 outputs:2 builtin_options_type:3 builtin_options:4 }
    op = Table({0: ("u32", 0), 1: ("ref", Vec("<i", [0, 1, 2])), 2: ("ref", Vec("<i", [3])),
                3: ("u8", 8),  # BuiltinOptions.FullyConnectedOptions
                4: ("ref", Table({0: ("i8", activation)}))})
    subgraph = Table({0: ("ref", tensors), 1: ("ref", Vec("<i", [0])), 2: ("ref", Vec("<i", [3])),
                      3: ("ref", TableVec([op]))})
    buffers = TableVec([
        Table({}),  # buffer 0: the always-empty buffer
        Table({0: ("ref", Vec(None, np.ascontiguousarray(weights, np.int8).tobytes(), align=16))}),
        Table({0: ("ref", Vec(None, np.ascontiguousarray(bias, np.int32).tobytes(), align=16))}),
    ])
    # Model { version:0 operator_codes:1 subgraphs:2 buffers:4 }
    model = Table({0: ("u32", 3),
                   1: ("ref", TableVec([Table({0: ("i8", 9), 3: ("i32", 9)})])),  # FULLY_CONNECTED
                   2: ("ref", TableVec([subgraph])),
                   4: ("ref", buffers)})
    fb = FB()
    fb.buf += bytes(4) + b"TFL3"      # root uoffset + file identifier
    root = emit(fb, model)
    fb.u32_at(0, root)
    return bytes(fb.buf)


def synthetic_fc(M, K, N, wzp=0, seed=5, activation=0, u8=False):
    """BASELINE config 5: uniform int8 weights, input zp -128, scales chosen so that the
    outputs spread over the int8 range instead of saturating.
    u8=True: the same model with UINT8 tensors (`wzp` then is a u8 value)."""
    if u8:
        from make_u8_model import to_u8
        return to_u8(synthetic_fc(M, K, N, wzp - 128, seed, activation))
    rng = np.random.default_rng(seed)
    w = rng.integers(-128, 128, (N, K), dtype=np.int8)
    bias = rng.integers(-4096, 4096, N).astype(np.int64)
    # a non-zero weight zero point shifts every accumulator by about -E[x - izp] * K * wzp:
    # compensate in the bias so that the outputs still spread instead of saturating
    bias = (bias + int(round(127.5 * K * wzp))).astype(np.int32)
    in_scale, w_scale = 1.0 / 128.0, 1.0 / 128.0
    # std of the accumulator ~ sqrt(K) * 74 * 74 ; map ~3 sigma onto the int8 range
    out_scale = float(np.float32(in_scale * w_scale * np.sqrt(K) * 74.0 * 74.0 * 3.0 / 127.0))
    data = fc_model(M, K, N, w, bias, (in_scale, -128), (w_scale, wzp),
                    (in_scale * w_scale, 0), (out_scale, 3), activation)
    return data


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--wzp", type=int, default=0)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--u8", action="store_true", help="UINT8 tensors (wzp is then a u8 value)")
    a = ap.parse_args()
    blob = synthetic_fc(a.m, a.k, a.n, a.wzp, a.seed, u8=a.u8)
    open(a.out, "wb").write(blob)
    print(a.out, len(blob), "bytes")
