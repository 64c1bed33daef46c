#!/usr/bin/env python3
"""Rewrites an INT8 .tflite model as the equivalent UINT8 model (test input for the `T = u8`
instantiation of the operator path; the reference ships no u8 model, only the generic code
-- src/quantize.rs:43-53, microflow-macros/src/lib.rs:118-128).

The file is patched in place, nothing moves:
  * every INT8 tensor's `type` byte becomes UINT8 (tflite.fbs TensorType: INT8 = 9, UINT8 = 3),
  * every such tensor's zero points get +128,
  * the bytes of every buffer an INT8 tensor owns get ^ 0x80 (value + 128).
Real values are unchanged (scale * (q - zp)), so the u8 model computes the same function up
to the f32 rounding differences of the reference's u8 arithmetic -- which is exactly what the
u8 oracle / kernel parity tests exercise.

    python tools/make_u8_model.py models/person_detect.tflite out_u8.tflite
"""
import struct
import sys

TT_UINT8, TT_INT8 = 3, 9


class Reader:
    """Just enough FlatBuffers navigation to find field positions."""

    def __init__(self, buf):
        self.b = buf

    def u16(self, p):
        return struct.unpack_from("<H", self.b, p)[0]

    def i32(self, p):
        return struct.unpack_from("<i", self.b, p)[0]

    def u32(self, p):
        return struct.unpack_from("<I", self.b, p)[0]

    def field(self, table, idx):
        """absolute position of field `idx` of the table at `table`, or 0 when absent"""
        vt = table - self.i32(table)
        if 4 + 2 * idx >= self.u16(vt):
            return 0
        off = self.u16(vt + 4 + 2 * idx)
        return table + off if off else 0

    def indirect(self, p):
        return p + self.u32(p)

    def vector(self, table, idx):
        """(position of element 0, length) of a vector field, or (0, 0)"""
        f = self.field(table, idx)
        if not f:
            return 0, 0
        v = self.indirect(f)
        return v + 4, self.u32(v)


def to_u8(data):
    buf = bytearray(data)
    r = Reader(buf)
    model = r.indirect(0)
    subgraphs, n_sg = r.vector(model, 2)    # Model { subgraphs:2 buffers:4 }
    buffers, n_buf = r.vector(model, 4)
    if n_sg < 1:
        raise ValueError("no subgraph")
    sg = r.indirect(subgraphs)
    tensors, n_t = r.vector(sg, 0)          # SubGraph { tensors:0 }
    flipped, flipped_zp = set(), set()
    count = 0
    for i in range(n_t):
        t = r.indirect(tensors + 4 * i)     # Tensor { shape:0 type:1 buffer:2 name:3 quantization:4 }
        tf = r.field(t, 1)
        if not tf or buf[tf] != TT_INT8:
            continue
        buf[tf] = TT_UINT8
        count += 1
        qf = r.field(t, 4)
        if qf:                              # QuantizationParameters { scale:2 zero_point:3 }
            q = r.indirect(qf)
            zp, n_zp = r.vector(q, 3)
            if zp in flipped_zp:            # a vector shared by two tensors is shifted once
                n_zp = 0
            flipped_zp.add(zp)
            for k in range(n_zp):
                (z,) = struct.unpack_from("<q", buf, zp + 8 * k)
                struct.pack_into("<q", buf, zp + 8 * k, z + 128)
        bf = r.field(t, 2)
        bi = r.u32(bf) if bf else 0
        if bi and bi < n_buf and bi not in flipped:
            b = r.indirect(buffers + 4 * bi)  # Buffer { data:0 }
            d, n = r.vector(b, 0)
            for k in range(n):
                buf[d + k] ^= 0x80
            flipped.add(bi)
    if not count:
        raise ValueError("model has no INT8 tensor")
    return bytes(buf)


def main():
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    with open(sys.argv[1], "rb") as f:
        out = to_u8(f.read())
    with open(sys.argv[2], "wb") as f:
        f.write(out)


if __name__ == "__main__":
    main()
