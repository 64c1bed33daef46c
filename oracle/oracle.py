"""ctypes binding of the CPU oracle (oracle/mf_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.  The oracle is a
plain-C restatement of the reference algorithm (see mf_oracle.h for the
file:line citations and the parity-pinning statement).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmf_oracle.so")

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 3
PAD_SAME, PAD_VALID = 0, 1
OP_AVERAGE_POOL_2D, OP_CONV_2D, OP_DEPTHWISE_CONV_2D = 1, 3, 4
OP_FULLY_CONNECTED, OP_RESHAPE, OP_SOFTMAX = 9, 22, 25
OP_NAMES = {1: "average_pool_2d", 3: "conv_2d", 4: "depthwise_conv_2d", 9: "fully_connected",
            22: "reshape", 25: "softmax"}


# -O3 as SURVEY.md 8d asks of the CPU baseline; -ffp-contract=off / no fast-math keep every f32 operation
# individually rounded (value-safe at any -O level), so the same build serves as checker and as baseline
CFLAGS = ["-O3", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]


def build(force=False):
    """Compile oracle/libmf_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "mf_oracle.c")
    deps = [src, os.path.join(_HERE, "mf_oracle.h"), os.path.join(_HERE, "mf_oracle_ops.inc")]
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(d) for d in deps)):
        return _SO
    subprocess.check_call(["gcc"] + CFLAGS + ["-shared", "-o", _SO, src, "-lm"])
    return _SO


class _OpInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("in_shape", C.c_int * 4), ("in_rank", C.c_int),
                ("out_shape", C.c_int * 4), ("out_rank", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("sh", C.c_int), ("sw", C.c_int),
                ("pad", C.c_int), ("act", C.c_int), ("n_c0", C.c_int), ("n_c1", C.c_int),
                ("in_scale", C.c_float), ("out_scale", C.c_float),
                ("in_zp", C.c_int), ("out_zp", C.c_int), ("out_elems", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    i8p, f32p, i32p, u8p = (C.POINTER(C.c_int8), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                            C.POINTER(C.c_uint8))
    L.orc_roundf.restype = C.c_float
    L.orc_roundf.argtypes = [C.c_float]
    L.orc_expf.restype = C.c_float
    L.orc_expf.argtypes = [C.c_float]
    L.orc_sat_i8.restype = C.c_int8
    L.orc_sat_i8.argtypes = [C.c_float]
    L.orc_quantize.restype = C.c_int8
    L.orc_quantize.argtypes = [C.c_float, C.c_float, C.c_int8]
    L.orc_dequantize.restype = C.c_float
    L.orc_dequantize.argtypes = [C.c_int8, C.c_float, C.c_int8]
    L.orc_relu.restype = C.c_int8
    L.orc_relu.argtypes = [C.c_int8, C.c_int8]
    L.orc_relu6.restype = C.c_int8
    L.orc_relu6.argtypes = [C.c_int8, C.c_float, C.c_int8]
    L.orc_softmax_scalar.restype = C.c_int8
    L.orc_softmax_scalar.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int8]
    L.orc_view.restype = C.c_int
    L.orc_view.argtypes = [i8p] + [C.c_int] * 10 + [i8p, u8p]
    L.orc_fully_connected.restype = None
    L.orc_fully_connected.argtypes = [i8p, C.c_int, C.c_int, i8p, C.c_int, C.c_int8, C.c_float,
                                      C.c_int8, C.c_int, f32p, C.c_float, i32p, C.c_int32, i8p]
    L.orc_conv_2d.restype = C.c_int
    L.orc_conv_2d.argtypes = [i8p, C.c_int, C.c_int, C.c_int, i8p, C.c_int, C.c_int, C.c_int, i8p,
                              C.c_int, C.c_int8, C.c_float, C.c_int8, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, i8p]
    L.orc_depthwise_conv_2d.restype = C.c_int
    L.orc_depthwise_conv_2d.argtypes = L.orc_conv_2d.argtypes
    L.orc_average_pool_2d.restype = C.c_int
    L.orc_average_pool_2d.argtypes = [i8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_int8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_float, C.c_float, i8p]
    L.orc_softmax.restype = None
    L.orc_softmax.argtypes = [i8p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int8, i8p]
    L.orc_preprocess_fully_connected.restype = None
    L.orc_preprocess_fully_connected.argtypes = [C.c_float, C.c_int8, C.c_int, i8p, C.c_int,
                                                 C.c_int, C.c_float, C.c_int8, i32p, C.c_float,
                                                 C.c_int32, C.c_float, f32p, f32p, i32p, i32p]
    L.orc_preprocess_conv.restype = None
    L.orc_preprocess_conv.argtypes = [C.c_float, C.c_int, i32p, f32p, i32p, C.c_int, f32p, C.c_int,
                                      C.c_float, f32p, f32p]
    L.orc_preprocess_average_pool_2d.restype = None
    L.orc_preprocess_average_pool_2d.argtypes = [C.c_float, C.c_int8, C.c_float, C.c_int8, f32p,
                                                 f32p]
    # T = u8 instantiations: same signatures with every i8 replaced by u8
    for n in ("orc_quantize", "orc_dequantize", "orc_relu", "orc_relu6", "orc_softmax_scalar",
              "orc_view", "orc_fully_connected", "orc_conv_2d", "orc_depthwise_conv_2d",
              "orc_average_pool_2d", "orc_softmax", "orc_preprocess_fully_connected"):
        f, g = getattr(L, n), getattr(L, n + "_u8")
        sw = {C.c_int8: C.c_uint8, i8p: u8p}
        g.restype = sw.get(f.restype, f.restype)
        g.argtypes = [sw.get(t, t) for t in f.argtypes]
    L.orc_sat_u8.restype = C.c_uint8
    L.orc_sat_u8.argtypes = [C.c_float]
    L.orc_model_is_u8.restype = C.c_int
    L.orc_model_is_u8.argtypes = [C.c_void_p]
    L.orc_model_load.restype = C.c_void_p
    L.orc_model_load.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p)]
    L.orc_model_free.restype = None
    L.orc_model_free.argtypes = [C.c_void_p]
    L.orc_model_num_ops.restype = C.c_int
    L.orc_model_num_ops.argtypes = [C.c_void_p]
    L.orc_model_op_info.restype = C.c_int
    L.orc_model_op_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(_OpInfo)]
    L.orc_model_op_constants.restype = C.c_int
    L.orc_model_op_constants.argtypes = [C.c_void_p, C.c_int, f32p, f32p, i32p, i32p]
    for n in ("orc_model_input_elems", "orc_model_output_elems", "orc_model_layers_elems"):
        getattr(L, n).restype = C.c_size_t
        getattr(L, n).argtypes = [C.c_void_p]
    L.orc_model_io_quant.restype = None
    L.orc_model_io_quant.argtypes = [C.c_void_p, f32p, C.POINTER(C.c_int), f32p,
                                     C.POINTER(C.c_int)]
    L.orc_model_io_shape.restype = None
    L.orc_model_io_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_model_run_quantized.restype = C.c_int
    L.orc_model_run_quantized.argtypes = [C.c_void_p, i8p, i8p, i8p]
    L.orc_model_predict_quantized.restype = C.c_int
    L.orc_model_predict_quantized.argtypes = [C.c_void_p, i8p, f32p]
    L.orc_model_predict.restype = C.c_int
    L.orc_model_predict.argtypes = [C.c_void_p, f32p, f32p]
    L.orc_model_run_quantized_batch.restype = C.c_int
    L.orc_model_run_quantized_batch.argtypes = [C.c_void_p, i8p, C.c_size_t, i8p]
    _lib = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


# Element type (the reference's `T: Quantized`): decided by the dtype of the input array --
# np.uint8 selects the u8 instantiation, anything else i8.
def _dt(a):
    return np.uint8 if getattr(a, "dtype", None) == np.uint8 else np.int8


def _ct(dt):
    return C.c_uint8 if dt == np.uint8 else C.c_int8


def _fn(name, dt):
    return getattr(lib(), name + ("_u8" if dt == np.uint8 else ""))


def _q(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _f32(a):
    return np.ascontiguousarray(np.atleast_1d(a), dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(np.atleast_1d(a), dtype=np.int32)


# ---- scalar primitives ---------------------------------------------------
def roundf(x):
    return lib().orc_roundf(np.float32(x))


def expf(x):
    return lib().orc_expf(np.float32(x))


def quantize(x, scale, zp, dt=np.int8):
    return int(_fn("orc_quantize", dt)(np.float32(x), np.float32(scale), int(zp)))


def dequantize(q, scale, zp, dt=np.int8):
    return np.float32(_fn("orc_dequantize", dt)(int(q), np.float32(scale), int(zp)))


def relu(x, zp, dt=np.int8):
    return int(_fn("orc_relu", dt)(int(x), int(zp)))


def relu6(x, scale, zp, dt=np.int8):
    return int(_fn("orc_relu6", dt)(int(x), np.float32(scale), int(zp)))


def softmax_scalar(x, s, scale, zp, dt=np.int8):
    return int(_fn("orc_softmax_scalar", dt)(np.float32(x), np.float32(s), np.float32(scale),
                                             int(zp)))


def quantize_array(x, scale, zp, dt=np.int8):
    x = np.asarray(x, dtype=np.float32)
    out = np.empty(x.shape, dt)
    f = _fn("orc_quantize", dt)
    xf, of = x.reshape(-1), out.reshape(-1)
    for i in range(xf.size):
        of[i] = f(xf[i], np.float32(scale), int(zp))
    return out


# ---- view ------------------------------------------------------------------
def view(inp, focus, kshape, pad, strides):
    """inp [H][W][C] int8 -> (buffer [KH][KW][C], mask [KH][KW] bool, len)."""
    dt = _dt(inp)
    ct = _ct(dt)
    inp = _q(inp, dt)
    H, W, Cc = inp.shape
    KH, KW = kshape
    buf = np.zeros((KH, KW, Cc), dt)
    mask = np.zeros((KH, KW), np.uint8)
    n = _fn("orc_view", dt)(_p(inp, ct), H, W, Cc, focus[0], focus[1], KH, KW, pad, strides[0],
                       strides[1], _p(buf, ct), _p(mask, C.c_uint8))
    return buf, mask.astype(bool), n


# ---- operators -------------------------------------------------------------
def fully_connected(inp, w_nk, wzp, oscale, ozp, act, c0, c1, c2, c3):
    """inp [M][K]; w_nk [N][K] (TFLite order)."""
    dt = _dt(inp)
    ct = _ct(dt)
    inp, w_nk = _q(inp, dt), _q(w_nk, dt)
    M, K = inp.shape
    N = w_nk.shape[0]
    assert w_nk.shape[1] == K
    c0, c2 = _f32(c0), _i32(c2)
    out = np.empty((M, N), dt)
    _fn("orc_fully_connected", dt)(_p(inp, ct), M, K, _p(w_nk, ct), N, int(wzp),
                              np.float32(oscale), int(ozp), act, _p(c0, C.c_float), np.float32(c1),
                              _p(c2, C.c_int32), int(c3), _p(out, ct))
    return out


def conv_2d(inp, filters, fzp, izp, oscale, ozp, act, pad, strides, out_hw, c0, c1):
    """inp [H][W][C]; filters [N][KH][KW][C] -> [OH][OW][N]."""
    dt = _dt(inp)
    ct = _ct(dt)
    inp, filters, fzp = _q(inp, dt), _q(filters, dt), _q(np.atleast_1d(fzp), dt)
    H, W, Cc = inp.shape
    N, KH, KW, C2 = filters.shape
    assert C2 == Cc
    c0, c1 = _f32(c0), _f32(c1)
    OH, OW = out_hw
    out = np.empty((OH, OW, N), dt)
    rc = _fn("orc_conv_2d", dt)(_p(inp, ct), H, W, Cc, _p(filters, ct), N, KH, KW,
                           _p(fzp, ct), fzp.size, int(izp), np.float32(oscale), int(ozp), act,
                           pad, strides[0], strides[1], OH, OW, _p(c0, C.c_float),
                           _p(c1, C.c_float), c1.size, _p(out, ct))
    if rc:
        raise ValueError("orc_conv_2d: view out of range")
    return out


def depthwise_conv_2d(inp, weights, wzp, izp, oscale, ozp, act, pad, strides, out_hw, c0, c1):
    """inp [H][W][Cin]; weights [KH][KW][WC] (leading 1 optional) -> [OH][OW][WC]."""
    dt = _dt(inp)
    ct = _ct(dt)
    inp, weights, wzp = _q(inp, dt), _q(weights, dt), _q(np.atleast_1d(wzp), dt)
    if weights.ndim == 4:
        weights = weights[0]
    H, W, Cin = inp.shape
    KH, KW, WC = weights.shape
    c0, c1 = _f32(c0), _f32(c1)
    OH, OW = out_hw
    out = np.empty((OH, OW, WC), dt)
    rc = _fn("orc_depthwise_conv_2d", dt)(_p(inp, ct), H, W, Cin, _p(weights, ct), KH, KW,
                                     WC, _p(wzp, ct), wzp.size, int(izp), np.float32(oscale),
                                     int(ozp), act, pad, strides[0], strides[1], OH, OW,
                                     _p(c0, C.c_float), _p(c1, C.c_float), c1.size,
                                     _p(out, ct))
    if rc:
        raise ValueError("orc_depthwise_conv_2d: view out of range")
    return out


def average_pool_2d(inp, fshape, oscale, ozp, act, pad, strides, out_hw, c0, c1):
    dt = _dt(inp)
    ct = _ct(dt)
    inp = _q(inp, dt)
    H, W, Cc = inp.shape
    OH, OW = out_hw
    out = np.empty((OH, OW, Cc), dt)
    rc = _fn("orc_average_pool_2d", dt)(_p(inp, ct), H, W, Cc, fshape[0], fshape[1],
                                   np.float32(oscale), int(ozp), act, pad, strides[0], strides[1],
                                   OH, OW, np.float32(c0), np.float32(c1), _p(out, ct))
    if rc:
        raise ValueError("orc_average_pool_2d: view out of range")
    return out


def softmax(inp, iscale, oscale, ozp):
    dt = _dt(inp)
    ct = _ct(dt)
    inp = _q(inp, dt)
    rows, cols = inp.shape
    out = np.empty((rows, cols), dt)
    _fn("orc_softmax", dt)(_p(inp, ct), rows, cols, np.float32(iscale), np.float32(oscale),
                      int(ozp), _p(out, ct))
    return out


# ---- preprocess ------------------------------------------------------------
def preprocess_fully_connected(iscale, izp, in_shape1, w_nk, wscale, wzp, bias, bscale, bzp,
                               oscale):
    dt = _dt(w_nk)
    ct = _ct(dt)
    w_nk, bias = _q(w_nk, dt), _i32(bias)
    N, K = w_nk.shape
    c0 = np.empty(N, np.float32)
    c1 = np.empty(1, np.float32)
    c2 = np.empty(N, np.int32)
    c3 = np.empty(1, np.int32)
    _fn("orc_preprocess_fully_connected", dt)(np.float32(iscale), int(izp), int(in_shape1),
                                         _p(w_nk, ct), K, N, np.float32(wscale), int(wzp),
                                         _p(bias, C.c_int32), np.float32(bscale), int(bzp),
                                         np.float32(oscale), _p(c0, C.c_float), _p(c1, C.c_float),
                                         _p(c2, C.c_int32), _p(c3, C.c_int32))
    return c0, np.float32(c1[0]), c2, int(c3[0])


def preprocess_conv(iscale, bias, bscale, bzp, fscale, oscale):
    bias, bscale, bzp, fscale = _i32(bias), _f32(bscale), _i32(bzp), _f32(fscale)
    n = bias.size
    c0 = np.empty(n, np.float32)
    c1 = np.empty(fscale.size, np.float32)
    lib().orc_preprocess_conv(np.float32(iscale), n, _p(bias, C.c_int32), _p(bscale, C.c_float),
                              _p(bzp, C.c_int32), min(bscale.size, bzp.size),
                              _p(fscale, C.c_float), fscale.size, np.float32(oscale),
                              _p(c0, C.c_float), _p(c1, C.c_float))
    return c0, c1


def preprocess_average_pool_2d(iscale, izp, oscale, ozp):
    c0 = np.empty(1, np.float32)
    c1 = np.empty(1, np.float32)
    lib().orc_preprocess_average_pool_2d(np.float32(iscale), int(izp), np.float32(oscale),
                                         int(ozp), _p(c0, C.c_float), _p(c1, C.c_float))
    return np.float32(c0[0]), np.float32(c1[0])


# ---- whole model -----------------------------------------------------------
class Model:
    """Oracle-side model: the CPU restatement of #[model("x.tflite")]."""

    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray)):
            data = bytes(path_or_bytes)
        else:
            with open(path_or_bytes, "rb") as f:
                data = f.read()
        err = C.c_char_p()
        self._h = lib().orc_model_load(data, len(data), C.byref(err))
        if not self._h:
            raise ValueError("oracle: " + (err.value.decode() if err.value else "load failed"))
        L = lib()
        self.num_ops = L.orc_model_num_ops(self._h)
        self.dtype = np.uint8 if L.orc_model_is_u8(self._h) else np.int8  # element type T
        self.in_elems = L.orc_model_input_elems(self._h)
        self.out_elems = L.orc_model_output_elems(self._h)
        self.layers_elems = L.orc_model_layers_elems(self._h)
        s_i, s_o = C.c_float(), C.c_float()
        z_i, z_o = C.c_int(), C.c_int()
        L.orc_model_io_quant(self._h, C.byref(s_i), C.byref(z_i), C.byref(s_o), C.byref(z_o))
        self.in_scale, self.in_zp = np.float32(s_i.value), z_i.value
        self.out_scale, self.out_zp = np.float32(s_o.value), z_o.value
        ish, osh = (C.c_int * 4)(), (C.c_int * 4)()
        ir, orr = C.c_int(), C.c_int()
        L.orc_model_io_shape(self._h, ish, C.byref(ir), osh, C.byref(orr))
        self.in_shape = tuple(ish[: ir.value])
        self.out_shape = tuple(osh[: orr.value])
        self.ops = []
        for i in range(self.num_ops):
            info = _OpInfo()
            L.orc_model_op_info(self._h, i, C.byref(info))
            self.ops.append(dict(
                kind=info.kind, name=OP_NAMES.get(info.kind, "?"),
                in_shape=tuple(info.in_shape[: info.in_rank]),
                out_shape=tuple(info.out_shape[: info.out_rank]),
                KH=info.KH, KW=info.KW, sh=info.sh, sw=info.sw, pad=info.pad, act=info.act,
                n_c0=info.n_c0, n_c1=info.n_c1, in_scale=np.float32(info.in_scale),
                out_scale=np.float32(info.out_scale), in_zp=info.in_zp, out_zp=info.out_zp,
                out_elems=info.out_elems))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            _lib.orc_model_free(h)
            self._h = None

    def _bytes(self, q):
        """Quantized input as the raw bytes the C side takes (int8_t* carries u8 for a u8 model)."""
        q = np.asarray(q)
        if q.dtype != self.dtype:
            if q.dtype.kind in "iu" and q.dtype.itemsize == 1:
                raise TypeError(f"model element type is {np.dtype(self.dtype).name}, got {q.dtype}")
            q = q.astype(self.dtype)
        return np.ascontiguousarray(q).view(np.int8)

    def op_constants(self, i):
        op = self.ops[i]
        c0 = np.zeros(max(op["n_c0"], 1), np.float32)
        c1 = np.zeros(max(op["n_c1"], 1), np.float32)
        c2 = np.zeros(max(op["n_c0"], 1), np.int32)
        c3 = np.zeros(1, np.int32)
        lib().orc_model_op_constants(self._h, i, _p(c0, C.c_float), _p(c1, C.c_float),
                                     _p(c2, C.c_int32), _p(c3, C.c_int32))
        return c0, c1, c2, int(c3[0])

    def run_quantized(self, in_q, layers=False):
        """predict_inner on one input; returns int8 output (and per-op outputs)."""
        in_q = self._bytes(in_q).reshape(-1)
        assert in_q.size == self.in_elems
        out = np.empty(self.out_elems, np.int8)
        lay = np.empty(self.layers_elems, np.int8) if layers else None
        rc = lib().orc_model_run_quantized(self._h, _p(in_q, C.c_int8), _p(out, C.c_int8),
                                           _p(lay, C.c_int8) if layers else None)
        if rc:
            raise RuntimeError("oracle run failed")
        out = out.view(self.dtype)
        if not layers:
            return out
        lay = lay.view(self.dtype)
        outs, off = [], 0
        for op in self.ops:
            outs.append(lay[off: off + op["out_elems"]].reshape(op["out_shape"]))
            off += op["out_elems"]
        return out, outs

    def run_quantized_batch(self, in_q):
        in_q = self._bytes(in_q).reshape(-1, self.in_elems)
        n = in_q.shape[0]
        out = np.empty((n, self.out_elems), np.int8)
        rc = lib().orc_model_run_quantized_batch(self._h, _p(in_q, C.c_int8), n,
                                                 _p(out, C.c_int8))
        if rc:
            raise RuntimeError("oracle run failed")
        return out.view(self.dtype)

    def predict_quantized(self, in_q):
        in_q = self._bytes(in_q).reshape(-1)
        assert in_q.size == self.in_elems
        out = np.empty(self.out_elems, np.float32)
        rc = lib().orc_model_predict_quantized(self._h, _p(in_q, C.c_int8), _p(out, C.c_float))
        if rc:
            raise RuntimeError("oracle run failed")
        return out.reshape(self.out_shape)

    def predict(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        assert x.size == self.in_elems
        out = np.empty(self.out_elems, np.float32)
        rc = lib().orc_model_predict(self._h, _p(x, C.c_float), _p(out, C.c_float))
        if rc:
            raise RuntimeError("oracle run failed")
        return out.reshape(self.out_shape)
