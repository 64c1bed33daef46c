"""microflow::ops -- the reference's operator functions (src/ops/mod.rs) with the same
names, argument order and meaning, executed by the HIP kernels behind the C ABI.

    fully_connected(input, weights, output_scale, output_zero_point, options, constants)
    conv_2d(input, filters, output_scale, output_zero_point, options, constants, output_shape)
    depthwise_conv_2d(input, weights, output_scale, output_zero_point, options, constants, output_shape)
    average_pool_2d(input, filter_shape, output_scale, output_zero_point, options, constants, output_shape)
    softmax(input, output_scale, output_zero_point)
    reshape(input, output_shape)

Differences forced by the language: output shapes are const generics in Rust and an
explicit `output_shape=(rows, cols)` here; buffers are row-major / NHWC arrays
(include/microflow_amd.h).  A leading batch of independent inferences is allowed on
every operator (`input.buffer` may have one extra leading dimension).
"""
import ctypes as C
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from . import _lib
from .tensor import FusedActivation, Tensor2D, Tensor4D, TensorViewPadding


@dataclass
class FullyConnectedOptions:              # src/ops/fully_connected.rs:9-11
    fused_activation: FusedActivation = FusedActivation.NONE


@dataclass
class Conv2DOptions:                      # src/ops/conv_2d.rs:11-15
    fused_activation: FusedActivation = FusedActivation.NONE
    view_padding: TensorViewPadding = TensorViewPadding.SAME
    strides: Tuple[int, int] = (1, 1)


@dataclass
class DepthwiseConv2DOptions:             # src/ops/depthwise_conv_2d.rs:11-15
    fused_activation: FusedActivation = FusedActivation.NONE
    view_padding: TensorViewPadding = TensorViewPadding.SAME
    strides: Tuple[int, int] = (1, 1)


@dataclass
class AveragePool2DOptions:               # src/ops/average_pool_2d.rs:12-16
    fused_activation: FusedActivation = FusedActivation.NONE
    view_padding: TensorViewPadding = TensorViewPadding.SAME
    strides: Tuple[int, int] = (1, 1)


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _lib.MicroflowError(_lib.MF_ERR_NO_DEVICE,
                                  "no GPU visible: microflow ops have no CPU fallback")
    return torch


def _host(a, dtype):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _elem(a):
    """Element type T (src/quantize.rs:32-53) of a buffer: np.uint8 -> u8, everything else i8."""
    dt = getattr(a, "dtype", None)
    if dt is None:
        return np.int8
    try:
        import torch
        if dt == torch.uint8:
            return np.uint8
    except ImportError:
        pass
    return np.uint8 if dt == np.uint8 else np.int8


def _fn(name, dtype):
    return getattr(_lib.lib(), name + ("_u8" if dtype == np.uint8 else ""))


class PreparedOp:
    """A prepared operator (mf_op): weights + folded constants resident in HBM."""

    def __init__(self, handle, in_tail, out_tail, dtype=np.int8):
        self._h = handle
        self.in_tail, self.out_tail = tuple(in_tail), tuple(out_tail)
        self.dtype = dtype

    @property
    def kernel(self):
        return _lib.lib().mf_op_kernel_name(self._h).decode()

    def set_generic(self, generic=True):
        _lib.check(_lib.lib().mf_op_set_generic(self._h, int(generic)))
        return self

    def __call__(self, x):
        """x: numpy (host) or torch cuda tensor of the operator's element type (int8 / uint8),
        shape [..batch] + in_tail."""
        torch = _torch()
        is_np = not isinstance(x, torch.Tensor)
        tdt = torch.uint8 if self.dtype == np.uint8 else torch.int8
        if is_np and isinstance(x, np.ndarray) and x.dtype.kind in "iu" and x.dtype.itemsize == 1 \
                and x.dtype != self.dtype:
            raise TypeError("%s tensor expected" % np.dtype(self.dtype).name)
        xt = torch.as_tensor(np.ascontiguousarray(x, dtype=self.dtype)).cuda() if is_np else x.contiguous()
        if xt.dtype != tdt:
            raise TypeError("%s tensor expected" % np.dtype(self.dtype).name)
        in_elems = int(np.prod(self.in_tail))
        if xt.numel() % in_elems:
            raise ValueError("input size %d is not a multiple of %d" % (xt.numel(), in_elems))
        batch = xt.numel() // in_elems
        nt = len(self.in_tail)
        if xt.dim() >= nt and tuple(xt.shape[-nt:]) == self.in_tail:
            lead = tuple(xt.shape[:-nt])   # [..batch dims] + operator shape
        else:
            lead = (batch,)
        out = torch.empty(lead + self.out_tail, dtype=tdt, device=xt.device)
        stream = torch.cuda.current_stream(xt.device).cuda_stream
        _lib.check(_lib.lib().mf_op_run(self._h, xt.data_ptr(), batch, out.data_ptr(), stream))
        if is_np:
            return out.cpu().numpy()
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().mf_op_destroy(h)
            except Exception:  # noqa: BLE001
                pass


def _device():
    torch = _torch()
    return torch.cuda.current_device()


def prepare_fully_connected(M, weights_nk, weights_zero_point, output_scale, output_zero_point,
                            options, constants):
    dt = _elem(weights_nk)
    w = _host(weights_nk, dt)
    N, K = w.shape
    c0, c1, c2, c3 = constants
    c0, c2 = _host(np.reshape(c0, -1), np.float32), _host(np.reshape(c2, -1), np.int32)
    h = C.c_void_p()
    _lib.check(_fn("mf_fully_connected_create", dt)(
        _device(), M, K, N, _ptr(w), int(weights_zero_point), float(output_scale),
        int(output_zero_point), int(options.fused_activation), _ptr(c0), float(c1), _ptr(c2),
        int(c3), C.byref(h)))
    return PreparedOp(h, (M, K), (M, N), dt)


def fully_connected(input: Tensor2D, weights: Tensor2D, output_scale, output_zero_point,
                    options: FullyConnectedOptions, constants) -> Tensor2D:
    """microflow::ops::fully_connected (src/ops/fully_connected.rs:24-41).
    `weights.buffer` is K x N like the reference's Tensor2D<T, INPUT_COLS, WEIGHTS_COLS>."""
    w_kn = np.asarray(weights.buffer, dtype=_elem(weights.buffer))
    M = int(input.buffer.shape[-2])
    op = prepare_fully_connected(M, w_kn.T, weights.zero_point[0], output_scale[0],
                                 output_zero_point[0], options, constants)
    return Tensor2D(op(input.buffer), list(output_scale), list(output_zero_point))


def prepare_conv_2d(in_hwc, filters, filters_zero_point, input_zero_point, output_scale,
                    output_zero_point, options, constants, out_hw):
    dt = _elem(filters)
    f = _host(filters, dt)
    N, KH, KW, Cc = f.shape
    H, W, C_in = in_hwc
    if C_in != Cc:
        raise ValueError("filter channels != input channels")
    fzp = _host(np.reshape(filters_zero_point, -1), dt)
    c0, c1 = (_host(np.reshape(c, -1), np.float32) for c in constants)
    h = C.c_void_p()
    _lib.check(_fn("mf_conv_2d_create", dt)(
        _device(), H, W, Cc, N, KH, KW, _ptr(f), _ptr(fzp), fzp.size, int(input_zero_point),
        float(output_scale), int(output_zero_point), int(options.fused_activation),
        int(options.view_padding), options.strides[0], options.strides[1], out_hw[0], out_hw[1],
        _ptr(c0), _ptr(c1), c1.size, C.byref(h)))
    return PreparedOp(h, (H, W, Cc), (out_hw[0], out_hw[1], N), dt)


def conv_2d(input: Tensor4D, filters: Tensor4D, output_scale, output_zero_point,
            options: Conv2DOptions, constants, output_shape) -> Tensor4D:
    """microflow::ops::conv_2d (src/ops/conv_2d.rs:28-49)."""
    shp = tuple(input.buffer.shape)
    op = prepare_conv_2d(shp[-3:], filters.buffer, filters.zero_point, input.zero_point[0],
                         output_scale[0], output_zero_point[0], options, constants, output_shape)
    return Tensor4D(op(input.buffer), list(output_scale), list(output_zero_point))


def prepare_depthwise_conv_2d(in_hwc, weights, weights_zero_point, input_zero_point, output_scale,
                              output_zero_point, options, constants, out_hw):
    dt = _elem(weights)
    w = _host(weights, dt)
    if w.ndim == 4:
        w = w[0]
    KH, KW, WC = w.shape
    H, W, Cin = in_hwc
    wzp = _host(np.reshape(weights_zero_point, -1), dt)
    c0, c1 = (_host(np.reshape(c, -1), np.float32) for c in constants)
    h = C.c_void_p()
    _lib.check(_fn("mf_depthwise_conv_2d_create", dt)(
        _device(), H, W, Cin, KH, KW, WC, _ptr(w), _ptr(wzp), wzp.size, int(input_zero_point),
        float(output_scale), int(output_zero_point), int(options.fused_activation),
        int(options.view_padding), options.strides[0], options.strides[1], out_hw[0], out_hw[1],
        _ptr(c0), _ptr(c1), c1.size, C.byref(h)))
    return PreparedOp(h, (H, W, Cin), (out_hw[0], out_hw[1], WC), dt)


def depthwise_conv_2d(input: Tensor4D, weights: Tensor4D, output_scale, output_zero_point,
                      options: DepthwiseConv2DOptions, constants, output_shape) -> Tensor4D:
    """microflow::ops::depthwise_conv_2d (src/ops/depthwise_conv_2d.rs:28-49)."""
    shp = tuple(input.buffer.shape)
    op = prepare_depthwise_conv_2d(shp[-3:], weights.buffer, weights.zero_point,
                                   input.zero_point[0], output_scale[0], output_zero_point[0],
                                   options, constants, output_shape)
    return Tensor4D(op(input.buffer), list(output_scale), list(output_zero_point))


def prepare_average_pool_2d(in_hwc, filter_shape, output_scale, output_zero_point, options,
                            constants, out_hw, dtype=np.int8):
    H, W, Cc = in_hwc
    h = C.c_void_p()
    _lib.check(_fn("mf_average_pool_2d_create", dtype)(
        _device(), H, W, Cc, filter_shape[0], filter_shape[1], float(output_scale),
        int(output_zero_point), int(options.fused_activation), int(options.view_padding),
        options.strides[0], options.strides[1], out_hw[0], out_hw[1], float(constants[0]),
        float(constants[1]), C.byref(h)))
    return PreparedOp(h, (H, W, Cc), (out_hw[0], out_hw[1], Cc), dtype)


def average_pool_2d(input: Tensor4D, filter_shape, output_scale, output_zero_point,
                    options: AveragePool2DOptions, constants, output_shape) -> Tensor4D:
    """microflow::ops::average_pool_2d (src/ops/average_pool_2d.rs:29-45)."""
    shp = tuple(input.buffer.shape)
    op = prepare_average_pool_2d(shp[-3:], filter_shape, output_scale[0], output_zero_point[0],
                                 options, constants, output_shape, dtype=_elem(input.buffer))
    return Tensor4D(op(input.buffer), list(output_scale), list(output_zero_point))


def prepare_softmax(rows, cols, input_scale, output_scale, output_zero_point, dtype=np.int8):
    h = C.c_void_p()
    _lib.check(_fn("mf_softmax_create", dtype)(_device(), rows, cols, float(input_scale),
                                               float(output_scale), int(output_zero_point),
                                               C.byref(h)))
    return PreparedOp(h, (rows, cols), (rows, cols), dtype)


def softmax(input: Tensor2D, output_scale, output_zero_point) -> Tensor2D:
    """microflow::ops::softmax (src/ops/softmax.rs:15-19)."""
    rows, cols = input.buffer.shape[-2:]
    op = prepare_softmax(int(rows), int(cols), input.scale[0], output_scale[0], output_zero_point[0],
                         dtype=_elem(input.buffer))
    return Tensor2D(op(input.buffer), list(output_scale), list(output_zero_point))


def reshape(input, output_shape):
    """microflow::ops::reshape (src/ops/reshape.rs:3-8): the 2D<->4D conversions of
    src/tensor.rs:103-141 keep logical NHWC order, i.e. a plain reshape of row-major memory."""
    buf = input.buffer.reshape(tuple(output_shape))
    cls = Tensor2D if len(output_shape) == 2 else Tensor4D
    return cls(buf, list(input.scale), list(input.zero_point))


def quantize(x, scale, zero_point, dtype=np.int8):
    """Tensor{2D,4D}::quantize (src/tensor.rs:80-86,246-256) on the device; dtype = T."""
    torch = _torch()
    is_np = not isinstance(x, torch.Tensor)
    xt = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda() if is_np else x.contiguous()
    out = torch.empty(xt.shape, dtype=torch.uint8 if dtype == np.uint8 else torch.int8, device=xt.device)
    _lib.check(_fn("mf_quantize", dtype)(xt.device.index or 0, xt.data_ptr(), xt.numel(), float(scale),
                                      int(zero_point), out.data_ptr(),
                                      torch.cuda.current_stream(xt.device).cuda_stream))
    return out.cpu().numpy() if is_np else out


def dequantize(q, scale, zero_point):
    """Tensor{2D,4D}::dequantize (src/tensor.rs:89-92,259-262) on the device."""
    torch = _torch()
    is_np = not isinstance(q, torch.Tensor)
    dt = _elem(q)
    qt = torch.as_tensor(np.ascontiguousarray(q, dtype=dt)).cuda() if is_np else q.contiguous()
    out = torch.empty(qt.shape, dtype=torch.float32, device=qt.device)
    _lib.check(_fn("mf_dequantize", dt)(qt.device.index or 0, qt.data_ptr(), qt.numel(),
                                        float(scale), int(zero_point), out.data_ptr(),
                                        torch.cuda.current_stream(qt.device).cuda_stream))
    return out.cpu().numpy() if is_np else out
