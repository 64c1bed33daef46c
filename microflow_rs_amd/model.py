"""The `#[model("x.tflite")]` surface (microflow-macros/src/lib.rs:185-203) on top of
the C ABI: predict / predict_quantized, plus the batched forms the MI355X build adds.

numpy arrays are treated as HOST buffers (the library stages them through HBM);
torch CUDA tensors are consumed and produced in place on the device.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

OP_NAMES = {1: "average_pool_2d", 3: "conv_2d", 4: "depthwise_conv_2d", 9: "fully_connected",
            22: "reshape", 25: "softmax"}


class Model:
    def __init__(self, path_or_bytes, device=None, max_batch=0, autotune=False):
        if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            data = bytes(path_or_bytes)
        else:
            if not os.path.exists(path_or_bytes):  # lib.rs:50-55
                raise FileNotFoundError("couldn't find '%s', please provide a valid path"
                                        % path_or_bytes)
            with open(path_or_bytes, "rb") as f:
                data = f.read()
        L = _lib.lib()
        h = C.c_void_p()
        _lib.check(L.mf_model_create(data, len(data), C.byref(h)))
        self._h = h
        info = _lib.ModelInfo()
        _lib.check(L.mf_model_get_info(h, C.byref(info)))
        self.input_shape = tuple(info.input_shape[: info.input_rank])
        self.output_shape = tuple(info.output_shape[: info.output_rank])
        self.input_scale, self.input_zero_point = np.float32(info.input_scale), info.input_zero_point
        self.output_scale, self.output_zero_point = np.float32(info.output_scale), info.output_zero_point
        self.input_elems, self.output_elems = info.input_elems, info.output_elems
        self.num_ops = info.num_ops
        # element type T of the model's quantized tensors (i8 or u8, microflow src/quantize.rs:32-53)
        self.dtype = np.uint8 if info.element_type == _lib.MF_ELEM_U8 else np.int8
        self._device = device
        self._prepared = False
        if autotune:
            self.set_autotune(True)
        if max_batch:
            self.prepare(max_batch)

    # ---- introspection (no GPU needed) -------------------------------------
    def op(self, i):
        d = _lib.OpDesc()
        _lib.check(_lib.lib().mf_model_get_op(self._h, i, C.byref(d)))
        return dict(kind=d.kind, name=OP_NAMES.get(d.kind, "?"),
                    in_shape=tuple(d.in_shape[: d.in_rank]), out_shape=tuple(d.out_shape[: d.out_rank]),
                    KH=d.KH, KW=d.KW, sh=d.stride_h, sw=d.stride_w, pad=d.padding, act=d.activation,
                    n_c0=d.n_c0, n_c1=d.n_c1, in_scale=np.float32(d.in_scale),
                    out_scale=np.float32(d.out_scale), in_zp=d.in_zero_point, out_zp=d.out_zero_point,
                    out_elems=d.out_elems, kernel=(d.kernel or b"").decode())

    def op_epilogue_mode(self, i):
        """requantisation form (k_common.hpp: 0, 1, 2) of the launch that starts at operator i of a prepared model; -1: none"""
        mode = C.c_int(-1)
        _lib.check(_lib.lib().mf_model_get_op_epilogue_mode(self._h, int(i), C.byref(mode)))
        return mode.value

    @property
    def ops(self):
        return [self.op(i) for i in range(self.num_ops)]

    def op_constants(self, i):
        d = self.op(i)
        c0 = np.zeros(max(d["n_c0"], 1), np.float32)
        c1 = np.zeros(max(d["n_c1"], 1), np.float32)
        c2 = np.zeros(max(d["n_c0"], 1), np.int32)
        c3 = C.c_int32(0)
        _lib.check(_lib.lib().mf_model_get_op_constants(
            self._h, i, c0.ctypes.data_as(C.c_void_p), c1.ctypes.data_as(C.c_void_p),
            c2.ctypes.data_as(C.c_void_p), C.byref(c3)))
        return c0, c1, c2, c3.value

    # ---- device -------------------------------------------------------------
    def prepare(self, max_batch=1, device=None):
        if device is None:
            device = self._device
        if device is None:
            import torch
            if not torch.cuda.is_available():
                raise _lib.MicroflowError(_lib.MF_ERR_NO_DEVICE,
                                          "no GPU visible: microflow has no CPU fallback")
            device = torch.cuda.current_device()
        self._device = int(device)
        _lib.check(_lib.lib().mf_model_prepare(self._h, self._device, int(max_batch)))
        self._prepared = True
        return self

    def set_generic(self, generic=True):
        _lib.check(_lib.lib().mf_model_set_generic(self._h, int(generic)))
        return self

    def set_autotune(self, enabled=True):
        """before prepare(): time the fused-chain candidates of run-time-geometry models on the device instead of planning them
        from the cost model (mf_model_set_autotune)"""
        _lib.check(_lib.lib().mf_model_set_autotune(self._h, int(bool(enabled))))

    def set_fusion(self, enabled=True):
        _lib.check(_lib.lib().mf_model_set_fusion(self._h, int(enabled)))
        return self

    def set_graph(self, enabled=True):
        """hipGraph replay of device-resident calls that repeat the same buffers (small batches)."""
        _lib.check(_lib.lib().mf_model_set_graph(self._h, int(enabled)))
        return self

    @property
    def graph_launches(self):
        return int(_lib.lib().mf_model_graph_launches(self._h))

    def sync(self):
        _lib.check(_lib.lib().mf_model_sync(self._h))

    def _io(self, x, np_dtype, elems, out_elems, out_torch_dtype, out_np_dtype, trailing, out=None):
        """Normalise an input to (pointer, batch, mem, output buffer, finish())."""
        import torch
        if isinstance(x, torch.Tensor):
            if not x.is_cuda:
                x = x.numpy()
        if np_dtype != np.float32:  # quantized input: the other 1-byte integer type is a caller bug
            got = x.dtype if isinstance(x, np.ndarray) else getattr(x, "dtype", None)
            other = {np.int8: (np.dtype(np.uint8), torch.uint8),
                     np.uint8: (np.dtype(np.int8), torch.int8)}[np_dtype]
            if got is not None and got in other:
                raise TypeError("model element type is %s, got %s" % (np.dtype(np_dtype).name, got))
        if isinstance(x, np.ndarray) or not isinstance(x, torch.Tensor):
            a = np.ascontiguousarray(x, dtype=np_dtype)
            if a.size % elems:
                raise ValueError("input has %d elements, expected a multiple of %d" % (a.size, elems))
            batch = a.size // elems
            single = a.size == elems and a.ndim <= len(self.input_shape)
            out = np.empty((batch, out_elems), out_np_dtype)
            fin = (lambda: out.reshape(trailing) if single else out.reshape((batch,) + trailing))
            return a.ctypes.data_as(C.c_void_p), batch, _lib.MF_MEM_HOST, out.ctypes.data_as(C.c_void_p), fin, a
        # a device tensor is passed by pointer: its dtype and device must be exactly what the kernels read
        want = torch.float32 if np_dtype == np.float32 else self._tdtype()
        if x.dtype != want:
            raise TypeError("device input must be a %s tensor, got %s (no implicit cast of device memory)"
                            % (want, x.dtype))
        if self._device is None and not self._prepared:
            self._device = x.device.index       # an unprepared model follows its first device input
        if self._device is not None and x.device.index != self._device:
            raise ValueError("input lives on cuda:%s but the model belongs to cuda:%s" % (x.device.index, self._device))
        xt = x.contiguous()
        if xt.numel() % elems:
            raise ValueError("input has %d elements, expected a multiple of %d" % (xt.numel(), elems))
        batch = xt.numel() // elems
        single = xt.numel() == elems and xt.dim() <= len(self.input_shape)
        if out is None:
            out = torch.empty((batch, out_elems), dtype=out_torch_dtype, device=xt.device)
        elif (not out.is_cuda or out.dtype != out_torch_dtype or out.numel() != batch * out_elems
              or not out.is_contiguous()):
            raise ValueError("out= must be a contiguous cuda tensor of %d %s values"
                             % (batch * out_elems, out_torch_dtype))
        _lib.check(_lib.lib().mf_model_set_stream(self._h, torch.cuda.current_stream(xt.device).cuda_stream))
        fin = (lambda: out.reshape(trailing) if single else out.reshape((batch,) + trailing))
        return xt.data_ptr(), batch, _lib.MF_MEM_DEVICE, out.data_ptr(), fin, xt

    def _tdtype(self):
        import torch
        return torch.uint8 if self.dtype == np.uint8 else torch.int8

    def _ensure(self, batch):
        if not self._prepared:
            self.prepare(batch)

    # ---- the #[model] methods -------------------------------------------------
    def predict(self, x, out=None):
        """M::predict (lib.rs:188-191).  x: f32 [input_shape] or [B, *input_shape].
        out= (cuda inputs only): preallocated result tensor; feeding the same x / out buffers
        again is what lets set_graph() replay the launch sequence."""
        import torch
        p, batch, mem, o, fin, keep = self._io(x, np.float32, self.input_elems, self.output_elems,
                                               torch.float32, np.float32, self.output_shape, out)
        self._ensure(batch)
        _lib.check(_lib.lib().mf_model_predict(self._h, p, batch, o, mem))
        return fin()

    def predict_quantized(self, xq, out=None):
        """M::predict_quantized (lib.rs:193-196).  xq: the model's element type (int8 / uint8)."""
        import torch
        p, batch, mem, o, fin, keep = self._io(xq, self.dtype, self.input_elems, self.output_elems,
                                               torch.float32, np.float32, self.output_shape, out)
        self._ensure(batch)
        _lib.check(_lib.lib().mf_model_predict_quantized(self._h, p, batch, o, mem))
        return fin()

    predict_batch = predict                      # new surface: B independent inferences
    predict_quantized_batch = predict_quantized

    def run_quantized(self, xq, out=None):
        """predict_inner (lib.rs:198-201): int8 in, int8 out (before dequantize)."""
        import torch
        p, batch, mem, o, fin, keep = self._io(xq, self.dtype, self.input_elems, self.output_elems,
                                               self._tdtype(), self.dtype, self.output_shape, out)
        self._ensure(batch)
        _lib.check(_lib.lib().mf_model_run_quantized(self._h, p, batch, o, mem))
        return fin()

    def run_until(self, xq, last_op):
        """int8 output of op `last_op` for the whole batch (per-layer parity localisation)."""
        import torch
        d = self.op(last_op)
        p, batch, mem, o, fin, keep = self._io(xq, self.dtype, self.input_elems, d["out_elems"],
                                               self._tdtype(), self.dtype, d["out_shape"])
        self._ensure(batch)
        _lib.check(_lib.lib().mf_model_run_until(self._h, p, batch, int(last_op), o, mem))
        return fin()

    def time_device(self, d_in, d_out, batch, warmup=3, iters=20, per_op=True):
        """HIP-event timing of `iters` passes over a device-resident batch (torch tensors)."""
        import torch
        self._ensure(batch)
        _lib.check(_lib.lib().mf_model_set_stream(self._h, torch.cuda.current_stream().cuda_stream))
        avg = C.c_float(0)
        per = (C.c_float * self.num_ops)() if per_op else None
        _lib.check(_lib.lib().mf_model_time_device(self._h, d_in.data_ptr(), int(batch),
                                                   d_out.data_ptr(), int(warmup), int(iters),
                                                   C.byref(avg), per))
        return avg.value, (list(per) if per_op else None)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().mf_model_destroy(h)
            except Exception:  # noqa: BLE001
                pass


def model(path, device=None):
    """`#[model("path.tflite")] struct M;` -> `M = model("path.tflite")`; then
    `M.predict(x)` / `M.predict_quantized(xq)` like the generated associated functions."""
    return Model(path, device=device)


def run_sharded(models, x, entry="run_quantized"):
    """One host batch over several prepared replicas of a model -- normally one per GPU of the node
    (`[Model(path, device=d, max_batch=B) for d in range(n_gpus)]`) -- from a single process:
    mf_models_{run_quantized,predict,predict_quantized}.  x: numpy, [B, *input_shape]."""
    m0 = models[0]
    kinds = {"run_quantized": (m0.dtype, m0.dtype, "mf_models_run_quantized"),
             "predict": (np.float32, np.float32, "mf_models_predict"),
             "predict_quantized": (m0.dtype, np.float32, "mf_models_predict_quantized")}
    in_dt, out_dt, fn = kinds[entry]
    a = np.ascontiguousarray(x, dtype=in_dt)
    if a.size % m0.input_elems:
        raise ValueError("input has %d elements, expected a multiple of %d" % (a.size, m0.input_elems))
    batch = a.size // m0.input_elems
    for m in models:
        m._ensure(max(1, -(-batch // len(models))))
    out = np.empty((batch, m0.output_elems), out_dt)
    handles = (C.c_void_p * len(models))(*[m._h for m in models])
    _lib.check(getattr(_lib.lib(), fn)(handles, len(models), a.ctypes.data_as(C.c_void_p), batch,
                                       out.ctypes.data_as(C.c_void_p)))
    return out.reshape((batch,) + tuple(m0.output_shape))


def synth_i8(seed, first_byte, n, device=None):
    """Counter-based synthetic int8 stream generated directly in HBM (mf_synth_i8)."""
    import torch
    dev = torch.cuda.current_device() if device is None else device
    out = torch.empty(int(n), dtype=torch.int8, device="cuda:%d" % dev)
    _lib.check(_lib.lib().mf_synth_i8(dev, int(seed), int(first_byte), int(n), out.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream))
    return out


def checksum_i8(t):
    """Position-sensitive 64-bit checksum of an int8 CUDA tensor (mf_checksum_i8)."""
    import torch
    t = t.contiguous()
    res = C.c_uint64(0)
    _lib.check(_lib.lib().mf_checksum_i8(t.device.index or 0, t.data_ptr(), t.numel(), C.byref(res),
                                         torch.cuda.current_stream(t.device).cuda_stream))
    return res.value
