"""ctypes binding of libmicroflow_amd.so (include/microflow_amd.h).

The library is built in-tree by microflow_rs_amd/build.py (hipcc, gfx950).  If it
is missing and cannot be built, loading fails loudly -- the product has no CPU path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmicroflow_amd.so")

MF_OK, MF_ERR_INVALID_MODEL, MF_ERR_UNSUPPORTED, MF_ERR_INVALID_ARG = 0, 1, 2, 3
MF_ERR_NO_DEVICE, MF_ERR_HIP, MF_ERR_OOM = 4, 5, 6
MF_MEM_HOST, MF_MEM_DEVICE = 0, 1
MF_ELEM_I8, MF_ELEM_U8 = 0, 1
STATUS_NAMES = {0: "MF_OK", 1: "MF_ERR_INVALID_MODEL", 2: "MF_ERR_UNSUPPORTED",
                3: "MF_ERR_INVALID_ARG", 4: "MF_ERR_NO_DEVICE", 5: "MF_ERR_HIP", 6: "MF_ERR_OOM"}


class MicroflowError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status
        self.message = message


class ModelInfo(C.Structure):
    _fields_ = [("input_rank", C.c_int), ("input_shape", C.c_int * 4),
                ("output_rank", C.c_int), ("output_shape", C.c_int * 4),
                ("input_scale", C.c_float), ("output_scale", C.c_float),
                ("input_zero_point", C.c_int), ("output_zero_point", C.c_int),
                ("input_elems", C.c_size_t), ("output_elems", C.c_size_t), ("num_ops", C.c_int),
                ("element_type", C.c_int)]


class OpDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("in_rank", C.c_int), ("in_shape", C.c_int * 4),
                ("out_rank", C.c_int), ("out_shape", C.c_int * 4),
                ("KH", C.c_int), ("KW", C.c_int), ("stride_h", C.c_int), ("stride_w", C.c_int),
                ("padding", C.c_int), ("activation", C.c_int), ("n_c0", C.c_int),
                ("n_c1", C.c_int), ("in_scale", C.c_float), ("out_scale", C.c_float),
                ("in_zero_point", C.c_int), ("out_zero_point", C.c_int),
                ("out_elems", C.c_size_t), ("kernel", C.c_char_p)]


_i8p, _f32p, _i32p = C.POINTER(C.c_int8), C.POINTER(C.c_float), C.POINTER(C.c_int32)
_vp = C.c_void_p

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against
# the declarations in include/microflow_amd.h
SIGNATURES = {
    "mf_last_error": (C.c_char_p, []),
    "mf_abi_version": (C.c_int, []),
    "mf_build_info": (C.c_char_p, []),
    "mf_device_count": (C.c_int, []),
    "mf_preprocess_fully_connected": (C.c_int, [C.c_float, C.c_int8, C.c_int, _vp, C.c_int, C.c_int,
                                                C.c_float, C.c_int8, _vp, C.c_float, C.c_int32,
                                                C.c_float, _vp, _vp, _vp, _vp]),
    "mf_preprocess_conv_2d": (C.c_int, [C.c_float, C.c_int, _vp, _vp, _vp, C.c_int, _vp, C.c_int,
                                        C.c_float, _vp, _vp]),
    "mf_preprocess_depthwise_conv_2d": (C.c_int, [C.c_float, C.c_int, _vp, _vp, _vp, C.c_int, _vp,
                                                  C.c_int, C.c_float, _vp, _vp]),
    "mf_preprocess_average_pool_2d": (C.c_int, [C.c_float, C.c_int8, C.c_float, C.c_int8, _vp, _vp]),
    "mf_fully_connected_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int8,
                                            C.c_float, C.c_int8, C.c_int, _vp, C.c_float, _vp,
                                            C.c_int32, C.POINTER(_vp)]),
    "mf_conv_2d_create": (C.c_int, [C.c_int] * 7 + [_vp, _vp, C.c_int, C.c_int8, C.c_float, C.c_int8,
                                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_int, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "mf_depthwise_conv_2d_create": (C.c_int, [C.c_int] * 7 + [_vp, _vp, C.c_int, C.c_int8, C.c_float,
                                                              C.c_int8, C.c_int, C.c_int, C.c_int,
                                                              C.c_int, C.c_int, C.c_int, _vp, _vp,
                                                              C.c_int, C.POINTER(_vp)]),
    "mf_average_pool_2d_create": (C.c_int, [C.c_int] * 6 + [C.c_float, C.c_int8, C.c_int, C.c_int,
                                                            C.c_int, C.c_int, C.c_int, C.c_int,
                                                            C.c_float, C.c_float, C.POINTER(_vp)]),
    "mf_softmax_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int8,
                                    C.POINTER(_vp)]),
    "mf_op_run": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp]),
    "mf_op_input_elems": (C.c_size_t, [_vp]),
    "mf_op_output_elems": (C.c_size_t, [_vp]),
    "mf_op_kernel_name": (C.c_char_p, [_vp]),
    "mf_op_set_generic": (C.c_int, [_vp, C.c_int]),
    "mf_op_destroy": (None, [_vp]),
    "mf_quantize": (C.c_int, [C.c_int, _vp, C.c_size_t, C.c_float, C.c_int8, _vp, _vp]),
    "mf_dequantize": (C.c_int, [C.c_int, _vp, C.c_size_t, C.c_float, C.c_int8, _vp, _vp]),
    "mf_model_create": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(_vp)]),
    "mf_model_destroy": (None, [_vp]),
    "mf_model_get_info": (C.c_int, [_vp, C.POINTER(ModelInfo)]),
    "mf_model_get_op": (C.c_int, [_vp, C.c_int, C.POINTER(OpDesc)]),
    "mf_model_get_op_constants": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "mf_model_get_op_epilogue_mode": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "mf_model_prepare": (C.c_int, [_vp, C.c_int, C.c_size_t]),
    "mf_model_set_stream": (C.c_int, [_vp, _vp]),
    "mf_model_sync": (C.c_int, [_vp]),
    "mf_model_predict": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_int]),
    "mf_model_predict_quantized": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_int]),
    "mf_model_run_quantized": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_int]),
    "mf_model_run_until": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_int]),
    "mf_models_run_quantized": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, C.c_size_t, _vp]),
    "mf_models_predict": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, C.c_size_t, _vp]),
    "mf_models_predict_quantized": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, C.c_size_t, _vp]),
    "mf_model_set_generic": (C.c_int, [_vp, C.c_int]),
    "mf_model_set_fusion": (C.c_int, [_vp, C.c_int]),
    "mf_model_set_autotune": (C.c_int, [_vp, C.c_int]),
    "mf_model_set_graph": (C.c_int, [_vp, C.c_int]),
    "mf_model_graph_launches": (C.c_ulonglong, [_vp]),
    "mf_synth_i8": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_size_t, _vp, _vp]),
    "mf_checksum_i8": (C.c_int, [C.c_int, _vp, C.c_size_t, C.POINTER(C.c_uint64), _vp]),
    "mf_preprocess_softmax": (C.c_int, [C.c_float, C.c_int, _vp]),
    "mf_verify_quant_div": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "mf_selftest_rounding": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "mf_selftest_requant": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                      C.POINTER(C.c_uint64)]),
    "mf_fma_epilogue_search": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.POINTER(C.c_float),
                                         C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]),
    "mf_fma_epilogue_check_host": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_longlong, C.c_longlong, C.c_float, C.c_float,
                                             C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_uint64)]),
    "mf_selftest_fma_epilogue": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_int, C.c_longlong, C.c_longlong, C.c_float,
                                           C.c_float, C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_uint64)]),
    "mf_selftest_cvt_pk": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "mf_model_time_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_int, C.c_int,
                                       C.POINTER(C.c_float), _vp]),
}

# T = u8 instantiations: the i8 signature with every int8_t value parameter as uint8_t
for _n in ("mf_preprocess_fully_connected", "mf_preprocess_average_pool_2d",
           "mf_fully_connected_create", "mf_conv_2d_create", "mf_depthwise_conv_2d_create",
           "mf_average_pool_2d_create", "mf_softmax_create", "mf_quantize", "mf_dequantize"):
    _r, _a = SIGNATURES[_n]
    SIGNATURES[_n + "_u8"] = (_r, [C.c_uint8 if t is C.c_int8 else t for t in _a])

_lib = None


def lib_path():
    return _LIB


def lib():
    """Load (building first if the .so is missing) and return the ctypes library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        try:
            from . import build as _build
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise ImportError(
                "libmicroflow_amd.so is missing and could not be built (%s). Run "
                "`python microflow_rs_amd/build.py`; this package has no CPU fallback." % e)
    # PyTorch-ROCm bundles its own libamdhip64; if it is going to be used in this process (device
    # memory, streams) it must be loaded FIRST so that the library binds to the same HIP runtime
    # -- two runtimes in one process see no devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = ABI / header drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    if L.mf_abi_version() != 3:
        raise ImportError("libmicroflow_amd.so ABI version mismatch")
    import re
    info = (L.mf_build_info() or b"").decode()
    if re.search(r"-DMF_\w*(KO|DIAG)\w*(=(?!0(\s|$))|\s|$)", info) and not os.environ.get("MF_ALLOW_DIAG_BUILD"):
        raise ImportError("libmicroflow_amd.so was built with knock-out / diagnostic switches (%s): its kernels are wrong on "
                          "purpose.  Rebuild without MF_EXTRA_HIPCC_FLAGS (python microflow_rs_amd/build.py --force), or set "
                          "MF_ALLOW_DIAG_BUILD=1 for a profiling script." % info)
    _lib = L
    return L


def check(status):
    if status != MF_OK:
        msg = lib().mf_last_error()
        raise MicroflowError(status, msg.decode() if msg else "")
    return status
