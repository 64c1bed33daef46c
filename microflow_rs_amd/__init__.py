"""MI355X-native implementation of MicroFlow's quantized operator hot path behind the C ABI of
include/microflow_amd.h.  Python is host plumbing only (ctypes + torch for device memory and streams); there is no
CPU fallback: every compute entry point raises if the library or a GPU is missing."""
from ._lib import MicroflowError, lib, lib_path  # noqa: F401
from .model import Model, model, run_sharded  # noqa: F401
from . import ops  # noqa: F401
from .tensor import (FusedActivation, Tensor2D, Tensor4D, TensorViewPadding)  # noqa: F401

__all__ = ["model", "Model", "run_sharded", "ops", "Tensor2D", "Tensor4D", "FusedActivation",
           "TensorViewPadding", "MicroflowError", "lib", "lib_path"]
