"""microflow_rs_amd -- MI355X-native implementation of MicroFlow's quantized
operator hot path (FullyConnected / Conv2D / DepthwiseConv2D / AveragePool2D +
activation / quantize), behind the C ABI of include/microflow_amd.h.

Python here is host plumbing only (ctypes binding + torch for device memory and
streams); the arithmetic runs in hand-written HIP kernels inside
libmicroflow_amd.so.  There is NO CPU fallback: importing works anywhere, but
every compute entry point raises if the library or a GPU is missing.

    from microflow_rs_amd import model
    PersonDetect = model("models/person_detect.tflite")     # ~ #[model("...")] struct PersonDetect;
    y = PersonDetect.predict(x)                              # ~ PersonDetect::predict(x)
"""
from ._lib import MicroflowError, lib, lib_path  # noqa: F401
from .model import Model, model, run_sharded  # noqa: F401
from . import ops  # noqa: F401
from .tensor import (FusedActivation, Tensor2D, Tensor4D, TensorViewPadding)  # noqa: F401

__all__ = ["model", "Model", "run_sharded", "ops", "Tensor2D", "Tensor4D", "FusedActivation",
           "TensorViewPadding", "MicroflowError", "lib", "lib_path"]
