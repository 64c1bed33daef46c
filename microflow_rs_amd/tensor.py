"""Host-side mirrors of the reference's tensor types (src/tensor.rs:27-47,
src/activation.rs:6-13, src/tensor.rs:9-15) for the Python test/host layer.

Layout contract (include/microflow_amd.h): 2-D = [rows][cols] row-major, 4-D =
[batch][rows][cols][chans] (NHWC).  `buffer` is a numpy int8 array (host) or a
torch int8 CUDA tensor (HBM); the ops accept either and return the same kind.
"""
import enum
from dataclasses import dataclass, field
from typing import Any, List


class FusedActivation(enum.IntEnum):      # src/activation.rs:6-13 (values = TFLite's)
    NONE = 0
    RELU = 1
    RELU6 = 3


class TensorViewPadding(enum.IntEnum):    # src/tensor.rs:9-15
    SAME = 0
    VALID = 1


@dataclass
class Tensor2D:                           # src/tensor.rs:27-31
    buffer: Any
    scale: List[float] = field(default_factory=lambda: [1.0])
    zero_point: List[int] = field(default_factory=lambda: [0])


@dataclass
class Tensor4D:                           # src/tensor.rs:37-47
    buffer: Any
    scale: List[float] = field(default_factory=lambda: [1.0])
    zero_point: List[int] = field(default_factory=lambda: [0])
