"""eventpretrain_amd: MI355X-native (gfx950) implementation of the BIT-Vision/EventPretrain pre-training hot path.

Python here is host glue mirroring the reference's module layout (model/, trainer/, utils/, dataset/); all arithmetic
runs in hand-written HIP kernels (csrc/ -> libevtpretrain.so, C-ABI in include/evtpretrain.h). There is no CPU or
PyTorch fallback: without the library or a HIP device, calls raise."""
from ._lib import EvpError, build_library, load  # noqa: F401
from .ops import get_compute_dtype, set_compute_dtype  # noqa: F401

__version__ = "0.1.0"
