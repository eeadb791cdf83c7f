"""hdmoe_hip: MI355X (gfx950) kernels + autograd bindings for the HDMOEM denoising hot path.

    import hdmoe_hip
    hdmoe_hip.set_compute_dtype(torch.bfloat16)   # experts / attention / gate in bf16, stem + routers stay fp32

There is no CPU fallback: importing is cheap, but the first op call loads libhdmoe_hip.so and raises if it is
missing or the tensors are not on a GPU.
"""
import torch

from . import _lib, ops                                    # noqa: F401
from ._lib import LIB_PATH, lib                            # noqa: F401
from .ops import manual_seed                               # noqa: F401
from .bank import invalidate_weights                       # noqa: F401

_policy = {"compute_dtype": torch.float32}


def set_compute_dtype(dtype: torch.dtype) -> None:
    """Arithmetic type of the expert banks, attention and output gate inside HDMOEM.
    The stem conv, both router trunks, all statistics and the (B,F) embedding path are always fp32
    (router top-k indices must match the fp32 reference)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("compute dtype must be float32 or bfloat16")
    _policy["compute_dtype"] = dtype


def compute_dtype() -> torch.dtype:
    return _policy["compute_dtype"]
