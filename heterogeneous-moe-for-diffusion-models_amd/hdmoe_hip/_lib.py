"""ctypes binding of libhdmoe_hip.so (C ABI declared in include/hdmoe.h).

The product path has no CPU / eager fallback: if the shared library is missing
or a kernel returns an error code, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HDMOE_LIB_PATH") or os.path.join(_HERE, "libhdmoe_hip.so")   # (override: development ablation builds)

F32, BF16, F16 = 0, 1, 3                   # (F16: hdmoe_cast only -- fp16 tensors are converted at the module boundary)
MAX_GROUPS = 8

# one letter per argument:
#   p device pointer (tensor / None)      P host array of device pointers (list of tensors / None entries)
#   I host int array (list of ints)       F host float array     i int   l long   f float   u unsigned long long   s stream
# The table is derived from include/hdmoe.h itself, so the binding cannot drift from the declared C ABI.
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "hdmoe.h"))
_HOST_INT_ARRAYS = {"kh", "kw", "pt", "pl", "lens", "sb", "dims"}
_HOST_FLOAT_ARRAYS = {"group_lr", "group_wd", "src_scale", "taps"}


def _parse_header(path: str) -> dict:
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\bint\s+(hdmoe_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        if args in ("", "void"):
            sigs[name] = ""
            continue
        code = ""
        for a in args.split(","):
            a = " ".join(a.split())
            pname = a.split()[-1].lstrip("*")
            if a.startswith("HS "):
                code += "s"
            elif "* const*" in a or "*const*" in a:
                code += "P"
            elif "*" in a:
                code += "I" if pname in _HOST_INT_ARRAYS else ("F" if pname in _HOST_FLOAT_ARRAYS else "p")
            elif a.startswith("unsigned long long"):
                code += "u"
            elif a.startswith("long"):
                code += "l"
            elif a.startswith("float"):
                code += "f"
            elif a.startswith("int"):
                code += "i"
            else:
                raise RuntimeError(f"hdmoe.h: cannot map parameter '{a}' of {name}")
        sigs[name] = code
    return sigs


SIGNATURES = _parse_header(HEADER_PATH)

_CT = {"p": ctypes.c_void_p, "P": ctypes.c_void_p, "I": ctypes.c_void_p, "F": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long,
       "f": ctypes.c_float, "u": ctypes.c_ulonglong, "s": ctypes.c_void_p}

_lib = None


def lib():
    """Load the shared library once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(make -C heterogeneous-moe-for-diffusion-models_amd/csrc). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, sig in SIGNATURES.items():
            fn = getattr(_lib, name)          # AttributeError here == the .so does not export a declared symbol
            fn.restype = ctypes.c_int
            fn.argtypes = [_CT[c] for c in sig]
    return _lib


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError(f"hdmoe_hip kernels support float32 and bfloat16 activations, got {dt}")


def dtype_code_cast(dt: torch.dtype) -> int:
    """dtype codes hdmoe_cast accepts: float16 as well (module-boundary conversion of `.half()` tensors)."""
    return F16 if dt == torch.float16 else dtype_code(dt)


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("hdmoe_hip: tensors must live on the GPU (no CPU fallback in the product path)")
    return ctypes.c_void_p(t.data_ptr())


def _ptr_array(ts: Sequence):
    if ts is None:
        return None
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _int_array(v: Sequence[int]):
    return (ctypes.c_int * len(v))(*[int(a) for a in v])


_ERR = {-1: "invalid argument", -2: "unsupported dtype", -3: "kernel launch failed"}


CALL_LOG = None                                 # dev aid (tools/call_order.py): a list collects (entry point, stream id) per call


def call(name: str, *args):
    """Invoke a C-ABI entry point on the current torch stream; raises on a non-zero status."""
    if CALL_LOG is not None:
        CALL_LOG.append((name, torch.cuda.current_stream().stream_id))
    sig = SIGNATURES[name]
    fn = getattr(lib(), name)
    if not sig.endswith("s"):
        raise TypeError(f"{name} takes no stream; call it through lib() directly")
    if len(args) != len(sig) - 1:
        raise TypeError(f"{name}: expected {len(sig) - 1} arguments, got {len(args)}")
    conv, keep = [], []
    for c, a in zip(sig, args):
        if c == "p":
            conv.append(_ptr(a))
        elif c == "P":
            arr = _ptr_array(a); keep.append(arr); conv.append(ctypes.cast(arr, ctypes.c_void_p) if arr is not None else None)
        elif c == "I":
            arr = _int_array(a); keep.append(arr); conv.append(ctypes.cast(arr, ctypes.c_void_p))
        elif c == "F":
            arr = (ctypes.c_float * len(a))(*[float(v) for v in a]); keep.append(arr); conv.append(ctypes.cast(arr, ctypes.c_void_p))
        elif c == "f":
            conv.append(float(a))
        else:
            conv.append(int(a))
    conv.append(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    rc = fn(*conv)
    if rc < 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")
    return rc                                  # > 0: "not applicable, nothing launched" (entry points that document it)
