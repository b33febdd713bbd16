"""`torch.ops.livecc_amd.*`: the operator-level entry points registered with PyTorch's dispatcher (csrc/torch_ops.cpp, TORCH_LIBRARY).

The C-ABI (include/livecc_amd.h) stays the source of truth -- the registered ops forward to the same `lcc_*` symbols on
`c10::hip::getCurrentHIPStream()` with `TORCH_CHECK` validation -- but unlike ctypes-on-`data_ptr()` they are visible to the
dispatcher: profiler ranges, schema-checked arguments, mutation annotations, TorchScript / export.  `livecc_amd.ops` routes
`rmsnorm / layernorm / swiglu / linear / rope_kv_append / attn_prefill / attn_decode / patchify_norm / sample_greedy` and
`livecc_amd.resize.resize_bicubic_aa` through them when the library is present (LCC_TORCH_OPS=0: ctypes only), so the HF operator
plugins (`plugin.py`) run through `torch.ops`.  No CPU kernels are registered: a CPU tensor raises NotImplementedError.
"""
from __future__ import annotations

import os
from collections import Counter

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_C", "liblivecc_torch_ops.so")
OP_NAMES = ("rmsnorm", "layernorm", "swiglu", "linear", "rope_kv_append", "attn_prefill", "attn_decode", "patchify_norm",
            "resize_bicubic_aa", "sample_greedy")
CALLS: Counter = Counter()          # calls routed through torch.ops per op name (tests assert that the dispatcher path really ran)
_ns = None
_tried = False


def available() -> bool:
    return os.path.exists(LIB_PATH)


def load():
    """Load the registration library (idempotent) and return the `torch.ops.livecc_amd` namespace; raises when it is not built."""
    global _ns, _tried
    if _ns is not None:
        return _ns
    import torch
    from . import _lib
    _lib.load()          # the C-ABI library first (one HIP runtime per process: see _lib.load)
    if not available():
        raise _lib.LccError(f"{LIB_PATH} not found: build it first (python -m livecc_amd.build)")
    torch.ops.load_library(LIB_PATH)
    _ns = torch.ops.livecc_amd
    return _ns


def try_load():
    """The namespace, or None when the library is not built or LCC_TORCH_OPS=0 (callers then use the ctypes binding)."""
    global _tried
    if _ns is not None:
        return _ns
    if _tried or os.environ.get("LCC_TORCH_OPS", "1") == "0":
        return None
    _tried = True
    try:
        return load()
    except Exception:          # noqa: BLE001 -- not built / not loadable: ctypes remains the binding
        return None


_WRAPPED: dict = {}


def op(name: str):
    """The registered op `name` when the dispatcher path is on, else None; counts the call.  Errors keep the binding's contract: a failed
    TORCH_CHECK / a non-zero `lcc_*` status surfaces as `LccError` (a RuntimeError), exactly as through ctypes; a CPU tensor is refused
    by the dispatcher itself (NotImplementedError: no CPU kernel is registered)."""
    ns = try_load()
    if ns is None:
        return None
    CALLS[name] += 1
    fn = _WRAPPED.get(name)
    if fn is None:
        from . import _lib
        raw = getattr(ns, name)

        def fn(*args, _raw=raw):
            try:
                return _raw(*args)
            except NotImplementedError:
                raise
            except RuntimeError as e:
                raise _lib.LccError(str(e).split("\nException raised from")[0]) from None
        _WRAPPED[name] = fn
    return fn
