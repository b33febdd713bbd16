"""ctypes binding of the C-ABI declared in include/livecc_amd.h (the product's only native entry point).

The library is built in-tree by `livecc_amd/build.py` (hipcc, gfx950) as `livecc_amd/_C/liblivecc_amd.so`.
There is no CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List

_HERE = os.path.dirname(os.path.abspath(__file__))
# LCC_LIB_PATH: A/B runs of two builds of the library on one box (tools/); the product always loads the in-tree build
LIB_PATH = os.environ.get("LCC_LIB_PATH") or os.path.join(_HERE, "_C", "liblivecc_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "livecc_amd.h")

_lib = None


class LccError(RuntimeError):
    pass


class KvLayout(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("n_kv_heads", C.c_int), ("lmax", C.c_int), ("head_dim", C.c_int)]


class ModelConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("vocab_size", "hidden_size", "intermediate_size", "n_layers", "n_q_heads",
                                       "n_kv_heads", "head_dim")] + [("rms_eps", C.c_float)] + \
               [(n, C.c_int) for n in ("mrope_sec_t", "mrope_sec_h", "mrope_sec_w", "vit_depth", "vit_embed",
                                       "vit_heads", "vit_mlp", "patch_dim", "merge", "llm_fp8")]


class EngineLimits(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("max_slots", "max_kv_len", "max_new_rows", "max_patches", "max_history")]


class Clip(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("pixel_values", C.c_void_p), ("layout", C.c_int), ("T", C.c_int),
                ("H", C.c_int), ("W", C.c_int)]


class Sampling(C.Structure):
    _fields_ = [("repetition_penalty", C.c_float), ("thr_token", C.c_int), ("use_thr", C.c_int),
                ("thr_base", C.c_float), ("thr_step", C.c_float), ("eos_token", C.c_int), ("suppress_eos", C.c_int),
                ("scores_out", C.c_void_p), ("logits_out", C.c_void_p), ("eos_token2", C.c_int), ("do_sample", C.c_int),
                ("temperature", C.c_float), ("top_k", C.c_int), ("top_p", C.c_float), ("seed", C.c_uint64)]


def declared_symbols() -> List[str]:
    """Every function name declared in include/livecc_amd.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lcc_[a-z0-9_]+)\s*\(", src)))


def lib_available() -> bool:
    return os.path.exists(LIB_PATH)


def load():
    """Load the shared library (no GPU needed to load; compute calls need one)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LccError(f"{LIB_PATH} not found: build the HIP extension first "
                       f"(python -c 'import __graft_entry__ as g; g.build()' or python -m livecc_amd.build)")
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so; if this library were loaded first it
    # would pull /opt/rocm's copy and torch would then load a second runtime (kernels launched through one runtime on
    # memory owned by the other fail with "no ROCm-capable device is detected").  Importing torch first makes the
    # dynamic loader resolve our libamdhip64.so.N dependency to the copy that is already mapped.
    import torch  # noqa: F401
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        try:
            C.CDLL(tl, mode=C.RTLD_GLOBAL)
        except OSError:
            pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32, i64, sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t
    sig: Dict[str, tuple] = {
        "lcc_last_error": (C.c_char_p, []),
        "lcc_version": (C.c_char_p, []),
        "lcc_device_info": (i32, [C.POINTER(i32), C.POINTER(sz), C.c_char_p, i32]),
        "lcc_gemm_bf16": (i32, [vp, i32, vp, i32, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
        "lcc_debug_set_gemv_variant": (i32, [i32]),
        "lcc_debug_set_gemm_variant": (i32, [i32]),
        "lcc_debug_set_attn_variant": (i32, [i32]),
        "lcc_debug_set_fused_tails": (i32, [i32]),
        "lcc_debug_set_decode_path": (i32, [i32]),
        "lcc_debug_set_decode_chain": (i32, [i32]),
        "lcc_debug_set_resid_waves": (i32, [i32]),
        "lcc_debug_set_skinny_rows": (i32, [i32]),
        "lcc_debug_set_vit_fused_qkv": (i32, [i32]),
        "lcc_dgemv_down_qkv": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, KvLayout, i32, vp, i32, i32, i32, vp, C.c_uint32,
                                     vp, vp]),
        "lcc_gemv_num_splits": (i32, [i32, i32]),
        "lcc_debug_mfma_probe": (i32, [vp, vp, vp, vp]),
        "lcc_patchify_norm_u8": (i32, [vp, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, i32, vp]),
        "lcc_cast_f32_bf16": (i32, [vp, vp, i64, vp]),
        "lcc_layernorm_bf16": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
        "lcc_rmsnorm_bf16": (i32, [vp, vp, vp, i32, i32, f32, vp]),
        "lcc_add_rmsnorm_bf16": (i32, [vp, vp, vp, i32, vp, vp, i32, i32, f32, vp]),
        "lcc_swiglu_bf16": (i32, [vp, vp, vp, i64, vp]),
        "lcc_vit_rope_vt_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
        "lcc_attn_vit_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, vp]),
        "lcc_attn_vit32_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "lcc_mrope_table": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
        "lcc_rope_kv_append_bf16": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, KvLayout, i32, vp, i32, i32, vp]),
        "lcc_attn_prefill_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, KvLayout, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
        "lcc_resize_bicubic_aa_u8": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp]),
        "lcc_gemm_w8_bf16": (i32, [vp, i32, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
        "lcc_attn_decode_bf16": (i32, [vp, vp, vp, vp, vp, KvLayout, i32, i32, i32, i32, vp, vp, vp]),
        "lcc_attn_decode_fused_bf16": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, KvLayout, i32, vp, i32, i32, i32, vp, vp, vp, vp]),
        "lcc_debug_set_fused_attn": (i32, [i32]),
        "lcc_debug_launch_counts": (i32, [vp, i32, i32]),
        "lcc_debug_gemm_plan": (i32, [i32, i32, i32, i32, i32, i32, vp, vp]),
        "lcc_debug_attn_tile_rows": (i32, [i32, i32]),
        "lcc_debug_attn_plan": (i32, [vp, i32, i32, i32, i32, i32, vp, vp]),
        "lcc_debug_bench_grid_barrier": (i32, [i32, i32, i32, vp, C.c_size_t, vp, vp, vp]),
        "lcc_debug_bench_attn_decode": (i32, [i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, KvLayout, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp,
                                            C.POINTER(f32), vp]),
        "lcc_embed_gather_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
        "lcc_decode_step_begin": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp]),
        "lcc_dgemv_norm_linear": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, vp]),
        "lcc_dgemv_resid": (i32, [vp, vp, vp, i32, vp, vp, i32, i32, i32, vp]),
        "lcc_dgemv_qkv_rope": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, KvLayout, i32, vp, i32, i32, i32, vp]),
        "lcc_seen_set": (i32, [vp, i32, vp, vp, i32, vp]),
        "lcc_sample_greedy": (i32, [vp, i32, i32, i32, vp, i32, vp, f32, i32, i32, f32, i32, i32, i32, vp, vp, vp, i32, vp,
                                    vp, vp, vp]),
        "lcc_sample_topk_topp": (i32, [vp, i32, i32, i32, vp, i32, vp, f32, i32, i32, f32, i32, i32, i32, vp, vp, vp, i32, vp,
                                       vp, f32, i32, f32, C.c_uint64, vp, vp]),
        "lcc_engine_create": (vp, [C.POINTER(ModelConfig), C.POINTER(EngineLimits)]),
        "lcc_engine_destroy": (None, [vp]),
        "lcc_engine_workspace_bytes": (sz, [vp]),
        "lcc_engine_state_bytes": (sz, [vp]),
        "lcc_engine_kv_bytes_per_slot": (sz, [vp]),
        "lcc_engine_meta_bytes": (sz, [vp]),
        "lcc_engine_bind_buffers": (i32, [vp, vp, sz, vp, sz, vp, vp, sz]),
        "lcc_engine_bind_kv": (i32, [vp, i32, vp, sz]),
        "lcc_engine_vit_workspace_bytes": (sz, [vp]),
        "lcc_engine_vit_meta_bytes": (sz, [vp]),
        "lcc_engine_bind_vit_buffers": (i32, [vp, vp, sz, vp, vp, sz]),
        "lcc_engine_set_vit_grid_cap": (i32, [vp, i32]),
        "lcc_engine_set_weight": (i32, [vp, C.c_char_p, vp, i64]),
        "lcc_engine_weights_ready": (i32, [vp, C.c_char_p, i32]),
        "lcc_engine_profile": (i32, [vp, i32, i32]),
        "lcc_engine_profile_read": (i32, [vp, vp, i32, C.POINTER(i32)]),
        "lcc_engine_profile_read_steps": (i32, [vp, vp, i32, C.POINTER(i32)]),
        "lcc_engine_profile_read_step_index": (i32, [vp, vp, i32, C.POINTER(i32)]),
        "lcc_debug_set_llm_taps": (i32, [vp, vp, vp, i32]),
        "lcc_debug_set_vit_taps": (i32, [vp, vp, vp, i32]),
        "lcc_debug_set_forced_tokens": (i32, [vp, vp, i32, i32]),      # engine, int32 device tokens [steps][streams], steps, streams
        "lcc_slot_reset": (i32, [vp, i32, vp]),
        "lcc_slot_set_length": (i32, [vp, i32, i32, i32, vp]),
        "lcc_slot_get_length": (i32, [vp, i32, C.POINTER(i32), C.POINTER(i32)]),
        "lcc_vit_encode": (i32, [vp, i32, C.POINTER(Clip), C.POINTER(f32), C.POINTER(f32), vp, vp, vp, vp]),
        "lcc_llm_prefill": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, C.POINTER(Sampling), vp]),
        "lcc_llm_decode": (i32, [vp, i32, vp, i32, i32, C.POINTER(Sampling), vp]),
        "lcc_slot_read_tokens": (i32, [vp, i32, vp, i32, C.POINTER(i32), vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib._lcc_signatures = sig
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().lcc_last_error().decode(errors="replace")
        raise LccError(f"{what or 'livecc_amd'} failed ({code}): {msg}")
