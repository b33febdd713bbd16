"""Operator-level Python wrappers over the C-ABI (torch tensors in, pointers out).  Used by the HF plugin
(`plugin.py`) and by the per-kernel parity tests.  Every wrapper validates dtype/device/contiguity and raises;
nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, torch_ops as _T

EPI_NONE, EPI_QUICK_GELU, EPI_GELU_ERF, EPI_RESIDUAL, EPI_SWIGLU = range(5)


def _st(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(t: Optional[torch.Tensor], dtype, name: str) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.LccError(f"{name}: expected a GPU tensor (livecc_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear weight [N,K] -> MFMA-fragment order [N/16][ceil(K/32)][4][16][8] (flat), K zero-padded to 32.
    Element (n,k) lands at ((n//16 * K32 + k//32) * 4 + (k%32)//8) * 128 + (n%16)*8 + k%8 (include/livecc_amd.h)."""
    N, K = w.shape
    assert N % 16 == 0
    K32 = (K + 31) // 32
    if K32 * 32 != K:
        w = torch.nn.functional.pad(w, (0, K32 * 32 - K))
    return w.view(N // 16, 16, K32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
           residual: Optional[torch.Tensor] = None, packed_shape: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """y = epilogue(x @ w.T + bias); x [M,K] bf16; w [N,K] bf16 row-major, or a packed weight (pack_weight) with
    packed_shape=(N,K)."""
    M, K = x.shape
    N = w.shape[0] if packed_shape is None else packed_shape[0]
    assert (w.shape[1] if packed_shape is None else packed_shape[1]) == K
    _chk(x, torch.bfloat16, "x"), _chk(w, torch.bfloat16, "w"), _chk(bias, torch.bfloat16, "bias"), _chk(residual, torch.bfloat16, "residual")
    t = _T.op("linear")          # the dispatcher path (torch.ops.livecc_amd.linear) when the registration library is present
    if t is not None:
        return t(x, w, bias, int(epilogue), residual, 0 if packed_shape is None else int(N), 0 if packed_shape is None else int(K))
    out = torch.empty(M, N // 2 if epilogue == EPI_SWIGLU else N, dtype=torch.bfloat16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.lcc_gemm_bf16(_chk(x, torch.bfloat16, "x"), K, _chk(w, torch.bfloat16, "w"), K, 0 if packed_shape is None else 1,
                                 _chk(bias, torch.bfloat16, "bias"), _chk(residual, torch.bfloat16, "residual"), N,
                                 out.data_ptr(), out.shape[1], M, N, K, epilogue, None, 0, _st(x)), "lcc_gemm_bf16")
    return out


def linear_partial(x: torch.Tensor, w: torch.Tensor, nsplit: int, packed_shape: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """Skinny split-K path: fp32 slabs [nsplit, M, N] (M <= 16)."""
    M, K = x.shape
    N = w.shape[0] if packed_shape is None else packed_shape[0]
    out = torch.empty(nsplit, M, N, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.lcc_gemm_bf16(_chk(x, torch.bfloat16, "x"), K, _chk(w, torch.bfloat16, "w"), K, 0 if packed_shape is None else 1,
                                 None, None, 0, None, N, M, N, K, EPI_NONE, out.data_ptr(), nsplit, _st(x)), "lcc_gemm_bf16(partial)")
    return out


def quantize_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[N,K] -> (PACKED8 uint8 [N,K], fp32 row scales [N]) for linear_w8."""
    from .weights import pack_weight_fp8, quantize_fp8_rows
    q, scale = quantize_fp8_rows(w)
    return pack_weight_fp8(q), scale.contiguous()


def linear_w8(x: torch.Tensor, w8: torch.Tensor, scale: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
              residual: Optional[torch.Tensor] = None, nsplit: int = 0) -> torch.Tensor:
    """nn.Linear with fp8 e4m3 weights (PACKED8 bytes + fp32 row scales).  nsplit > 0: fp32 split-K slabs [nsplit, M, N]."""
    M, K = x.shape
    N = w8.shape[0]
    assert w8.dtype == torch.uint8 and w8.shape[1] == K and scale.dtype == torch.float32 and scale.numel() == N
    dq = torch.empty(N * K, dtype=torch.bfloat16, device=x.device) if M > 16 else None
    lib = _lib.load()
    if nsplit > 0:
        out = torch.empty(nsplit, M, N, dtype=torch.float32, device=x.device)
        _lib.check(lib.lcc_gemm_w8_bf16(_chk(x, torch.bfloat16, "x"), K, w8.data_ptr(), scale.data_ptr(), None, None, 0, None, N, M, N, K,
                                        EPI_NONE, out.data_ptr(), nsplit, dq.data_ptr() if dq is not None else None, _st(x)),
                   "lcc_gemm_w8_bf16(partial)")
        return out
    out = torch.empty(M, N // 2 if epilogue == EPI_SWIGLU else N, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.lcc_gemm_w8_bf16(_chk(x, torch.bfloat16, "x"), K, w8.data_ptr(), scale.data_ptr(), _chk(bias, torch.bfloat16, "bias"),
                                    _chk(residual, torch.bfloat16, "residual"), N, out.data_ptr(), out.shape[1], M, N, K, epilogue,
                                    None, 0, dq.data_ptr() if dq is not None else None, _st(x)), "lcc_gemm_w8_bf16")
    return out


GEMV_DEFAULT_VARIANT = 1


def set_gemv_variant(v: int) -> None:
    _lib.load().lcc_debug_set_gemv_variant(int(v))


GEMM_DEFAULT_VARIANT = 2
ATTN_DEFAULT_VARIANT = 3


LAUNCH_COUNTERS = ("attn_decode", "attn_decode_combine", "attn_decode_fused", "attn_decode_fused_merge", "gemv_fused_tail", "dgemv_v2",
                   "attn_prefill_mfma32", "attn_prefill_shared", "attn_prefill_per_wave", "attn_prefill_combine", "last_decode_nsplit",
                   "last_prefill_nsplit", "gemm_tall", "attn_vit32", "gemm_vh", "gemm_vit_qkv")


def attn_tile_rows(n_q_heads: int, n_kv_heads: int) -> int:
    """The tallest query tile the engine builds for the LLM prefill attention = the largest tile_rows attn_prefill takes besides 16 / 32
    (lcc_debug_attn_tile_rows: host logic, no GPU): 36 at 28 / 4 heads under the default attention variant (8 x 32 packed (row, head)
    columns per block / 7 heads), 32 at 64 / 8."""
    rc = _lib.load().lcc_debug_attn_tile_rows(int(n_q_heads), int(n_kv_heads))
    if rc < 0:
        _lib.check(rc, "lcc_debug_attn_tile_rows")
    return int(rc)


def attn_plan(n_new: Sequence[int], max_kv: int, n_q_heads: int, n_kv_heads: int, cu_count: int = 256) -> Tuple[int, int]:
    """(tile_rows, key_splits) of one LLM prefill attention launch as the engine plans it (lcc_debug_attn_plan: host logic, no GPU);
    (0, 0) when another kernel family serves the call."""
    import numpy as np
    nn = np.ascontiguousarray(np.asarray(list(n_new), dtype=np.int32))
    out = np.zeros(2, dtype=np.int32)
    _lib.check(_lib.load().lcc_debug_attn_plan(nn.ctypes.data, int(nn.size), int(max_kv), int(n_q_heads), int(n_kv_heads), int(cu_count),
                                               out[0:].ctypes.data, out[1:].ctypes.data), "lcc_debug_attn_plan")
    return int(out[0]), int(out[1])


def gemm_plan(M: int, N: int, K: int, epilogue: int = EPI_NONE, nsplit: int = 0, w_fp8: bool = False) -> Tuple[int, int]:
    """(tile_rows, engine_splits) of lcc_debug_gemm_plan: the kernel family that serves this packed-weight GEMM and the split count the
    engine's prefill asks for at this shape.  Host logic only: needs the library, not a GPU."""
    import numpy as np
    out = np.zeros(2, dtype=np.int32)
    _lib.check(_lib.load().lcc_debug_gemm_plan(int(M), int(N), int(K), int(epilogue), int(nsplit), 1 if w_fp8 else 0, out[0:].ctypes.data,
                                               out[1:].ctypes.data), "lcc_debug_gemm_plan")
    return int(out[0]), int(out[1])


def launch_counts(reset: bool = False) -> dict:
    """Host-side launch counters of the library (which kernel served the calls since the last reset): lcc_debug_launch_counts."""
    import numpy as np
    buf = np.zeros(16, dtype=np.int64)
    _lib.check(_lib.load().lcc_debug_launch_counts(buf.ctypes.data, 16, 1 if reset else 0), "lcc_debug_launch_counts")
    return {k: int(buf[i]) for i, k in enumerate(LAUNCH_COUNTERS)}


_ATTN_VARIANT = [ATTN_DEFAULT_VARIANT]


def set_attn_variant(v: int) -> None:
    _lib.load().lcc_debug_set_attn_variant(int(v))
    _ATTN_VARIANT[0] = int(v)


def set_gemm_variant(v: int) -> None:
    _lib.load().lcc_debug_set_gemm_variant(int(v))


def set_skinny_rows(rows: int) -> int:
    """Largest M served by the weight-streaming GEMV kernels (16..64, default 64); returns the previous value."""
    return _lib.load().lcc_debug_set_skinny_rows(int(rows))


def set_vit_fused_qkv(on: bool) -> int:
    """Vision tower: q|k|v projection with RoPE + V transpose in its epilogue (default on) vs separate launches; returns the previous value."""
    return _lib.load().lcc_debug_set_vit_fused_qkv(1 if on else 0)


def gemv_num_splits(N: int, K: int) -> int:
    return _lib.load().lcc_gemv_num_splits(N, K)


def layernorm(x, w, b, eps=1e-6):
    _chk(x, torch.bfloat16, "x"), _chk(w, torch.bfloat16, "w"), _chk(b, torch.bfloat16, "b")
    t = _T.op("layernorm")
    if t is not None:
        return t(x, w, b, float(eps))
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    _lib.check(_lib.load().lcc_layernorm_bf16(_chk(x, torch.bfloat16, "x"), _chk(w, torch.bfloat16, "w"),
                                              _chk(b, torch.bfloat16, "b"), y.data_ptr(), rows, x.shape[-1], eps, _st(x)),
               "lcc_layernorm_bf16")
    return y


def rmsnorm(x, w, eps=1e-6):
    _chk(x, torch.bfloat16, "x"), _chk(w, torch.bfloat16, "w")
    t = _T.op("rmsnorm")
    if t is not None:
        return t(x, w, float(eps))
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    _lib.check(_lib.load().lcc_rmsnorm_bf16(_chk(x, torch.bfloat16, "x"), _chk(w, torch.bfloat16, "w"), y.data_ptr(), rows,
                                            x.shape[-1], eps, _st(x)), "lcc_rmsnorm_bf16")
    return y


def add_rmsnorm_(h, w, eps=1e-6, delta=None, partial=None):
    """In place h += delta; returns rmsnorm(h)*w (or None when w is None)."""
    y = torch.empty_like(h) if w is not None else None
    rows = h.numel() // h.shape[-1]
    nsplit = partial.shape[0] if partial is not None else 0
    _lib.check(_lib.load().lcc_add_rmsnorm_bf16(_chk(h, torch.bfloat16, "h"), _chk(delta, torch.bfloat16, "delta"),
                                                _chk(partial, torch.float32, "partial"), nsplit, _chk(w, torch.bfloat16, "w"),
                                                y.data_ptr() if y is not None else None, rows, h.shape[-1], eps, _st(h)),
               "lcc_add_rmsnorm_bf16")
    return y


def swiglu(gate, up):
    _chk(gate, torch.bfloat16, "gate"), _chk(up, torch.bfloat16, "up")
    t = _T.op("swiglu")
    if t is not None:
        return t(gate, up)
    out = torch.empty_like(gate)
    _lib.check(_lib.load().lcc_swiglu_bf16(_chk(gate, torch.bfloat16, "gate"), _chk(up, torch.bfloat16, "up"), out.data_ptr(),
                                           gate.numel(), _st(gate)), "lcc_swiglu_bf16")
    return out


def patchify_norm(frames: torch.Tensor, layout: str, mean255, std255) -> torch.Tensor:
    lay = 0 if layout == "THWC" else 1
    if lay == 0:
        T, H, W, _ = frames.shape
    else:
        T, _, H, W = frames.shape
    _chk(frames, torch.uint8, "frames")
    t = _T.op("patchify_norm")
    if t is not None:
        return t(frames, lay, [float(v) for v in mean255], [float(v) for v in std255])
    P = ((T + 1) // 2) * (H // 14) * (W // 14)
    out = torch.empty(P, 1176, dtype=torch.bfloat16, device=frames.device)
    m = (C.c_float * 3)(*[float(v) for v in mean255])
    s = (C.c_float * 3)(*[float(v) for v in std255])
    _lib.check(_lib.load().lcc_patchify_norm_u8(_chk(frames, torch.uint8, "frames"), lay, T, H, W, m, s, out.data_ptr(), 1176,
                                                _st(frames)), "lcc_patchify_norm_u8")
    return out


def cast_f32_bf16(x):
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().lcc_cast_f32_bf16(_chk(x, torch.float32, "x"), out.data_ptr(), x.numel(), _st(x)), "lcc_cast_f32_bf16")
    return out


def _i32(a, device) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a, dtype=np.int32), device=device)


def vit_segments(grids: Sequence[Sequence[int]], device):
    """Segment / tile tables of the ViT attention for clips with grids (t,h,w): one segment per temporal slice."""
    seg_start, seg_len, seg_blk, seg_of_patch, tile_seg, tile_q0, grp_seg, grp_q0 = [], [], [], [], [], [], [], []
    g8_seg, g8_q0 = [], []        # groups of 8 x 32 rows (attention variant 3, attn_vit32_kernel)
    P = blocks = 0
    for t, h, w in grids:
        n = h * w
        for _ in range(t):
            sg = len(seg_start)
            seg_start.append(P); seg_len.append(n); seg_blk.append(blocks)
            for q in range(0, n, 32):
                tile_seg.append(sg); tile_q0.append(q)
            for q in range(0, n, 128):
                grp_seg.append(sg); grp_q0.append(q)
            for q in range(0, n, 256):
                g8_seg.append(sg); g8_q0.append(q)
            seg_of_patch += [sg] * n
            P += n; blocks += (n + 31) // 32
    d = dict(seg_start=_i32(seg_start, device), seg_len=_i32(seg_len, device), seg_blk=_i32(seg_blk, device),
             seg_of_patch=_i32(seg_of_patch, device), tile_seg=_i32(tile_seg, device), tile_q0=_i32(tile_q0, device),
             grp_seg=_i32(grp_seg, device), grp_q0=_i32(grp_q0, device), n_groups=len(grp_seg),
             g8_seg=_i32(g8_seg, device), g8_q0=_i32(g8_q0, device), n_groups8=len(g8_seg),
             P=P, blocks=blocks, n_tiles=len(tile_seg))
    return d


def vit_attention(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, grids, heads: int, group_rows: int = 256) -> torch.Tensor:
    """qkv [P, 3E] bf16 (un-rotated; modified in place), cos/sin fp32 [P,40] -> attention output [P, E].  group_rows (attention
    variant 3 only): 256 = 8-wave workgroups, 128 = 4-wave workgroups (the engine's choice for grids that do not fill the chip)."""
    lib = _lib.load()
    seg = vit_segments(grids, qkv.device)
    P, E = qkv.shape[0], qkv.shape[1] // 3
    vt = torch.zeros(heads * seg["blocks"] * 80 * 32, dtype=torch.bfloat16, device=qkv.device)
    out = torch.empty(P, E, dtype=torch.bfloat16, device=qkv.device)
    _lib.check(lib.lcc_vit_rope_vt_bf16(_chk(qkv, torch.bfloat16, "qkv"), _chk(cos, torch.float32, "cos"),
                                        _chk(sin, torch.float32, "sin"), seg["seg_of_patch"].data_ptr(),
                                        seg["seg_start"].data_ptr(), seg["seg_blk"].data_ptr(), vt.data_ptr(), P, heads,
                                        seg["blocks"], _st(qkv)), "lcc_vit_rope_vt_bf16")
    if _ATTN_VARIANT[0] == 3:
        gs, gq, ng = (("g8_seg", "g8_q0", "n_groups8") if group_rows == 256 else ("grp_seg", "grp_q0", "n_groups"))
        _lib.check(lib.lcc_attn_vit32_bf16(qkv.data_ptr(), vt.data_ptr(), out.data_ptr(), seg[gs].data_ptr(), seg[gq].data_ptr(),
                                           seg["seg_start"].data_ptr(), seg["seg_len"].data_ptr(), seg["seg_blk"].data_ptr(), seg[ng],
                                           heads, seg["blocks"], int(group_rows), _st(qkv)), "lcc_attn_vit32_bf16")
        return out
    _lib.check(lib.lcc_attn_vit_bf16(qkv.data_ptr(), vt.data_ptr(), out.data_ptr(), seg["tile_seg"].data_ptr(),
                                     seg["tile_q0"].data_ptr(), seg["seg_start"].data_ptr(), seg["seg_len"].data_ptr(),
                                     seg["seg_blk"].data_ptr(), seg["n_tiles"], heads, seg["blocks"], seg["grp_seg"].data_ptr(),
                                     seg["grp_q0"].data_ptr(), seg["n_groups"], _st(qkv)),
               "lcc_attn_vit_bf16")
    return out


def mrope_table(pos3: torch.Tensor, inv_freq: torch.Tensor, sections) -> Tuple[torch.Tensor, torch.Tensor]:
    S = pos3.shape[1]
    cos = torch.empty(S, 64, dtype=torch.bfloat16, device=pos3.device)
    sin = torch.empty_like(cos)
    _lib.check(_lib.load().lcc_mrope_table(_chk(pos3, torch.int32, "pos3"), _chk(inv_freq, torch.float32, "inv_freq"), S,
                                           int(sections[0]), int(sections[1]), cos.data_ptr(), sin.data_ptr(), _st(pos3)),
               "lcc_mrope_table")
    return cos, sin


class KvArena:
    """Stand-alone KV arena(s) for the operator-level attention tests / plugin: [layer][K|V][Hkv][Lmax][128]."""

    def __init__(self, n_slots, n_layers, n_kv_heads, lmax, device):
        self.lay = _lib.KvLayout(n_layers, n_kv_heads, lmax, 128)
        self.per_slot = n_layers * 2 * n_kv_heads * lmax * 128
        self.buf = torch.zeros(n_slots, self.per_slot, dtype=torch.bfloat16, device=device)
        ptrs = np.asarray([self.buf[i].data_ptr() for i in range(n_slots)], dtype=np.uint64)
        self.ptrs = torch.as_tensor(ptrs.view(np.int64), device=device)
        self.n_kv_heads, self.lmax, self.n_layers = n_kv_heads, lmax, n_layers

    def k_view(self, slot, layer):  # [Hkv, Lmax, 128]
        o = layer * 2 * self.n_kv_heads * self.lmax * 128
        return self.buf[slot, o:o + self.n_kv_heads * self.lmax * 128].view(self.n_kv_heads, self.lmax, 128)

    def v_view(self, slot, layer):  # logical [Hkv, Lmax, 128] gathered from the blocked-transposed storage
        o = (layer * 2 + 1) * self.n_kv_heads * self.lmax * 128
        raw = self.buf[slot, o:o + self.n_kv_heads * self.lmax * 128].view(self.n_kv_heads, self.lmax // 32, 128, 32)
        return raw.permute(0, 1, 3, 2).reshape(self.n_kv_heads, self.lmax, 128)


def rope_kv_append(qkv: Optional[torch.Tensor], cos, sin, tok_stream, tok_pos, kv: KvArena, layer: int, n_q_heads: int,
                   partial: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                   kv_len: Optional[torch.Tensor] = None) -> torch.Tensor:
    S = cos.shape[0]
    t = _T.op("rope_kv_append")
    if t is not None:
        for a, dt, nm in ((qkv, torch.bfloat16, "qkv"), (partial, torch.float32, "partial"), (bias, torch.bfloat16, "bias"), (cos, torch.bfloat16, "cos"),
                          (sin, torch.bfloat16, "sin"), (tok_stream, torch.int32, "tok_stream"), (tok_pos, torch.int32, "tok_pos"), (kv_len, torch.int32, "kv_len")):
            _chk(a, dt, nm)
        return t(qkv, partial, bias, cos, sin, tok_stream, tok_pos, kv_len, kv.ptrs, kv.buf, kv.n_layers, kv.n_kv_heads, kv.lmax, int(layer), int(n_q_heads))
    # n_q_heads == 0: append-only use (the HF Cache plugin): the kernel still wants a valid pointer
    q = torch.empty(S, n_q_heads * 128, dtype=torch.bfloat16, device=cos.device)
    dummy = torch.empty(16, dtype=torch.bfloat16, device=cos.device) if n_q_heads == 0 else None
    _lib.check(_lib.load().lcc_rope_kv_append_bf16(
        _chk(qkv, torch.bfloat16, "qkv"), _chk(partial, torch.float32, "partial"), partial.shape[0] if partial is not None else 0,
        _chk(bias, torch.bfloat16, "bias"), _chk(cos, torch.bfloat16, "cos"), _chk(sin, torch.bfloat16, "sin"),
        _chk(tok_stream, torch.int32, "tok_stream"), _chk(tok_pos, torch.int32, "tok_pos"), _chk(kv_len, torch.int32, "kv_len"),
        kv.ptrs.data_ptr(), kv.lay, layer, q.data_ptr() if n_q_heads > 0 else dummy.data_ptr(), S, n_q_heads, _st(cos)),
        "lcc_rope_kv_append_bf16")
    return q


def attn_prefill(q: torch.Tensor, kv: KvArena, layer: int, segments: Sequence[Tuple[int, int, int]], n_q_heads: int,
                 tile_rows: int = 32, nsplit: int = 1):
    """segments: (slot, n_new, past_len) per stream, rows packed in that order."""
    ts, tq, tn, tp = [], [], [], []
    row = 0
    for slot, n_new, past in segments:
        for o in range(0, n_new, tile_rows):
            ts.append(slot); tq.append(row + o); tn.append(min(tile_rows, n_new - o)); tp.append(past + o)
        row += n_new
    dev = q.device
    a, b, c, d = _i32(ts, dev), _i32(tq, dev), _i32(tn, dev), _i32(tp, dev)
    _chk(q, torch.bfloat16, "q")
    t = _T.op("attn_prefill")
    if t is not None:
        return t(q, kv.ptrs, kv.buf, kv.n_layers, kv.n_kv_heads, kv.lmax, int(layer), a, b, c, d, int(n_q_heads), int(tile_rows), int(nsplit))
    out = torch.empty_like(q)
    ws_o = torch.empty(q.shape[0] * n_q_heads * nsplit * 128, dtype=torch.float32, device=dev) if nsplit > 1 else None
    ws_ml = torch.empty(q.shape[0] * n_q_heads * nsplit * 2, dtype=torch.float32, device=dev) if nsplit > 1 else None
    _lib.check(_lib.load().lcc_attn_prefill_bf16(_chk(q, torch.bfloat16, "q"), out.data_ptr(), a.data_ptr(), b.data_ptr(),
                                                 c.data_ptr(), d.data_ptr(), kv.ptrs.data_ptr(), kv.lay, layer, len(ts),
                                                 n_q_heads, tile_rows, nsplit, q.shape[0], ws_o.data_ptr() if ws_o is not None else None,
                                                 ws_ml.data_ptr() if ws_ml is not None else None, _st(q)), "lcc_attn_prefill_bf16")
    return out


def attn_decode(q: torch.Tensor, kv: KvArena, layer: int, slots: torch.Tensor, kv_len: torch.Tensor, n_q_heads: int,
                nsplit: int):
    B = q.shape[0]
    _chk(q, torch.bfloat16, "q"), _chk(slots, torch.int32, "slots"), _chk(kv_len, torch.int32, "kv_len")
    t = _T.op("attn_decode")
    if t is not None:
        return t(q, kv.ptrs, kv.buf, kv.n_layers, kv.n_kv_heads, kv.lmax, int(layer), slots, kv_len, int(n_q_heads), int(nsplit))
    out = torch.empty_like(q)
    ws_o = torch.empty(B * kv.n_kv_heads * nsplit * 16 * 128, dtype=torch.float32, device=q.device)
    ws_ml = torch.empty(B * kv.n_kv_heads * nsplit * 16 * 2, dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().lcc_attn_decode_bf16(_chk(q, torch.bfloat16, "q"), out.data_ptr(), _chk(slots, torch.int32, "slots"),
                                                _chk(kv_len, torch.int32, "kv_len"), kv.ptrs.data_ptr(), kv.lay, layer, B,
                                                n_q_heads, nsplit, ws_o.data_ptr(), ws_ml.data_ptr(), _st(q)),
               "lcc_attn_decode_bf16")
    return out


def embed_gather(ids: torch.Tensor, table: torch.Tensor, vit_index: Optional[torch.Tensor] = None,
                 vit_rows: Optional[torch.Tensor] = None, indirect: Optional[torch.Tensor] = None, n_rows: Optional[int] = None):
    """`embed_tokens(input_ids)` + `masked_scatter` of the video rows (Q2VL:1159-1176) in one pass: row s of the result is
    `vit_rows[vit_index[s]]` where `vit_index[s] >= 0`, else `table[ids[s]]` (decode: `table[ids[indirect[s]]]`, the current token of
    stream slot `indirect[s]`).  int32 index tensors on the GPU; bf16 rows, dim % 8 == 0."""
    S = int(n_rows if n_rows is not None else (indirect.numel() if indirect is not None else ids.numel()))
    out = torch.empty(S, table.shape[1], dtype=torch.bfloat16, device=table.device)
    _lib.check(_lib.load().lcc_embed_gather_bf16(
        _chk(ids, torch.int32, "ids"), _chk(indirect, torch.int32, "indirect") if indirect is not None else None,
        _chk(vit_index, torch.int32, "vit_index") if vit_index is not None else None, _chk(table, torch.bfloat16, "table"),
        _chk(vit_rows, torch.bfloat16, "vit_rows") if vit_rows is not None else None, out.data_ptr(), S, table.shape[1], _st(out)),
        "lcc_embed_gather_bf16")
    return out


def attn_decode_fused(qkv_partial: torch.Tensor, bias: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, kv: KvArena, layer: int,
                      slots: torch.Tensor, kv_len: torch.Tensor, n_q_heads: int, nsplit: int, counters: Optional[torch.Tensor] = None):
    """qkv_partial: fp32 [NS, B, (Hq+2Hkv)*128] split-K slabs of the q/k/v Linear.  Appends the new K/V at kv_len[slot] and returns
    the attention output [B, Hq*128] (one launch)."""
    NS, B, _ = qkv_partial.shape
    out = torch.empty(B, n_q_heads * 128, dtype=torch.bfloat16, device=qkv_partial.device)
    ws_o = torch.empty(B * kv.n_kv_heads * nsplit * 16 * 128, dtype=torch.float32, device=out.device)
    ws_ml = torch.empty(B * kv.n_kv_heads * nsplit * 16 * 2, dtype=torch.float32, device=out.device)
    if counters is None:
        counters = torch.zeros(B * kv.n_kv_heads, dtype=torch.int32, device=out.device)
    _lib.check(_lib.load().lcc_attn_decode_fused_bf16(
        _chk(qkv_partial, torch.float32, "qkv_partial"), NS, _chk(bias, torch.bfloat16, "bias"), _chk(cos, torch.bfloat16, "cos"),
        _chk(sin, torch.bfloat16, "sin"), _chk(slots, torch.int32, "slots"), _chk(kv_len, torch.int32, "kv_len"), kv.ptrs.data_ptr(),
        kv.lay, layer, out.data_ptr(), B, n_q_heads, nsplit, ws_o.data_ptr(), ws_ml.data_ptr(), counters.data_ptr(), _st(out)),
        "lcc_attn_decode_fused_bf16")
    return out, counters


def sample_greedy(logits: torch.Tensor, seen: torch.Tensor, slots: torch.Tensor, repetition_penalty: float = 1.0,
                  thr_token: int = -1, thr_value: Optional[float] = None, eos_token: int = -1, suppress_eos: bool = False,
                  want_scores: bool = False, two_stage: bool = False, eos_token2: int = -1, done: Optional[torch.Tensor] = None):
    B, V = logits.shape
    n_slots, words = seen.shape
    if not two_stage and done is None:          # the single-launch form is what torch.ops.livecc_amd.sample_greedy registers
        _chk(logits, torch.bfloat16, "logits"), _chk(seen, torch.int32, "seen"), _chk(slots, torch.int32, "slots")
        t = _T.op("sample_greedy")
        if t is not None:
            out, scores = t(logits, seen, slots, float(repetition_penalty), int(thr_token), thr_value is not None, float(thr_value or 0.0),
                            int(eos_token), int(eos_token2), bool(suppress_eos), bool(want_scores))
            return out, (scores if want_scores else None)
    out = torch.zeros(n_slots, dtype=torch.int32, device=logits.device)
    scores = torch.empty(B, V, dtype=torch.float32, device=logits.device) if want_scores else None
    ws = torch.empty(B * 256, dtype=torch.float32, device=logits.device) if two_stage else None
    _lib.check(_lib.load().lcc_sample_greedy(
        _chk(logits, torch.bfloat16, "logits"), V, B, V, _chk(seen, torch.int32, "seen"), words, _chk(slots, torch.int32, "slots"),
        repetition_penalty, thr_token, 1 if thr_value is not None else 0, float(thr_value or 0.0), eos_token, eos_token2,
        1 if suppress_eos else 0, _chk(done, torch.int32, "done") if done is not None else None, out.data_ptr(), None, 0, None,
        scores.data_ptr() if scores is not None else None, ws.data_ptr() if ws is not None else None, _st(logits)), "lcc_sample_greedy")
    return out, scores


def sample_topk_topp(logits: torch.Tensor, seen: torch.Tensor, slots: torch.Tensor, temperature: float = 1.0, top_k: int = 0,
                     top_p: float = 1.0, seed: int = 0, rng_ctr: Optional[torch.Tensor] = None, repetition_penalty: float = 1.0,
                     thr_token: int = -1, thr_value: Optional[float] = None, eos_token: int = -1, eos_token2: int = -1,
                     suppress_eos: bool = False, want_scores: bool = False):
    """do_sample=True path: processors -> temperature -> top-k -> top-p -> one multinomial draw per stream.  Returns
    (tokens int32 [n_slots], processed scores fp32 [B,V] or None).  `rng_ctr`: int32 [n_slots] draw counters (advanced in place)."""
    B, V = logits.shape
    n_slots, words = seen.shape
    out = torch.zeros(n_slots, dtype=torch.int32, device=logits.device)
    scores = torch.empty(B, V, dtype=torch.float32, device=logits.device) if want_scores else None
    if rng_ctr is None:
        rng_ctr = torch.zeros(n_slots, dtype=torch.int32, device=logits.device)
    _lib.check(_lib.load().lcc_sample_topk_topp(
        _chk(logits, torch.bfloat16, "logits"), V, B, V, _chk(seen, torch.int32, "seen"), words, _chk(slots, torch.int32, "slots"),
        repetition_penalty, thr_token, 1 if thr_value is not None else 0, float(thr_value or 0.0), eos_token, eos_token2,
        1 if suppress_eos else 0, None, out.data_ptr(), None, 0, None, scores.data_ptr() if scores is not None else None,
        float(temperature), int(top_k), float(top_p), int(seed) & 0xFFFFFFFFFFFFFFFF, _chk(rng_ctr, torch.int32, "rng_ctr"),
        _st(logits)), "lcc_sample_topk_topp")
    return out, scores


# ---- decode pipeline v2 operators (csrc/decode_v2.hip) ----
def tile_stats(h: torch.Tensor) -> torch.Tensor:
    """fp32 [M, K/16]: sums of squares of every 16-channel tile of the bf16 rows (what the residual epilogues emit)."""
    M, K = h.shape
    return h.float().pow(2).view(M, K // 16, 16).sum(-1).contiguous()


def _w_ptr(w: torch.Tensor, wscale: Optional[torch.Tensor]) -> int:
    """weight pointer of the dgemv operators: packed bf16, or (with `wscale`) e4m3 bytes in the PACKED8 order"""
    return _chk(w, torch.uint8 if wscale is not None else torch.bfloat16, "w")


def dgemv_norm_linear(w_packed: torch.Tensor, h: torch.Tensor, stats: torch.Tensor, norm_w: torch.Tensor, eps: float, shape: Tuple[int, int],
                      bias: Optional[torch.Tensor] = None, swiglu: bool = False, wscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    N, K = shape
    M = h.shape[0]
    out = torch.empty(M, N // 2 if swiglu else N, dtype=torch.bfloat16, device=h.device)
    _lib.check(_lib.load().lcc_dgemv_norm_linear(_w_ptr(w_packed, wscale), _chk(wscale, torch.float32, "wscale"), _chk(h, torch.bfloat16, "h"), _chk(stats, torch.float32, "stats"),
                                                 _chk(norm_w, torch.bfloat16, "norm_w"), float(eps), _chk(bias, torch.bfloat16, "bias"),
                                                 out.data_ptr(), out.shape[1], M, N, K, 1 if swiglu else 0, _st(h)), "lcc_dgemv_norm_linear")
    return out


def dgemv_resid_(w_packed: torch.Tensor, x: torch.Tensor, h: torch.Tensor, shape: Tuple[int, int], wscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """h += Linear(x) in place; returns the per-tile sums of squares of the new h."""
    N, K = shape
    M = x.shape[0]
    stats = torch.empty(M, N // 16, dtype=torch.float32, device=h.device)
    _lib.check(_lib.load().lcc_dgemv_resid(_w_ptr(w_packed, wscale), _chk(wscale, torch.float32, "wscale"), _chk(x, torch.bfloat16, "x"), x.shape[1], _chk(h, torch.bfloat16, "h"),
                                           stats.data_ptr(), M, N, K, _st(h)), "lcc_dgemv_resid")
    return stats


def dgemv_qkv_rope(w_dec_packed: torch.Tensor, h: torch.Tensor, stats: torch.Tensor, norm_w: torch.Tensor, eps: float, bias: torch.Tensor,
                   cos: torch.Tensor, sin: torch.Tensor, tok_stream: torch.Tensor, kv_len: torch.Tensor, kv: "KvArena", layer: int,
                   n_q_heads: int, wscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    M, K = h.shape
    q = torch.empty(M, n_q_heads * 128, dtype=torch.bfloat16, device=h.device)
    _lib.check(_lib.load().lcc_dgemv_qkv_rope(
        _w_ptr(w_dec_packed, wscale), _chk(wscale, torch.float32, "wscale"), _chk(h, torch.bfloat16, "h"), _chk(stats, torch.float32, "stats"), _chk(norm_w, torch.bfloat16, "norm_w"),
        float(eps), _chk(bias, torch.bfloat16, "bias"), _chk(cos, torch.bfloat16, "cos"), _chk(sin, torch.bfloat16, "sin"),
        _chk(tok_stream, torch.int32, "tok_stream"), _chk(kv_len, torch.int32, "kv_len"), kv.ptrs.data_ptr(), kv.lay, layer, q.data_ptr(),
        n_q_heads, M, K, _st(h)), "lcc_dgemv_qkv_rope")
    return q


def dgemv_down_qkv_(w_down_packed: torch.Tensor, x: torch.Tensor, h: torch.Tensor, k_down: int, w_qkv_dec_packed: torch.Tensor,
                    norm_w: torch.Tensor, eps: float, bias: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, tok_stream: torch.Tensor,
                    kv_len: torch.Tensor, kv: "KvArena", layer: int, n_q_heads: int, counter: torch.Tensor, counter_before: int,
                    err: torch.Tensor):
    """The chained decode launch as an operator (lcc_dgemv_down_qkv): h += Linear_down(x) in place (+ tile statistics), then q|k|v of
    RMSNorm(h) with bias + M-RoPE + KV append.  Returns (stats [M, hidden/16], q [M, Hq*128]).  `counter` / `err`: int32 [1] device
    words; `counter_before` = the counter's value before this call (it grows by hidden/16 per call)."""
    M, hidden = h.shape
    stats = torch.empty(M, hidden // 16, dtype=torch.float32, device=h.device)
    q = torch.empty(M, n_q_heads * 128, dtype=torch.bfloat16, device=h.device)
    _lib.check(_lib.load().lcc_dgemv_down_qkv(
        _chk(w_down_packed, torch.bfloat16, "w_down"), _chk(x, torch.bfloat16, "x"), x.shape[1], _chk(h, torch.bfloat16, "h"), stats.data_ptr(),
        int(k_down), _chk(w_qkv_dec_packed, torch.bfloat16, "w_qkv"), _chk(norm_w, torch.bfloat16, "norm_w"), float(eps),
        _chk(bias, torch.bfloat16, "bias"), _chk(cos, torch.bfloat16, "cos"), _chk(sin, torch.bfloat16, "sin"),
        _chk(tok_stream, torch.int32, "tok_stream"), _chk(kv_len, torch.int32, "kv_len"), kv.ptrs.data_ptr(), kv.lay, layer, q.data_ptr(),
        n_q_heads, M, hidden, _chk(counter, torch.int32, "counter"), int(counter_before), _chk(err, torch.int32, "err"), _st(h)),
        "lcc_dgemv_down_qkv")
    return stats, q
