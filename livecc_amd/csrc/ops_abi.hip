// Operator-level C-ABI shims (include/livecc_amd.h, "operator plugins"): one extern "C" entry per kernel family, argument validation,
// launch on the caller's stream; plus the library-wide debug switches and launch counters.
#include "engine_internal.h"

namespace lcc { long long g_launch_counts[LC_COUNT] = {}; }
extern "C" int lcc_debug_launch_counts(int64_t* out, int n, int reset) {
  if (n < 0 || n > LC_COUNT || (n > 0 && !out)) return fail(LCC_ERR_ARG, "bad argument");
  for (int i = 0; i < n; ++i) out[i] = (int64_t)g_launch_counts[i];
  if (reset) for (int i = 0; i < LC_COUNT; ++i) g_launch_counts[i] = 0;
  return 0;
}
// ------------------------------------------------------------------------------------------------
// operator-level C-ABI wrappers
// ------------------------------------------------------------------------------------------------
static KvLayout to_lay(lcc_kv_layout l) { return KvLayout{l.n_layers, l.n_kv_heads, l.lmax, l.head_dim}; }
#define OP_RET(call, name)                                   \
  do {                                                       \
    int r__ = (call);                                        \
    if (r__ != 0) return fail(r__, "%s: invalid arguments (%d)", name, r__); \
    return check_launch(name);                               \
  } while (0)

// process-global routing knobs: refused while a model-level call is in flight (engine_internal.h: g_calls_in_flight)
extern "C" int lcc_debug_set_gemv_variant(int variant) { if (int kg__ = lcc_knob_guard("lcc_debug_set_gemv_variant")) return kg__; set_gemv_variant(variant); return 0; }
extern "C" int lcc_debug_set_gemm_variant(int variant) { if (int kg__ = lcc_knob_guard("lcc_debug_set_gemm_variant")) return kg__; set_gemm_variant(variant); return 0; }
extern "C" int lcc_debug_set_attn_variant(int variant) { if (int kg__ = lcc_knob_guard("lcc_debug_set_attn_variant")) return kg__; set_attn_variant(variant); return 0; }
extern "C" int lcc_debug_attn_tile_rows(int n_q_heads, int n_kv_heads) {   // host logic only: no launch
  if (n_kv_heads < 1 || n_q_heads < n_kv_heads || n_q_heads % n_kv_heads) return fail(LCC_ERR_ARG, "lcc_debug_attn_tile_rows: invalid head counts");
  const int G = n_q_heads / n_kv_heads;
  return (get_attn_variant() == 3 && G <= 8) ? attn32_tile_rows(G) : 32;
}
extern "C" int lcc_debug_attn_plan(const int32_t* n_new, int n_streams, int max_kv, int n_q_heads, int n_kv_heads, int cu_count, int32_t* tile_rows,
                                   int32_t* splits) {   // host logic only: no launch
  if (!n_new || n_streams < 1 || max_kv < 1 || n_kv_heads < 1 || n_q_heads < n_kv_heads || n_q_heads % n_kv_heads || cu_count < 1 || !tile_rows || !splits)
    return fail(LCC_ERR_ARG, "lcc_debug_attn_plan: invalid arguments");
  for (int b = 0; b < n_streams; ++b)
    if (n_new[b] < 1) return fail(LCC_ERR_ARG, "lcc_debug_attn_plan: stream %d has no new rows", b);
  const int G = n_q_heads / n_kv_heads;
  if (get_attn_variant() != 3 || G > 8) { *tile_rows = 0; *splits = 0; return 0; }      // another kernel family plans this call
  int tr = 32, ks = 1;
  attn32_plan(n_new, n_streams, max_kv, G, n_kv_heads, cu_count, &tr, &ks);
  *tile_rows = tr; *splits = ks;
  return 0;
}
extern "C" int lcc_debug_gemm_plan(int M, int N, int K, int epilogue, int nsplit, int w_fp8, int32_t* tile_rows, int32_t* engine_splits) {
  if (M <= 0 || N <= 0 || K <= 0 || !tile_rows || !engine_splits) return fail(LCC_ERR_ARG, "lcc_debug_gemm_plan: invalid arguments");
  *engine_splits = gemm_tiled_num_splits(M, N, K, w_fp8 == 0);
  *tile_rows = w_fp8 ? 0 : gemm_plan(M, N, K, epilogue, nsplit, false);      // fp8 weights route inside gemm_w8 (not modelled here)
  return 0;
}
extern "C" int lcc_debug_set_fused_tails(int on);
extern "C" int lcc_gemm_bf16(const void* A, int lda, const void* W, int ldw, int w_layout, const void* bias, const void* residual,
                             int ldr, void* C, int ldc, int M, int N, int K, int epilogue, float* partial, int nsplit, void* stream) {
  if (!A || !W || (!C && !partial)) return fail(LCC_ERR_ARG, "lcc_gemm_bf16: null pointer");
  if (w_layout != 0 && w_layout != 1) return fail(LCC_ERR_ARG, "lcc_gemm_bf16: w_layout must be 0 or 1");
  GemmArgs g; g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.w_packed = w_layout; g.bias = (const bf16_t*)bias;
  g.residual = (const bf16_t*)residual; g.ldr = ldr; g.C = (bf16_t*)C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.epilogue = epilogue;
  g.partial = partial; g.nsplit = nsplit;
  if (partial && C == nullptr) g.C = (bf16_t*)partial;  // alignment check only
  OP_RET(gemm_bf16(g, (hipStream_t)stream), "lcc_gemm_bf16");
}
extern "C" int lcc_gemv_num_splits(int N, int K) { return gemv_num_splits(N, K); }
extern "C" int lcc_gemm_w8_bf16(const void* A, int lda, const void* W8, const float* wscale, const void* bias, const void* residual,
                                int ldr, void* C, int ldc, int M, int N, int K, int epilogue, float* partial, int nsplit,
                                void* dq_scratch, void* stream) {
  if (!A || !W8 || !wscale || (!C && !partial)) return fail(LCC_ERR_ARG, "lcc_gemm_w8_bf16: null pointer");
  if (M > 16 && !dq_scratch) return fail(LCC_ERR_ARG, "lcc_gemm_w8_bf16: M > 16 needs dq_scratch (N*K bf16)");
  GemmArgs g; g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W8; g.ldw = K; g.w_packed = 1; g.bias = (const bf16_t*)bias;
  g.residual = (const bf16_t*)residual; g.ldr = ldr; g.C = (bf16_t*)C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.epilogue = epilogue;
  g.partial = partial; g.nsplit = nsplit; g.w_fp8 = 1; g.wscale = wscale; g.dq_scratch = (bf16_t*)dq_scratch;
  if (partial && C == nullptr) g.C = (bf16_t*)partial;
  OP_RET(gemm_bf16(g, (hipStream_t)stream), "lcc_gemm_w8_bf16");
}
extern "C" int lcc_debug_mfma_probe(const void* A, const void* B, float* D, void* stream) {
  if (!A || !B || !D) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(mfma_probe((const bf16_t*)A, (const bf16_t*)B, D, (hipStream_t)stream), "lcc_debug_mfma_probe");
}
extern "C" int lcc_patchify_norm_u8(const uint8_t* frames, int layout, int T, int H, int W, const float mean255[3],
                                    const float std255[3], void* out, int ld, void* stream) {
  if (!frames || !out || !mean255 || !std255) return fail(LCC_ERR_ARG, "lcc_patchify_norm_u8: null pointer");
  OP_RET(patchify_norm_u8(frames, layout, T, H, W, mean255, std255, (bf16_t*)out, ld, (hipStream_t)stream), "lcc_patchify_norm_u8");
}
extern "C" int lcc_resize_bicubic_aa_u8(const uint8_t* src, int layout, int T, int Hin, int Win, uint8_t* dst, int Hout, int Wout,
                                        const int32_t* xmin, const int32_t* xsize, const float* wx, int kx, const int32_t* ymin,
                                        const int32_t* ysize, const float* wy, int ky, float* tmp, void* stream) {
  if (!src || !dst || !xmin || !xsize || !wx || !ymin || !ysize || !wy || !tmp) return fail(LCC_ERR_ARG, "lcc_resize_bicubic_aa_u8: null pointer");
  OP_RET(resize_bicubic_aa_u8(src, layout, T, Hin, Win, dst, Hout, Wout, xmin, xsize, wx, kx, ymin, ysize, wy, ky, tmp, (hipStream_t)stream),
         "lcc_resize_bicubic_aa_u8");
}
extern "C" int lcc_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream) {
  if (!in || !out) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(cast_f32_bf16(in, (bf16_t*)out, n, (hipStream_t)stream), "lcc_cast_f32_bf16");
}
extern "C" int lcc_layernorm_bf16(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps, void* stream) {
  if (!x || !w || !b || !y) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(layernorm_bf16((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, rows, dim, eps, (hipStream_t)stream), "lcc_layernorm_bf16");
}
extern "C" int lcc_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int dim, float eps, void* stream) {
  if (!x || !w || !y) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(rmsnorm_bf16((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rows, dim, eps, (hipStream_t)stream), "lcc_rmsnorm_bf16");
}
extern "C" int lcc_add_rmsnorm_bf16(void* h, const void* delta_bf16, const float* delta_partial, int nsplit, const void* w, void* y,
                                    int rows, int dim, float eps, void* stream) {
  if (!h || (w && !y)) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(add_rmsnorm_bf16((bf16_t*)h, (const bf16_t*)delta_bf16, delta_partial, nsplit, (const bf16_t*)w, (bf16_t*)y, rows, dim, eps,
                          (hipStream_t)stream), "lcc_add_rmsnorm_bf16");
}
extern "C" int lcc_swiglu_bf16(const void* gate, const void* up, void* out, int64_t n, void* stream) {
  if (!gate || !up || !out) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(swiglu_bf16((const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)out, n, (hipStream_t)stream), "lcc_swiglu_bf16");
}
extern "C" int lcc_vit_rope_vt_bf16(void* qkv, const float* cos, const float* sin, const int32_t* seg_of_patch, const int32_t* seg_start,
                                    const int32_t* seg_blk_start, void* vt, int P, int heads, int total_blocks, void* stream) {
  if (!qkv || !cos || !sin || !seg_of_patch || !seg_start || !seg_blk_start || !vt) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(vit_rope_vt_bf16((bf16_t*)qkv, cos, sin, seg_of_patch, seg_start, seg_blk_start, (bf16_t*)vt, P, heads, total_blocks,
                          (hipStream_t)stream), "lcc_vit_rope_vt_bf16");
}
extern "C" int lcc_attn_vit_bf16(const void* qkv, const void* vt, void* out, const int32_t* tile_seg, const int32_t* tile_q0,
                                 const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_tiles, int heads,
                                 int total_blocks, const int32_t* grp_seg, const int32_t* grp_q0, int n_groups, void* stream) {
  if (!qkv || !vt || !out || !tile_seg || !tile_q0 || !seg_start || !seg_len || !seg_blk_start) return fail(LCC_ERR_ARG, "null pointer");
  if (n_groups > 0 && (!grp_seg || !grp_q0)) return fail(LCC_ERR_ARG, "null group table");
  OP_RET(attn_vit_bf16((const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, tile_seg, tile_q0, seg_start, seg_len, seg_blk_start, n_tiles,
                       heads, total_blocks, grp_seg, grp_q0, n_groups, (hipStream_t)stream), "lcc_attn_vit_bf16");
}
extern "C" int lcc_attn_vit32_bf16(const void* qkv, const void* vt, void* out, const int32_t* grp_seg, const int32_t* grp_q0,
                                   const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_groups, int heads,
                                   int total_blocks, int group_rows, void* stream) {
  if (!qkv || !vt || !out || !grp_seg || !grp_q0 || !seg_start || !seg_len || !seg_blk_start) return fail(LCC_ERR_ARG, "null pointer");
  if (group_rows != 256 && group_rows != 128) return fail(LCC_ERR_ARG, "group_rows must be 256 or 128, got %d", group_rows);
  OP_RET(attn_vit32_launch((const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, grp_seg, grp_q0, seg_start, seg_len, seg_blk_start, n_groups, heads,
                           total_blocks, 1.4426950408889634f / sqrtf(80.f), (hipStream_t)stream, group_rows), "lcc_attn_vit32_bf16");
}
extern "C" int lcc_mrope_table(const int32_t* pos3, const float* inv_freq, int S, int sec_t, int sec_h, void* cos, void* sin, void* stream) {
  if (!pos3 || !inv_freq || !cos || !sin) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(mrope_table(pos3, inv_freq, S, sec_t, sec_h, (bf16_t*)cos, (bf16_t*)sin, (hipStream_t)stream), "lcc_mrope_table");
}
extern "C" int lcc_rope_kv_append_bf16(const void* qkv_bf16, const float* qkv_partial, int nsplit, const void* bias, const void* cos,
                                       const void* sin, const int32_t* tok_stream, const int32_t* tok_pos, const int32_t* kv_len,
                                       void* const* kv_base, lcc_kv_layout lay, int layer, void* q_out, int S, int n_q_heads, void* stream) {
  if ((!qkv_bf16 && !qkv_partial) || !cos || !sin || !tok_stream || !kv_base || !q_out) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(rope_kv_append_bf16((const bf16_t*)qkv_bf16, qkv_partial, nsplit, (const bf16_t*)bias, (const bf16_t*)cos, (const bf16_t*)sin,
                             tok_stream, tok_pos, kv_len, (bf16_t* const*)kv_base, to_lay(lay), layer, (bf16_t*)q_out, S, n_q_heads,
                             (hipStream_t)stream), "lcc_rope_kv_append_bf16");
}
extern "C" int lcc_attn_prefill_bf16(const void* q, void* out, const int32_t* tile_stream, const int32_t* tile_q0, const int32_t* tile_nq,
                                     const int32_t* tile_pos0, void* const* kv_base, lcc_kv_layout lay, int layer, int n_tiles,
                                     int n_q_heads, int tile_rows, int nsplit, int n_rows, float* ws_o, float* ws_ml, void* stream) {
  if (!q || !out || !tile_stream || !tile_q0 || !tile_nq || !tile_pos0 || !kv_base) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(attn_prefill_bf16((const bf16_t*)q, (bf16_t*)out, tile_stream, tile_q0, tile_nq, tile_pos0, (bf16_t* const*)kv_base, to_lay(lay),
                           layer, n_tiles, n_q_heads, tile_rows, nsplit, n_rows, ws_o, ws_ml, (hipStream_t)stream), "lcc_attn_prefill_bf16");
}
extern "C" int lcc_attn_decode_bf16(const void* q, void* out, const int32_t* slots, const int32_t* kv_len, void* const* kv_base,
                                    lcc_kv_layout lay, int layer, int B, int n_q_heads, int nsplit, float* ws_o, float* ws_ml, void* stream) {
  if (!q || !out || !slots || !kv_len || !kv_base || !ws_o || !ws_ml || nsplit < 1) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(attn_decode_bf16((const bf16_t*)q, (bf16_t*)out, slots, kv_len, (bf16_t* const*)kv_base, to_lay(lay), layer, B, n_q_heads, nsplit,
                          ws_o, ws_ml, (hipStream_t)stream), "lcc_attn_decode_bf16");
}
extern "C" int lcc_attn_decode_fused_bf16(const float* qkv_partial, int nsplit_qkv, const void* bias, const void* cos, const void* sin,
                                          const int32_t* slots, const int32_t* kv_len, void* const* kv_base, lcc_kv_layout lay, int layer,
                                          void* out, int B, int n_q_heads, int nsplit, float* ws_o, float* ws_ml, int32_t* counters,
                                          void* stream) {
  if (!qkv_partial || !bias || !cos || !sin || !slots || !kv_len || !kv_base || !out || !counters) return fail(LCC_ERR_ARG, "null pointer");
  if (nsplit > 1 && (!ws_o || !ws_ml)) return fail(LCC_ERR_ARG, "nsplit > 1 needs the partial workspaces");
  OP_RET(attn_decode_fused_bf16(qkv_partial, nsplit_qkv, (const bf16_t*)bias, (const bf16_t*)cos, (const bf16_t*)sin, slots, kv_len,
                                (bf16_t* const*)kv_base, to_lay(lay), layer, B, n_q_heads, nsplit, ws_o, ws_ml, counters, (bf16_t*)out,
                                (hipStream_t)stream), "lcc_attn_decode_fused_bf16");
}
// micro-benchmark of the decode attention chain of one layer, launched back to back `iters` times from C++ (a Python loop cannot
// issue 5-us kernels fast enough).  variant 0: rope_kv_append + attn_decode + combine (three launches, nsplit_sep key splits);
// 1: fused kernel + combine launch; 2: fused kernel with the in-launch merge.  Returns the average microseconds per chain.
extern "C" int lcc_debug_bench_attn_decode(int variant, int iters, const float* qkv_partial, int nsplit_qkv, const void* bias,
                                           const void* cos, const void* sin, const int32_t* slots, const int32_t* kv_len,
                                           void* const* kv_base, lcc_kv_layout lay, int layer, void* q_scratch, void* out, int B,
                                           int n_q_heads, int nsplit_sep, int nsplit_fused, float* ws_o, float* ws_ml,
                                           int32_t* counters, float* out_us, void* stream) {
  if (!qkv_partial || !bias || !cos || !sin || !slots || !kv_len || !kv_base || !q_scratch || !out || !ws_o || !ws_ml || !counters || !out_us)
    return fail(LCC_ERR_ARG, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  const KvLayout L = to_lay(lay);
  set_attn_fused_tail(variant == 2 ? 0 : 1);
  int rc = 0;
  for (int it = -3; it < iters && rc == 0; ++it) {
    if (it == 0) HIP_TRY(hipEventRecord(e0, st));
    if (variant == 0) {
      rc = rope_kv_append_bf16(nullptr, qkv_partial, nsplit_qkv, (const bf16_t*)bias, (const bf16_t*)cos, (const bf16_t*)sin, slots, nullptr,
                               kv_len, (bf16_t* const*)kv_base, L, layer, (bf16_t*)q_scratch, B, n_q_heads, st);
      if (rc == 0) rc = attn_decode_bf16((const bf16_t*)q_scratch, (bf16_t*)out, slots, kv_len, (bf16_t* const*)kv_base, L, layer, B, n_q_heads,
                                         nsplit_sep, ws_o, ws_ml, st);
    } else {
      rc = attn_decode_fused_bf16(qkv_partial, nsplit_qkv, (const bf16_t*)bias, (const bf16_t*)cos, (const bf16_t*)sin, slots, kv_len,
                                  (bf16_t* const*)kv_base, L, layer, B, n_q_heads, nsplit_fused, ws_o, ws_ml, counters, (bf16_t*)out, st);
    }
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *out_us = ms * 1000.f / (float)std::max(1, iters);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  set_attn_fused_tail(1);
  if (rc != 0) return fail(rc, "lcc_debug_bench_attn_decode: invalid arguments (%d)", rc);
  return check_launch("lcc_debug_bench_attn_decode");
}
// ---- what does a device-wide hand-off cost?  (the number the "one persistent launch per decode layer" design stands or falls with)
// mode 0: `iters` grid barriers inside ONE launch of `blocks` co-resident blocks (monotonic agent-scope counter: arrive = relaxed
//         fetch_add after a release fence, wait = acquire loads with s_sleep; every wait is BOUNDED -- a block that gives up counts
//         itself in *fails and leaves, so a mis-sized grid cannot hang the GPU);
// mode 1: `iters` dependent launches of a kernel of `blocks` blocks that touches one cache line per block (the kernel boundary);
// mode 2: mode 0 with a 16-KB streaming read per block between barriers (a barrier under memory load);
// mode 3 / 4: modes 0 / 2 with the XCD-hierarchical barrier of grid_sync.h (per-XCD arrival counters, one release fence per XCD leader,
//         per-XCD generation words) -- the form MI355X_MICROARCH.md prices at 4.1 us for 256 workgroups.
typedef __attribute__((ext_vector_type(4))) unsigned int bench_u32x4;
__global__ __launch_bounds__(256) void grid_barrier_bench_kernel(unsigned* counter, unsigned* fails, int iters, int nblocks, const bench_u32x4* stream_src,
                                                                 unsigned* sink) {
  unsigned acc = 0;
  for (int it = 1; it <= iters; ++it) {
    if (stream_src != nullptr) {
      const bench_u32x4 v = __builtin_nontemporal_load(stream_src + ((size_t)(blockIdx.x * 997 + it) % 4096) * 1024 + threadIdx.x * 4);
      acc += v.x ^ v.w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)it * (unsigned)nblocks;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 100000) { atomicAdd(fails, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// modes 3 / 4: the same loop on the XCD-hierarchical barrier of grid_sync.h (MI355X_MICROARCH.md "barrier-xcd")
__global__ __launch_bounds__(256) void grid_barrier_xcd_bench_kernel(GridSyncState* gs, int iters, const bench_u32x4* stream_src, unsigned* sink) {
  GridSync g = gs_begin(gs);
  if (!g.ok) return;
  unsigned acc = 0;
  for (int it = 1; it <= iters; ++it) {
    if (stream_src != nullptr) {
      const bench_u32x4 v = __builtin_nontemporal_load(stream_src + ((size_t)(blockIdx.x * 997 + it) % 4096) * 1024 + threadIdx.x * 4);
      acc += v.x ^ v.w;
    }
    if (!gs_barrier(g)) return;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void boundary_bench_kernel(unsigned* buf, int it) {
  if (threadIdx.x == 0) buf[blockIdx.x * 32] = buf[((blockIdx.x + 1) % gridDim.x) * 32] + (unsigned)it;
}
extern "C" int lcc_debug_bench_grid_barrier(int mode, int blocks, int iters, void* scratch, size_t scratch_bytes, float* out_us, int* out_fails,
                                            void* stream) {
  if (!scratch || !out_us || !out_fails || blocks < 1 || blocks > 1024 || iters < 1) return fail(LCC_ERR_ARG, "bad argument");
  if (mode < 0 || mode > 4) return fail(LCC_ERR_ARG, "mode must be 0..4");
  const bool streaming = mode == 2 || mode == 4;
  const size_t need = 4096 + (size_t)blocks * 128 + (streaming ? (size_t)4096 * 1024 * 16 : 0);
  if (scratch_bytes < need) return fail(LCC_ERR_ARG, "scratch too small: %zu bytes needed", need);
  hipStream_t st = (hipStream_t)stream;
  unsigned* ctr = (unsigned*)scratch;                       // [0] counter, [1] fails, [2] sink
  unsigned* buf = ctr + 1024;
  const bench_u32x4* src = streaming ? reinterpret_cast<const bench_u32x4*>((char*)scratch + 4096 + (size_t)blocks * 128) : nullptr;
  HIP_TRY(hipMemsetAsync(scratch, 0, 4096 + (size_t)blocks * 128, st));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  if (mode == 1) {
    for (int it = -8; it < iters; ++it) {
      if (it == 0) HIP_TRY(hipEventRecord(e0, st));
      boundary_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(buf, it);
    }
  } else if (mode >= 3) {
    static_assert(sizeof(GridSyncState) <= 3584, "GridSyncState must fit the first 3.5 KB of the scratch");
    GridSyncState* gs = (GridSyncState*)scratch;
    grid_barrier_xcd_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(gs, 8, src, ctr + 900);   // warm-up
    HIP_TRY(hipMemsetAsync(scratch, 0, 4096, st));
    HIP_TRY(hipEventRecord(e0, st));
    grid_barrier_xcd_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(gs, iters, src, ctr + 900);
  } else {
    grid_barrier_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(ctr, ctr + 1, 8, blocks, src, ctr + 2);   // warm-up
    HIP_TRY(hipMemsetAsync(scratch, 0, 64, st));
    HIP_TRY(hipEventRecord(e0, st));
    grid_barrier_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(ctr, ctr + 1, iters, blocks, src, ctr + 2);
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  unsigned f = 0;
  HIP_TRY(hipMemcpy(&f, mode >= 3 ? &((GridSyncState*)scratch)->fail[0] : ctr + 1, 4, hipMemcpyDeviceToHost));
  *out_us = ms * 1000.f / (float)iters;
  *out_fails = (int)f;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return check_launch("lcc_debug_bench_grid_barrier");
}
// ---- decode pipeline v2 operators (decode_v2.hip) ----
extern "C" int lcc_decode_step_begin(const int32_t* slots, const int32_t* cur_tok, const int32_t* done, uint32_t* seen, int words_per_stream,
                                     const void* embed_table, void* h, float* stats, int dim, const int32_t* pos, const float* inv_freq,
                                     void* cos, void* sin, int B, void* stream) {
  if (!slots || !cur_tok || !seen || !embed_table || !h || !stats || !pos || !inv_freq || !cos || !sin) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(decode_step_begin(slots, cur_tok, done, seen, words_per_stream, (const bf16_t*)embed_table, (bf16_t*)h, stats, dim, pos, inv_freq,
                           (bf16_t*)cos, (bf16_t*)sin, B, (hipStream_t)stream), "lcc_decode_step_begin");
}
extern "C" int lcc_dgemv_norm_linear(const void* W_packed, const float* wscale, const void* h, const float* stats, const void* norm_w, float eps,
                                     const void* bias, void* C, int ldc, int M, int N, int K, int swiglu, void* stream) {
  if (!W_packed || !h || !stats || !norm_w || !C) return fail(LCC_ERR_ARG, "null pointer");
  DgArgs a; a.W = (const bf16_t*)W_packed; a.wscale = wscale; a.M = M; a.N = N; a.K = K; a.H = (const bf16_t*)h; a.stats = stats; a.n_stat = K / 16;
  a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.bias = (const bf16_t*)bias; a.C = (bf16_t*)C; a.ldc = ldc;
  if (swiglu) OP_RET(dgemv_norm_swiglu(a, (hipStream_t)stream), "lcc_dgemv_norm_linear");
  OP_RET(dgemv_norm_bf16(a, (hipStream_t)stream), "lcc_dgemv_norm_linear");
}
extern "C" int lcc_dgemv_resid(const void* W_packed, const float* wscale, const void* x, int ldx, void* h, float* stats_out, int M, int N, int K,
                               void* stream) {
  if (!W_packed || !x || !h || !stats_out) return fail(LCC_ERR_ARG, "null pointer");
  DgArgs a; a.W = (const bf16_t*)W_packed; a.wscale = wscale; a.M = M; a.N = N; a.K = K; a.X = (const bf16_t*)x; a.ldx = ldx; a.Hres = (bf16_t*)h; a.stats_out = stats_out;
  OP_RET(dgemv_resid(a, (hipStream_t)stream), "lcc_dgemv_resid");
}
extern "C" int lcc_dgemv_qkv_rope(const void* W_dec_packed, const float* wscale, const void* h, const float* stats, const void* norm_w, float eps,
                                  const void* bias,
                                  const void* cos, const void* sin, const int32_t* tok_stream, const int32_t* kv_len, void* const* kv_base,
                                  lcc_kv_layout lay, int layer, void* q_out, int n_q_heads, int M, int K, void* stream) {
  if (!W_dec_packed || !h || !stats || !norm_w || !bias || !cos || !sin || !tok_stream || !kv_len || !kv_base || !q_out) return fail(LCC_ERR_ARG, "null pointer");
  DgArgs a; a.W = (const bf16_t*)W_dec_packed; a.wscale = wscale; a.M = M; a.N = (n_q_heads + 2 * lay.n_kv_heads) * 128; a.K = K; a.H = (const bf16_t*)h;
  a.stats = stats; a.n_stat = K / 16; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.bias = (const bf16_t*)bias; a.cs = (const bf16_t*)cos; a.sn = (const bf16_t*)sin;
  a.tok_stream = tok_stream; a.kv_len = kv_len; a.kv_base = (bf16_t* const*)kv_base; a.lay = to_lay(lay); a.layer = layer; a.q_out = (bf16_t*)q_out;
  a.n_q_heads = n_q_heads;
  OP_RET(dgemv_qkv_rope(a, (hipStream_t)stream), "lcc_dgemv_qkv_rope");
}
extern "C" int lcc_dgemv_down_qkv(const void* W_down_packed, const void* x, int ldx, void* h, float* stats, int K_down,
                                  const void* W_qkv_dec_packed, const void* norm_w, float eps, const void* bias, const void* cos, const void* sin,
                                  const int32_t* tok_stream, const int32_t* kv_len, void* const* kv_base, lcc_kv_layout lay, int layer,
                                  void* q_out, int n_q_heads, int M, int hidden, uint32_t* counter, uint32_t counter_before, uint32_t* err,
                                  void* stream) {
  if (!W_down_packed || !x || !h || !stats || !W_qkv_dec_packed || !norm_w || !bias || !cos || !sin || !tok_stream || !kv_len || !kv_base ||
      !q_out || !counter || !err) return fail(LCC_ERR_ARG, "null pointer");
  DgArgs d; d.W = (const bf16_t*)W_down_packed; d.M = M; d.N = hidden; d.K = K_down; d.X = (const bf16_t*)x; d.ldx = ldx; d.Hres = (bf16_t*)h;
  d.stats_out = stats;
  DgArgs a; a.W = (const bf16_t*)W_qkv_dec_packed; a.M = M; a.N = (n_q_heads + 2 * lay.n_kv_heads) * 128; a.K = hidden; a.H = (const bf16_t*)h;
  a.stats = stats; a.n_stat = hidden / 16; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.bias = (const bf16_t*)bias; a.cs = (const bf16_t*)cos;
  a.sn = (const bf16_t*)sin; a.tok_stream = tok_stream; a.kv_len = kv_len; a.kv_base = (bf16_t* const*)kv_base; a.lay = to_lay(lay); a.layer = layer;
  a.q_out = (bf16_t*)q_out; a.n_q_heads = n_q_heads;
  OP_RET(dgemv_down_qkv(d, a, counter, counter_before + (uint32_t)(hidden / 16), err, (hipStream_t)stream), "lcc_dgemv_down_qkv");
}
extern "C" int lcc_embed_gather_bf16(const int32_t* ids, const int32_t* indirect, const int32_t* vit_index, const void* table,
                                     const void* vit_rows, void* out, int S, int dim, void* stream) {
  if (!ids || !table || !out) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(embed_gather_bf16(ids, indirect, vit_index, (const bf16_t*)table, (const bf16_t*)vit_rows, (bf16_t*)out, S, dim, (hipStream_t)stream),
         "lcc_embed_gather_bf16");
}
extern "C" int lcc_seen_set(uint32_t* seen, int words_per_stream, const int32_t* ids, const int32_t* slot_of_id, int n, void* stream) {
  if (!seen || !ids || !slot_of_id) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(seen_set(seen, words_per_stream, ids, slot_of_id, n, 0, nullptr, (hipStream_t)stream), "lcc_seen_set");
}
extern "C" int lcc_sample_greedy(const void* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream, const int32_t* stream_slot,
                                 float repetition_penalty, int thr_token, int use_thr, float thr_value, int eos_token, int eos_token2,
                                 int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history, int hist_ld,
                                 int32_t* hist_col, float* scores_out, float* ws, void* stream) {
  if (!logits || !seen || !stream_slot || !out_tokens || (history && !hist_col)) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(sample_greedy((const bf16_t*)logits, ld, B, V, seen, words_per_stream, stream_slot, repetition_penalty, thr_token, use_thr,
                       thr_value, eos_token, eos_token2, suppress_eos, done, out_tokens, history, hist_ld, hist_col, scores_out, ws,
                       (hipStream_t)stream), "lcc_sample_greedy");
}
extern "C" int lcc_sample_topk_topp(const void* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream,
                                    const int32_t* stream_slot, float repetition_penalty, int thr_token, int use_thr, float thr_value,
                                    int eos_token, int eos_token2, int suppress_eos, int32_t* done, int32_t* out_tokens,
                                    int32_t* history, int hist_ld, int32_t* hist_col, float* scores_out, float temperature, int top_k,
                                    float top_p, uint64_t seed, uint32_t* rng_ctr, void* stream) {
  if (!logits || !seen || !stream_slot || !out_tokens || (history && !hist_col)) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(sample_topk_topp((const bf16_t*)logits, ld, B, V, seen, words_per_stream, stream_slot, repetition_penalty, thr_token, use_thr,
                          thr_value, eos_token, eos_token2, suppress_eos, done, out_tokens, history, hist_ld, hist_col, scores_out,
                          temperature, top_k, top_p, seed, rng_ctr, (hipStream_t)stream), "lcc_sample_topk_topp");
}
