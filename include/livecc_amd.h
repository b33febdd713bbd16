/* livecc_amd.h -- C-ABI of the MI355X-native LiveCC / Qwen2-VL streaming forward+generate hot path.
 *
 * Drop-in boundary.  The reference (showlab/livecc) is Python: demo/infer.py drives HuggingFace
 * `Qwen2VLForConditionalGeneration.generate` and patches operators into it through three plugin points.
 * Each entry point below names the reference interface it replaces (ref: = /root/reference, HF: = the
 * `transformers` package the reference depends on, Q2VL: = HF models/qwen2_vl/modeling_qwen2_vl.py).
 *
 * Conventions
 *   - plain C, no torch types.  All `dev` pointers are device (HBM) pointers on the current HIP device, borrowed:
 *     the library never frees or retains them except the buffers bound to an engine with lcc_engine_bind_*.
 *   - `stream` is a hipStream_t passed as void*; every launch goes to that stream; nothing synchronises the
 *     host except the functions documented as blocking.
 *   - bf16 tensors are raw uint16 storage; row-major; K (inner) dimension contiguous; pointers 16-byte aligned.
 *   - return value: 0 on success, a negative lcc_status on error (shape/dtype/alignment are validated up
 *     front, never UB on a bad shape); lcc_last_error() returns the message of the calling thread.
 *   - threads (SURVEY 8b "thread-compatible per stream-state object"; ref:demo/app.py:178 calls one model from five): the operator-level
 *     entry points are re-entrant (they touch only their arguments).  On ONE engine, lcc_llm_prefill / lcc_llm_decode / lcc_slot_* serialise
 *     on a mutex of that engine (one activation workspace, one meta ring, host mirrors of the slot lengths) and lcc_vit_encode on a second one
 *     (the vision tower has its own workspace + ring and may run beside the LLM calls on another stream); the caller still orders the DEVICE
 *     work, i.e. passes streams that respect the data flow.  The process-global lcc_debug_set_* knobs are launch-routing state: while any
 *     thread is inside lcc_vit_encode / lcc_llm_prefill / lcc_llm_decode they return LCC_ERR_STATE (the setters that return the previous value:
 *     -1) instead of changing the kernel family in the middle of a forward pass.
 *   - the same operator-level entry points are registered as PyTorch custom ops (torch.ops.livecc_amd.*) by livecc_amd/csrc/torch_ops.cpp, a
 *     separate library that only forwards to the symbols declared here.
 */
#ifndef LIVECC_AMD_H
#define LIVECC_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  LCC_OK = 0, LCC_ERR_ARG = -1, LCC_ERR_SHAPE = -2, LCC_ERR_ALIGN = -3, LCC_ERR_HIP = -4, LCC_ERR_STATE = -5
} lcc_status;

typedef enum {        /* GEMM epilogues, rounding points as HF's bf16 modules */
  LCC_EPI_NONE = 0,        /* nn.Linear                                      */
  LCC_EPI_QUICK_GELU = 1,  /* VisionMlp.fc1 + quick_gelu      Q2VL:293-301   */
  LCC_EPI_GELU_ERF = 2,    /* PatchMerger.mlp[0] + nn.GELU    Q2VL:283-287   */
  LCC_EPI_RESIDUAL = 3,    /* Linear + residual add           Q2VL:441-448   */
  LCC_EPI_SWIGLU = 4       /* gate/up (rows interleaved 16|16) + silu*mul  Q2VL:453-466 */
} lcc_epilogue;

const char* lcc_last_error(void);
const char* lcc_version(void);
int lcc_device_info(int* cu_count, size_t* hbm_bytes, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * Operator level -- the slots of the reference's own operator plugins
 *   (1) liger `apply_liger_kernel_to_qwen2_vl()`       ref:demo/infer.py:2-3   (RMSNorm/LayerNorm/SwiGLU/M-RoPE)
 *   (2) HF AttentionInterface (`attn_implementation=`) ref:demo/infer.py:46    HF:modeling_utils.py:5093-5131
 *   (3) HF Cache.update                                HF:cache_utils.py:127-146
 *   (4) HF LogitsProcessor / sampler                   ref:demo/infer.py:10-23 HF:generation/utils.py:2894-2925
 * ------------------------------------------------------------------------------------------------ */

/* nn.Linear / Conv3d-as-GEMM:  C[M,N] = A[M,K] * W[N,K]^T (+bias)(+epilogue).  K%8==0, N%16==0.
 * w_layout 0: W row-major [N,K] (nn.Linear);  1: W pre-packed in MFMA fragment order [N/16][ceil(K/32)][4][16][8]
 * (element (n,k) at ((n/16*ceil(K/32) + k/32)*4 + (k%32)/8)*128 + (n%16)*8 + k%8, K zero-padded to 32): every wave load
 * of the weight-streaming path is then one contiguous KB.  The engine stores all Linear weights packed.
 * M<=16 takes the HBM-bound skinny path; `partial` (fp32 [nsplit][M][N]) returns raw split-K slabs instead of C (both paths;
 * epilogue must be NONE, nsplit <= 8). */
int lcc_gemm_bf16(const void* A, int lda, const void* W, int ldw, int w_layout, const void* bias, const void* residual, int ldr,
                  void* C, int ldc, int M, int N, int K, int epilogue, float* partial, int nsplit, void* stream);
/* tuning knob of the skinny kernel (0: 2 chunks single stage, 1: 1-chunk two-stage pipeline, 2: 2-chunk two-stage) */
int lcc_debug_set_gemv_variant(int variant);
/* tiled kernels: 0 = 4-wave register-staged double-buffered LDS, 1 = 4-wave LDS-DMA (global_load_lds) 3-stage ring, 2 = scored per
 * shape (default: 8-wave 256x256 / 128x256 LDS-DMA kernel where its grid fills the chip, else the 4-wave kernels), 3 / 4 = force the
 * 8-wave kernel with 256 / 128 rows where eligible, 5 / 6 = the same with the compiler's fragment-read schedule, 7 = 2 without the
 * 8-wave kernel, 8 = force the "tall" kernel (one block row covers all of M <= 448, 448x160 tiles) wherever it is legal; the default
 * takes it when 256 < M <= 448 and ceil(N/160) fills 75-100 % of one round of the chip (LiveCC-7B gate/up of a streaming chunk);
 * 13 = the 192-row 8-wave tile wherever eligible; 14 = row tiles of variable height (256 / 272 / 288 rows: ceil(M/16) row fragments dealt
 * out over floor(ceil(M/16)/16) tiles, no ragged last tile; round 5) wherever legal, else as 3 -- the default picks them by score (e.g. the
 * gate/up GEMM of 8 co-scheduled chunks: 12 x 148 tiles instead of 13 x 148); 15 = the SMALL variable-height class (row tiles of 128 / 144
 * rows) for split-K slabs of 64 < M <= 448 wherever legal, else as 4 -- the default picks it by score for the o / down / q|k|v projections
 * of one streaming chunk (M = 386: 3 row tiles instead of 4, 6 / 6 / 4 splits; LCC_GEMM_VH_SMALL=0 restores the 128-row tiles and their
 * 4 / 4 / 3 splits).  Every tile shape produces the same bits for the same split count. */
int lcc_debug_set_gemm_variant(int variant);
/* Host logic only (no launch; works without a GPU): which kernel family serves a packed-weight GEMM of this shape under the current
 * variant -- *tile_rows = 16 weight-streaming GEMV, 448 tall, 272 / 144 variable-height tiles (big / small class), 256 / 192 / 128 the
 * 8-wave tile of that height, 64 a 4-wave tile kernel; nsplit > 0 asks for the split-K slab form -- and *engine_splits = the split count
 * the engine's prefill asks for at this shape (what csrc/engine_llm.hip passes as nsplit for the o / down / q|k|v projections).
 * w_fp8 != 0: *tile_rows = 0 (fp8 weights route inside their own entry point), *engine_splits as for fp8 weights. */
int lcc_debug_gemm_plan(int M, int N, int K, int epilogue, int nsplit, int w_fp8, int32_t* tile_rows, int32_t* engine_splits);
/* attention: 0 = per-wave kernels (operands straight from L2); 1 = prefill shares K/V tiles through an LDS-DMA ring,
 * ViT per-wave; 2 = LDS-shared for both; 3 (default) = 2 with the LLM prefill on the 32x32x16 MFMAs (csrc/attn32.hip; applies to
 * calls with tile_rows >= 32; the engine then builds tiles of 32 or lcc_debug_attn_tile_rows() rows) */
int lcc_debug_set_attn_variant(int variant);
/* The tallest query tile the engine builds for the LLM prefill attention under the current variant (it takes it when that saves a
 * round of blocks on the chip or the grid is below one round, else 32 rows), and the largest `tile_rows` that lcc_attn_prefill_bf16
 * accepts besides 16 and 32: under variant 3 a block of the 32x32x16 kernel packs (row, head) pairs into
 * 8 x 32 columns (4 x 32 at <= 4 query heads per KV head), i.e. 256 / G rows -- 36 for Qwen2-VL-7B (28 / 4 heads), 32 for the 72B
 * (64 / 8); 32 otherwise.  Host logic only (no launch, no GPU); < 0 = LCC_ERR_ARG for head counts that do not divide. */
int lcc_debug_attn_tile_rows(int n_q_heads, int n_kv_heads);
/* Host-side plan of ONE LLM prefill attention launch under attention variant 3 (the 32x32x16 kernel), as lcc_llm_prefill makes it: query-tile
 * height (32 or lcc_debug_attn_tile_rows) and key-split count for `n_streams` streams with n_new[b] new rows each against at most `max_kv` keys
 * on a device of `cu_count` compute units.  One 8-wave block per CU and (tile, KV head, split): cost = rounds of blocks x keys per block
 * + a per-split term (fp32 partials + merge); splits only for calls of <= 1,024 rows with >= 8 key tiles per split; the tall tile only where
 * strictly cheaper.  At 28 / 4 heads, 256 CUs, 6.5k keys: one 386-row chunk -> (36, 5); two chunks -> (32, 2); five -> (36, 1); eight ->
 * (32, 1); eight 1,131-row first turns -> (36, 1).  *tile_rows = *splits = 0 when another kernel family serves the call (variant != 3, or more than 8
 * query heads per KV head).  Host logic only (no launch, no GPU). */
int lcc_debug_attn_plan(const int32_t* n_new, int n_streams, int max_kv, int n_q_heads, int n_kv_heads, int cu_count, int32_t* tile_rows,
                        int32_t* splits);
/* 1: on the batch-1 decode path the consumers of a split-K GEMV (bias + M-RoPE + KV append; residual add + RMSNorm) run as the
 * TAIL of that GEMV in its last-arriving block (agent-scope release/acquire); 0 (default, measured faster): separate kernels */
int lcc_debug_set_fused_tails(int on);
/* decode launch sequence of the engine: 1 (default) = pipeline v2 where eligible (bf16 weights, the row-permuted decode copy
 * "llm.<i>.qkv_w_dec" of every q|k|v weight set): RMSNorm / bias + M-RoPE + KV append / residual add run inside the weight-streaming
 * GEMVs, 6 launches per layer, for decode batches of ONE or TWO streams (lcc_llm_decode: `n_streams <= 2`; measured: 640 vs 655 tokens/s
 * at four streams, so larger batches keep the round-1 sequence; the MR = 4 forms of the kernels are reachable only through the
 * lcc_dgemv_* operators below); bf16 weights, and since round 3 fp8 weights that carry the permuted decode copy + its scales;
 * 0 = the round-1 sequence of 9 launches per layer.  Merging the attention key splits inside the o_proj GEMV as well (5 launches) was
 * measured slower: 4-wave attention blocks 10.0 us + o_proj 9.8 us vs 7.3 + 4.9 + 6.8 us. */
int lcc_debug_set_decode_path(int path);
/* 1: pipeline v2 launches down_proj of layer l and the q/k/v GEMV of layer l+1 as ONE chained launch (5 launches per layer; default 0:
 * measured slower on MI355X, 3.1-3.2 vs 2.99 ms per decode step -- see csrc/engine_llm.hip):
 * the q/k/v blocks are resident next to the down_proj blocks, request their weights at once and then wait -- bounded -- for the
 * down_proj blocks to publish the residual stream (write-through stores + a monotonic counter, Guideline 16 R1), so the HBM stream
 * does not drain at the hand-off.  Used only when both grids fit the chip at once (LiveCC-7B: 224 + 288 blocks of 512 threads = 2 per
 * CU) and for <= 2 streams; 0 (default) = separate launches.  A hand-off that times out fails the call (lcc_slot_read_tokens) and disables it. */
int lcc_debug_set_decode_chain(int on);
/* Block shape of the o_proj / down_proj decode GEMV (lcc_dgemv_resid): 0 = 8 waves per block, 1 (default) = 16 waves for bf16 weights with
 * K >= 8192 (the down projection: half the dependent chain of load stages per wave), 2 = 16 waves for every call.  Returns the old mode.
 * The K split across the waves of a block -- and so the fp32 summation order -- differs between the two shapes. */
int lcc_debug_set_resid_waves(int mode);
/* Largest M (16..64, default 64) served by the weight-streaming GEMV kernels of lcc_gemm_bf16 and by the engine's decode step: with 17..64
 * rows (decode batches of 17-64 streams) a weight fragment is multiplied with 2-4 activation fragments, so every weight byte is still
 * read once per step.  16 restores the round-3 routing of such batches through the 64-row GEMM tiles.  Returns the old value. */
int lcc_debug_set_skinny_rows(int rows);
/* Vision tower (lcc_vit_encode): 1 (default; LCC_VIT_FUSED_QKV=0 in the environment starts with 0) = the q|k|v projection of every block
 * applies the 2-D RoPE to q, k and writes V blocked-transposed in its own epilogue (needs the `vit.<i>.qkv_w_rope` / `qkv_b_rope` copies of
 * the weights in the rotation-pair row order, head_dim 80, more than 64 patches and K % 64 == 0; one launch when E % 128 == 0, else a q|k
 * launch + a V launch); 0 = projection + lcc_vit_rope_vt_bf16 as separate launches.  Both forms produce the same bits (HF Q2VL:225-248,
 * 342-368).  Returns the old value. */
int lcc_debug_set_vit_fused_qkv(int on);
int lcc_gemv_num_splits(int N, int K);
/* nn.Linear with fp8 (OCP e4m3) weights, the 72B single-GPU path (BASELINE.json configs[4]): W8 = bytes in the PACKED8 order
 * [N/16][K/64][4 g][16 rows][16 k] (lane (g,row) owns 16 consecutive k), wscale = fp32 [N] per-output-row scale:
 * y[m][n] = epilogue(wscale[n] * sum_k x[m][k] * q[n][k] (+ bias[n])).  M <= 16: weight-streaming GEMV (e4m3 -> bf16 exactly
 * in registers, bf16 MFMA, fp32 accumulate).  M > 16: the 8-wave GEMM stages the fp8 fragments themselves (half the L2 -> LDS bytes) and
 * expands them to bf16 after the LDS read, the row scale in its epilogue; only shapes with too few 256-column tiles for that kernel fall
 * back to an exact dequantisation into dq_scratch (N*K bf16, caller-provided) + the bf16 GEMM.  K % 64 == 0, N % 16 == 0.  partial /
 * nsplit as in lcc_gemm_bf16. */
int lcc_gemm_w8_bf16(const void* A, int lda, const void* W8, const float* wscale, const void* bias, const void* residual, int ldr,
                     void* C, int ldc, int M, int N, int K, int epilogue, float* partial, int nsplit, void* dq_scratch, void* stream);
/* self-test of the MFMA fragment maps: D[16,16] fp32 = A[16,32] bf16 * B[32,16] bf16 on one wave */
int lcc_debug_mfma_probe(const void* A, const void* B, float* D, void* stream);

/* HF video processor rescale+normalise+patchify (HF:models/qwen2_vl/video_processing_qwen2_vl.py:236-274,
 * HF:image_processing_backends.py:307-333).  frames: uint8, layout 0=[T,H,W,3], 1=[T,3,H,W]; out bf16 [P,ld]. */
int lcc_patchify_norm_u8(const uint8_t* frames, int layout, int T, int H, int W, const float mean255[3],
                         const float std255[3], void* out, int ld, void* stream);
/* Frame resize in front of the hot path: ref livecc_utils/video_process_patch.py:150-155 =
 * torchvision.transforms.functional.resize(uint8 clip, [Hout, Wout], BICUBIC, antialias=True), i.e. ATen's float32 separable
 * antialias bicubic (width pass, then height pass, taps accumulated in order with FMAs), clamp to [0,255], round half to even.
 * src: uint8 frames, layout 0 = [T,Hin,Win,3], 1 = [T,3,Hin,Win]; dst: uint8 [T,3,Hout,Wout].  Tap tables per output index
 * (first source index, tap count, fp32 weights) are computed by the caller with ATen's arithmetic
 * (livecc_amd/resize.py:aa_bicubic_taps): wx is TAP-MAJOR [kx][Wout] (coalesced across the lanes of the width pass), wy is
 * [Hout][ky]; tmp = T*3*Hin*Wout floats.  Bit-identical to the reference's CPU result. */
int lcc_resize_bicubic_aa_u8(const uint8_t* src, int layout, int T, int Hin, int Win, uint8_t* dst, int Hout, int Wout,
                             const int32_t* xmin, const int32_t* xsize, const float* wx, int kx, const int32_t* ymin,
                             const int32_t* ysize, const float* wy, int ky, float* tmp, void* stream);
int lcc_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream);  /* pixel_values.type(bf16) Q2VL:1044 */

int lcc_layernorm_bf16(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps, void* stream); /* Q2VL:428-429,281 */
int lcc_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int dim, float eps, void* stream);                 /* Q2VL:96-110 */
/* h += delta (bf16 [rows,dim] or fp32 split-K slabs), then y = rmsnorm(h)*w (w==NULL: add only)   Q2VL:594-612 */
int lcc_add_rmsnorm_bf16(void* h, const void* delta_bf16, const float* delta_partial, int nsplit, const void* w,
                         void* y, int rows, int dim, float eps, void* stream);
int lcc_swiglu_bf16(const void* gate, const void* up, void* out, int64_t n, void* stream);                                /* Q2VL:465 */

/* ViT 2-D RoPE on q,k in place in qkv [P,3E] + V written blocked-transposed (Q2VL:225-248) */
int lcc_vit_rope_vt_bf16(void* qkv, const float* cos, const float* sin, const int32_t* seg_of_patch,
                         const int32_t* seg_start, const int32_t* seg_blk_start, void* vt, int P, int heads,
                         int total_blocks, void* stream);
/* VisionAttention core (Q2VL:375-417): non-causal attention inside each temporal slice (segment).  Work tables:
 * 32-row query tiles (tile_seg/tile_q0) and 128-row groups (grp_seg/grp_q0; 4 tiles of ONE segment per block, K/V tiles
 * shared through LDS).  n_groups == 0 selects the per-wave kernel. */
int lcc_attn_vit_bf16(const void* qkv, const void* vt, void* out, const int32_t* tile_seg, const int32_t* tile_q0,
                      const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_tiles,
                      int heads, int total_blocks, const int32_t* grp_seg, const int32_t* grp_q0, int n_groups, void* stream);

/* The same attention on 32x32x16 MFMAs (csrc/attn32.hip; attention variant 3 = the engine's default): grp_seg / grp_q0 describe groups
 * of `group_rows` query rows of ONE segment each: 256 (8 waves x 32 rows per workgroup) or 128 (4 waves: twice the workgroups, for
 * inputs whose 256-row groups would not fill the chip -- one 2-frame streaming chunk is 6 x 16 heads = 96 groups); d = 80 only. */
int lcc_attn_vit32_bf16(const void* qkv, const void* vt, void* out, const int32_t* grp_seg, const int32_t* grp_q0,
                        const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_groups, int heads,
                        int total_blocks, int group_rows, void* stream);

/* M-RoPE tables (Q2VL:156-169) and apply + in-place KV append (Q2VL:180-222 + HF:cache_utils.py:127-146) */
int lcc_mrope_table(const int32_t* pos3, const float* inv_freq, int S, int sec_t, int sec_h, void* cos, void* sin, void* stream);
typedef struct { int n_layers, n_kv_heads, lmax, head_dim; } lcc_kv_layout;
/* tok_stream[s] = stream slot of row s; tok_pos[s] = its cache index, or NULL -> kv_len[slot] (decode) */
int lcc_rope_kv_append_bf16(const void* qkv_bf16, const float* qkv_partial, int nsplit, const void* bias,
                            const void* cos, const void* sin, const int32_t* tok_stream, const int32_t* tok_pos,
                            const int32_t* kv_len, void* const* kv_base, lcc_kv_layout lay, int layer, void* q_out, int S,
                            int n_q_heads, void* stream);
/* Qwen2VLAttention core (Q2VL:537-556): causal GQA over the in-place cache.  A tile = up to tile_rows (16, 32, or
 * under attention variant 3 anything up to lcc_debug_attn_tile_rows()) consecutive new rows of one stream: slot, first row in q, valid
 * rows, cache index of the first row.  LCC_ERR_SHAPE for any other tile_rows. */
int lcc_attn_prefill_bf16(const void* q, void* out, const int32_t* tile_stream, const int32_t* tile_q0,
                          const int32_t* tile_nq, const int32_t* tile_pos0, void* const* kv_base, lcc_kv_layout lay,
                          int layer, int n_tiles, int n_q_heads, int tile_rows, int nsplit, int n_rows, float* ws_o,
                          float* ws_ml, void* stream);  /* nsplit > 1: keys split over blockIdx.z, fp32 partials in ws_o
                          [n_rows*Hq*nsplit*128] / ws_ml [n_rows*Hq*nsplit*2], merged by a combine kernel */
/* decode: row b belongs to slot slots[b]; attends to kv_len[slot]+1 keys (its own K/V already appended) */
int lcc_attn_decode_bf16(const void* q, void* out, const int32_t* slots, const int32_t* kv_len, void* const* kv_base,
                         lcc_kv_layout lay, int layer, int B, int n_q_heads, int nsplit, float* ws_o, float* ws_ml, void* stream);
/* fused decode step of one layer: [q/k/v bias + M-RoPE (Q2VL:180-222) + Cache.update (cache_utils.py:127-146)] + attention
 * (Q2VL:537-556) + split merge in ONE launch.  qkv_partial = the fp32 split-K slabs [nsplit_qkv][B][(Hq+2Hkv)*128] of the
 * q/k/v Linear (lcc_gemm_bf16 with `partial`); cos/sin = lcc_mrope_table rows of the B new tokens; the new K/V go to cache
 * index kv_len[slots[b]] (kv_len itself is not advanced).  counters: Hkv*B int32, zero before the first call (left at zero).
 * ws_o [B*Hkv*nsplit*16*128] / ws_ml [B*Hkv*nsplit*16*2] floats (used when nsplit > 1).  Same result as
 * lcc_rope_kv_append_bf16 + lcc_attn_decode_bf16 up to the fp32 merge order. */
int lcc_attn_decode_fused_bf16(const float* qkv_partial, int nsplit_qkv, const void* bias, const void* cos, const void* sin,
                               const int32_t* slots, const int32_t* kv_len, void* const* kv_base, lcc_kv_layout lay, int layer,
                               void* out, int B, int n_q_heads, int nsplit, float* ws_o, float* ws_ml, int32_t* counters,
                               void* stream);
/* bench only: `iters` back-to-back launches of one layer's decode attention chain from C++; variant 0 = rope_kv_append +
 * attn_decode + combine, 1 = fused + combine launch, 2 = fused with in-launch merge; *out_us = average microseconds per chain */
int lcc_debug_bench_attn_decode(int variant, int iters, const float* qkv_partial, int nsplit_qkv, const void* bias, const void* cos,
                                const void* sin, const int32_t* slots, const int32_t* kv_len, void* const* kv_base, lcc_kv_layout lay,
                                int layer, void* q_scratch, void* out, int B, int n_q_heads, int nsplit_sep, int nsplit_fused,
                                float* ws_o, float* ws_ml, int32_t* counters, float* out_us, void* stream);
/* Device-wide hand-off cost (design input for a persistent decode-layer kernel): mode 0 = `iters` grid barriers inside one launch of
 * `blocks` co-resident blocks (every wait bounded: a block that gives up is counted in *out_fails), 1 = `iters` dependent launches of a
 * trivial `blocks`-block kernel, 2 = mode 0 with a 16-KB streaming read per block between barriers, 3 / 4 = modes 0 / 2 on the
 * XCD-hierarchical barrier (csrc/grid_sync.h: per-XCD arrival counters + one release fence per XCD leader).  Microseconds per hand-off. */
int lcc_debug_bench_grid_barrier(int mode, int blocks, int iters, void* scratch, size_t scratch_bytes, float* out_us, int* out_fails,
                                 void* stream);
/* Launch counters (tests assert WHICH kernel served a call): out[i] for i < n (n <= 16), optionally reset afterwards.  Slots:
 * 0 per-wave decode attention, 1 decode split-merge launches, 2 fused decode attention (+ combine launch), 3 fused decode attention with
 * the in-launch merge, 4 split-K GEMV with a fused consumer tail, 5 decode-pipeline-v2 GEMVs, 6 prefill attention on 32x32x16 MFMAs
 * (attn32.hip), 7 LDS-shared prefill attention, 8 per-wave prefill attention, 9 prefill split-merge launches, 10 / 11 = key splits of the
 * most recent decode / prefill attention launch (values, not counts), 12 tall-tile GEMM, 13 32x32x16 vision attention, 14 variable-height
 * GEMM tiles (gemm_vh_kernel), 15 vision q|k|v projection with the RoPE / V-transpose epilogue. */
int lcc_debug_launch_counts(int64_t* out, int n, int reset);
int lcc_debug_set_fused_attn(int mode); /* bit 0 (default on): engine decode uses the fused kernel for batches of >= 16 (stream,
                                          KV head) pairs; bit 2: for every batch; bit 1: key splits merged in the same launch by the
                                          last-arriving block instead of a combine launch (default off) */

/* ---- decode pipeline v2 (csrc/decode_v2.hip) as operators: one new token per stream (M <= 4 rows here; the ENGINE uses them for batches
 * of one or two streams), the elementwise stages of the decoder layer
 * run INSIDE the weight-streaming GEMVs.  Residual stream h [M,K] bf16 + `stats` fp32 [M, K/16]: per-16-channel sums of squares of h,
 * the input of the RMSNorm prologue (Q2VL:96-110) of the next GEMV; K % 64 == 0, K <= 8192, M*K <= 16384.  W = MFMA-fragment-packed bf16
 * (`wscale` NULL) or, round 3, OCP e4m3 bytes in the PACKED8 order of lcc_gemm_w8_bf16 with `wscale` = fp32 scale per STORED row
 * (for the row-permuted q|k|v decode copy: the permuted scales): e4m3 -> bf16 exactly in registers, bf16 MFMA, fp32 accumulate,
 * scale applied to the fp32 sum before the epilogue -- the fp8 weight path (BASELINE configs[4]) on the 6-launch decode layer. */
/* step prologue (was seen_set + embed_gather + mrope_table_decode): marks cur_tok[slot] in the slot's seen-id bitmap (unless done),
 * h[b] = embedding row (Q2VL:1160) + its stats, cos/sin[b] = M-RoPE row of pos[slot] (Q2VL:1349-1351: one position on all axes) */
int lcc_decode_step_begin(const int32_t* slots, const int32_t* cur_tok, const int32_t* done, uint32_t* seen, int words_per_stream,
                          const void* embed_table, void* h, float* stats, int dim, const int32_t* pos, const float* inv_freq,
                          void* cos, void* sin, int B, void* stream);
/* C = Linear(RMSNorm(h) * norm_w) (+bias); swiglu = 1: W rows interleaved [16 gate | 16 up], C[:, N/2] = silu(g) * u (Q2VL:453-466) */
int lcc_dgemv_norm_linear(const void* W_packed, const float* wscale, const void* h, const float* stats, const void* norm_w, float eps,
                          const void* bias, void* C, int ldc, int M, int N, int K, int swiglu, void* stream);
/* h += Linear(x) in place (HF rounding: Linear output -> bf16, residual add -> bf16, Q2VL:594-612); stats_out [M, N/16] of the new h.
 * o_proj and down_proj: no inter-block K split, 8 waves per 16-row block. */
int lcc_dgemv_resid(const void* W_packed, const float* wscale, const void* x, int ldx, void* h, float* stats_out, int M, int N, int K,
                    void* stream);
/* q|k|v Linear of RMSNorm(h) + bias + M-RoPE (Q2VL:180-222) + in-place KV append at kv_len[tok_stream[m]] (cache_utils.py:127-146),
 * rotated q -> q_out [M, Hq*128].  W_dec = the q|k|v weight with rows permuted inside every 128-row head so that each 16-row tile
 * holds 8 channels and their rotation partners: stored row j*16 + half*8 + i = logical row half*64 + j*8 + i (weights.py). */
int lcc_dgemv_qkv_rope(const void* W_dec_packed, const float* wscale, const void* h, const float* stats, const void* norm_w, float eps,
                       const void* bias,
                       const void* cos, const void* sin, const int32_t* tok_stream, const int32_t* kv_len, void* const* kv_base,
                       lcc_kv_layout lay, int layer, void* q_out, int n_q_heads, int M, int K, void* stream);
/* the chained launch as an operator (tests): h += Linear_down(x) with its tile statistics, then q|k|v of RMSNorm(h) as
 * lcc_dgemv_qkv_rope -- `counter` / `err`: device uint32 words, *counter counts arrivals monotonically (pass its value BEFORE the call
 * as `counter_before`), *err is set if a consumer block gives up.  LCC_ERR_STATE when the two grids do not fit the chip at once. */
int lcc_dgemv_down_qkv(const void* W_down_packed, const void* x, int ldx, void* h, float* stats, int K_down,
                       const void* W_qkv_dec_packed, const void* norm_w, float eps, const void* bias, const void* cos, const void* sin,
                       const int32_t* tok_stream, const int32_t* kv_len, void* const* kv_base, lcc_kv_layout lay, int layer, void* q_out,
                       int n_q_heads, int M, int hidden, uint32_t* counter, uint32_t counter_before, uint32_t* err, void* stream);
int lcc_embed_gather_bf16(const int32_t* ids, const int32_t* indirect, const int32_t* vit_index, const void* table,
                          const void* vit_rows, void* out, int S, int dim, void* stream);             /* Q2VL:1159-1176 */
int lcc_seen_set(uint32_t* seen, int words_per_stream, const int32_t* ids, const int32_t* slot_of_id, int n, void* stream);
/* RepetitionPenalty -> ThresholdLogitsProcessor -> argmax  (HF:generation/logits_process.py, ref:demo/infer.py:10-23).
 * eos_token / eos_token2: the (up to two) EOS ids of generation_config.json (<|im_end|>, <|endoftext|>; -1 = unused): both are
 * masked while suppress_eos bit 0 is set (MinNewTokensLength) and either one sets the slot's done flag -- unless suppress_eos bit 1 is
 * set (the engine's teacher forcing: the forced stream decides where a slot ends; scores stay HF's, EOS unmasked). */
int lcc_sample_greedy(const void* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream,
                      const int32_t* stream_slot, float repetition_penalty, int thr_token, int use_thr, float thr_value,
                      int eos_token, int eos_token2, int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history, int hist_ld,
                      int32_t* hist_col, float* scores_out, float* ws /* optional scratch B*256 floats: two-stage path */,
                      void* stream);
/* do_sample=True (ref:demo/infer.py:68 default of live_cc; HF:generation/utils.py:1296-1318, 2920-2925): the processors above,
 * then TemperatureLogitsWarper (scores / temperature), TopKLogitsWarper (top_k 0 = off; keeps scores >= the k-th largest),
 * TopPLogitsWarper (top_p 1 = off; removes the ascending tail whose cumulative softmax mass is <= 1 - top_p, the largest score
 * is always kept), softmax, one multinomial draw per stream.  scores_out (optional, fp32 [B,V]) receives the processed scores
 * HF would hand to softmax (-inf where removed).  Randomness: Philox4x32-10 keyed by `seed`, counter = (rng_ctr[slot]++, slot):
 * a per-slot draw counter in device memory (uint32 [n_slots], zero for a fresh stream) -- reproducible for a given seed,
 * independent of batching.  Deterministic: no floating-point atomics (masses are accumulated as 2^40-scaled integers). */
int lcc_sample_topk_topp(const void* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream,
                         const int32_t* stream_slot, float repetition_penalty, int thr_token, int use_thr, float thr_value,
                         int eos_token, int eos_token2, int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history,
                         int hist_ld, int32_t* hist_col, float* scores_out, float temperature, int top_k, float top_p,
                         uint64_t seed, uint32_t* rng_ctr, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Model level -- what `model.generate(**inputs, past_key_values=...)` (ref:demo/infer.py:165-172) executes:
 * ViT (Q2VL:700-729) -> embed+scatter -> 28x decoder layer -> lm_head -> processors -> sample, with the
 * per-stream state (KV, length, rope_delta position, seen-id bitmap, generated ids) resident on the device.
 * ------------------------------------------------------------------------------------------------ */
typedef struct lcc_engine lcc_engine;

typedef struct {
  int vocab_size, hidden_size, intermediate_size, n_layers, n_q_heads, n_kv_heads, head_dim;
  float rms_eps;
  int mrope_sec_t, mrope_sec_h, mrope_sec_w;
  int vit_depth, vit_embed, vit_heads, vit_mlp, patch_dim, merge;
  int llm_fp8;   /* 1: the LLM Linear weights (q/k/v, o, gate/up, down, lm_head) are fp8 e4m3 (PACKED8 order) + fp32 row scales
                    set as "<name>.scale"; the workspace then includes a bf16 dequantisation scratch for prefill */
} lcc_model_config;

typedef struct {
  int max_slots;        /* concurrent video streams resident on this GPU                      */
  int max_kv_len;       /* KV capacity per stream (multiple of 32), e.g. 32768                 */
  int max_new_rows;     /* largest number of new LLM tokens in one prefill call (all streams)  */
  int max_patches;      /* largest number of ViT patches in one encode call                    */
  int max_history;      /* generated tokens kept on device per generate call                   */
} lcc_engine_limits;

lcc_engine* lcc_engine_create(const lcc_model_config* cfg, const lcc_engine_limits* lim);
void lcc_engine_destroy(lcc_engine* e);
/* bytes the caller must provide (allocated by PyTorch so that torch.distributed can broadcast into them) */
size_t lcc_engine_workspace_bytes(const lcc_engine* e);
size_t lcc_engine_state_bytes(const lcc_engine* e);
size_t lcc_engine_kv_bytes_per_slot(const lcc_engine* e);
size_t lcc_engine_meta_bytes(const lcc_engine* e);
int lcc_engine_bind_buffers(lcc_engine* e, void* workspace_dev, size_t ws_bytes, void* state_dev, size_t state_bytes,
                            void* meta_dev, void* meta_host_pinned, size_t meta_bytes);
int lcc_engine_bind_kv(lcc_engine* e, int slot, void* kv_dev, size_t bytes);   /* zero-initialised arena */
/* optional: a PRIVATE activation workspace and meta ring (device + pinned host) for lcc_vit_encode.  Without them the ViT shares the
 * engine workspace with the LLM (one stream).  With them lcc_vit_encode may be issued on a second HIP stream and overlap LLM calls:
 * the vision tower of the NEXT turn's frames (compute-bound MFMA work) runs under the current turn's decode steps (HBM-bound weight
 * streaming) -- the caller orders consumers behind it with events. */
size_t lcc_engine_vit_workspace_bytes(const lcc_engine* e);
size_t lcc_engine_vit_meta_bytes(const lcc_engine* e);
int lcc_engine_bind_vit_buffers(lcc_engine* e, void* workspace_dev, size_t ws_bytes, void* meta_dev, void* meta_host_pinned, size_t meta_bytes);
/* Workgroup budget of the NEXT lcc_vit_encode calls (round 5): cap > 0 = the tower's MFMA-bound kernels (GEMMs, attention) launch at most
 * `cap` workgroups and walk their tiles persistently, i.e. they occupy `cap` of the 256 CUs; 0 = whole chip (default).  For a tower
 * issued on the second stream UNDER another turn's decode steps: a resident 8-wave GEMM block leaves room for one weight-streaming
 * decode wave per SIMD where five fit on a free CU, so a tower spread over every CU starves the HBM-bound decode kernels of their
 * memory-level parallelism and simply adds its own duration to the decode steps; on half the CUs it takes twice as long under steps that
 * keep their speed.  Same results (the tile -> output mapping does not change).  No reference counterpart: HF runs the tower and the
 * decode loop strictly one after the other (Q2VL:1159-1176 inside forward). */
int lcc_engine_set_vit_grid_cap(lcc_engine* e, int cap);
/* weights by name, borrowed device pointers (bf16 unless stated):  see INTEGRATION.md for the name table */
int lcc_engine_set_weight(lcc_engine* e, const char* name, const void* dev, int64_t numel);
int lcc_engine_weights_ready(const lcc_engine* e, char* missing, int missing_len);

/* live timing of the dominant kernel (the decode gate/up weight-streaming GEMV): hipEvent pairs recorded on the launch
 * stream around up to max_samples launches; read back (blocking) as milliseconds per launch. */
int lcc_engine_profile(lcc_engine* e, int enable, int max_samples);
int lcc_engine_profile_read(lcc_engine* e, float* ms_out, int max_n, int* n_out);
/* the same switch also brackets every WHOLE decode step (28 layers + final norm + lm_head + sampler) with an event pair:
 * milliseconds per decode step, for the step-level roofline (weights + KV bytes / step time) */
int lcc_engine_profile_read_steps(lcc_engine* e, float* ms_out, int max_n, int* n_out);
/* for the same samples: the index of the step inside its lcc_llm_decode call (0 = the step right after the prefill, which shares the
 * GPU with a vision tower prefetched on the side stream; late steps run alone) */
int lcc_engine_profile_read_step_index(lcc_engine* e, int32_t* idx_out, int max_n, int* n_out);

/* Parity instrumentation (tests only; SURVEY section 7 step 1: per-stage tensors).  While bound, every lcc_llm_prefill / lcc_llm_decode
 * call copies the residual stream (HF `hidden_states`, Q2VL:762-844) of its rows into `taps` = bf16 [2*n_layers + 1][max_rows][hidden]:
 * index 0 = the embeddings entering layer 0 (Q2VL:1159-1176), 2l+1 = after the attention block of layer l (the input of
 * post_attention_layernorm, Q2VL:594-603), 2l+2 = the output of layer l (Q2VL:605-612).  `overrides` = bf16 [n_layers + 1][max_rows][hidden]
 * (prefill only): the input of layer l is REPLACED by overrides[l] before the layer runs and overrides[n_layers] replaces the input
 * of the final norm -- teacher forcing per layer, so that one layer's arithmetic (or final norm + lm_head) is compared in isolation
 * with HF's on the oracle's own input.  NULL, NULL, 0 unbinds.  A call with more rows than
 * max_rows fails with LCC_ERR_STATE.  lcc_debug_set_vit_taps: the same for the vision tower (Q2VL:700-729): taps bf16
 * [vit_depth + 1][max_rows][vit_embed] (0 = PatchEmbed output, l+1 = output of block l), overrides [vit_depth + 1][max_rows][vit_embed]
 * (input of block l; [vit_depth] = input of the PatchMerger). */
int lcc_debug_set_llm_taps(lcc_engine* e, void* taps, const void* overrides, int max_rows);
int lcc_debug_set_vit_taps(lcc_engine* e, void* taps, const void* overrides, int max_rows);
/* Teacher forcing (tests only; what the HF oracle does with a forcing LogitsProcessor): while bound, the token the sampler chose at
 * generate-step k (k = 0 the prefill's token, k >= 1 the decode steps; k < n_steps) of the b-th stream of the call is REPLACED by
 * dev_tokens[k * n_streams + b] (int32, device memory) as the current token and in the history -- raw logits and processed scores stay
 * the model's own, so that a committed reference stream (tests/golden/) can be followed step by step even where the reference's own
 * top-1 margin is inside bf16 noise.  Applies to calls with exactly n_streams streams.  NULL, 0, 0 unbinds. */
int lcc_debug_set_forced_tokens(lcc_engine* e, const int32_t* dev_tokens, int n_steps, int n_streams);

/* stream (slot) state */
int lcc_slot_reset(lcc_engine* e, int slot, void* stream);                 /* new video stream: empty KV, empty history */
int lcc_slot_set_length(lcc_engine* e, int slot, int kv_len, int next_pos, void* stream);  /* truncate after EOS */
int lcc_slot_get_length(const lcc_engine* e, int slot, int* kv_len, int* next_pos);

typedef struct {
  const uint8_t* frames;     /* device uint8 frames, or NULL when pixel_values is given          */
  const float* pixel_values; /* device fp32 [P,1176] (the HF processor output) or NULL           */
  int layout, T, H, W;       /* frames: layout 0 THWC / 1 TCHW; pixel_values: T=grid_t*2, H, W in pixels */
} lcc_clip;
/* ViT + merger over n clips -> out bf16 [sum_i P_i/4, hidden].  cos/sin: fp32 [P_total, 40] vision RoPE tables. */
int lcc_vit_encode(lcc_engine* e, int n_clips, const lcc_clip* clips, const float mean255[3], const float std255[3],
                   const float* rope_cos, const float* rope_sin, void* out_embeds, void* stream);

typedef struct {
  float repetition_penalty;  /* 1.0 = off */
  int thr_token;             /* ThresholdLogitsProcessor token id (" ...") or -1 */
  int use_thr; float thr_base, thr_step;
  int eos_token;             /* <|im_end|>: a slot that samples it is frozen for the rest of the call (HF stopping criteria) */
  int suppress_eos;          /* 1 = MinNewTokensLength active for the whole call (min_new_tokens == max_new_tokens) */
  float* scores_out;         /* optional device fp32 [n_streams, V] of processed scores of the LAST step */
  void* logits_out;          /* optional device bf16 [steps, n_streams, V] raw logits of every step (parity tests) */
  int eos_token2;            /* second EOS id of generation_config.json (<|endoftext|>) or -1 */
  int do_sample;             /* 0: argmax (lcc_sample_greedy); 1: lcc_sample_topk_topp with the fields below */
  float temperature;         /* > 0 */
  int top_k;                 /* 0 = off */
  float top_p;               /* (0, 1]; 1 = off */
  uint64_t seed;             /* Philox key; the per-slot draw counters live in the engine state (reset by lcc_slot_reset) */
} lcc_sampling;

/* Prefill of n_streams streams: host arrays are copied through the engine's pinned meta ring.
 *   slots[n]; n_new[n]; ids[S] (S = sum n_new); vit_index[S] (-1 text, else row of vit_embeds);
 *   pos3[3*S] M-RoPE positions (host computes them: Q2VL:914-1016 / 1349-1351).
 * Appends K/V, runs the layers, samples the first new token of every stream (history column 0). */
int lcc_llm_prefill(lcc_engine* e, int n_streams, const int32_t* slots, const int32_t* n_new, const int32_t* ids,
                    const int32_t* vit_index, const void* vit_embeds, const int32_t* pos3, const lcc_sampling* sp,
                    void* stream);
/* n_steps further decode steps for the same streams without any host round trip.  n_streams <= LCC_MAX_DECODE_BATCH (and
 * <= max_new_rows): up to 16 streams are one activation fragment per weight fragment of the weight-streaming GEMVs; 17..64 streams run the
 * same kernels with 2-4 activation fragments per weight fragment (bf16 weights; lcc_debug_set_skinny_rows lowers the limit) or, with fp8
 * weights, the 64-row GEMM tiles -- either way the weights are streamed once per step for the whole batch.  Larger batches: call once
 * per group. */
#define LCC_MAX_DECODE_BATCH 64
int lcc_llm_decode(lcc_engine* e, int n_streams, const int32_t* slots, int n_steps, int first_step_index,
                   const lcc_sampling* sp, void* stream);
/* blocking: copy the ids generated by the last generate call of a slot to the host (at most max_n), report how many
 * were generated (EOS included) and refresh the host mirror of the slot's KV length / next position. */
int lcc_slot_read_tokens(lcc_engine* e, int slot, int32_t* out, int max_n, int* n_generated, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIVECC_AMD_H */
