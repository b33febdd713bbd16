// Internal C++ launch API of the HIP kernels (the C-ABI in capi.hip and the engine call these).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/livecc_amd.h"  // lcc_status codes

namespace lcc {

typedef unsigned short bf16_t;


// ---- launch counters (tests assert WHICH kernel served a call: lcc_debug_launch_counts) ----
enum LaunchCounter {
  LC_ATTN_DECODE = 0,          // attn_decode_kernel (per-wave decode attention)
  LC_ATTN_DECODE_COMBINE = 1,  // attn_decode_combine_kernel
  LC_ATTN_DECODE_FUSED = 2,    // attn_decode_fused_kernel with the combine launch
  LC_ATTN_DECODE_FUSED_MERGE = 3,   // attn_decode_fused_kernel with the in-launch split merge
  LC_GEMV_FUSED_TAIL = 4,      // gemv_skinny_kernel MODE 3 (consumer op in the last-arriving block)
  LC_DGEMV_V2 = 5,             // decode pipeline v2 GEMVs
  LC_ATTN_PREFILL_MFMA32 = 6,  // attn_gqa32_kernel
  LC_ATTN_PREFILL_SHARED = 7,  // attn_shared_kernel (LLM prefill)
  LC_ATTN_PREFILL_PER_WAVE = 8,
  LC_ATTN_PREFILL_COMBINE = 9,
  LC_LAST_DECODE_NSPLIT = 10,  // key splits of the most recent decode attention launch (a value, not a count)
  LC_LAST_PREFILL_NSPLIT = 11, // key splits of the most recent prefill attention launch
  LC_GEMM_TALL = 12,
  LC_ATTN_VIT32 = 13,          // attn_vit32_kernel (vision attention on 32x32x16 MFMAs)
  LC_GEMM_VH = 14,             // gemm_vh_kernel (row tiles of 256 / 272 / 288 rows, small class 128 / 144; round 5 -- the slot of the retired gemm_pp_kernel)
  LC_GEMM_VIT_QKV = 15,        // gemm_big_kernel with the EPI_VIT_QK / EPI_VIT_V epilogues (RoPE / V transpose fused into the q|k|v projection; one EPI_VIT_QKV launch per tower block when E % 128 == 0, else a q|k launch + a V launch)
  LC_COUNT = 16
};
extern long long g_launch_counts[LC_COUNT];

// ---- per-stream KV arena layout ----
struct KvLayout {  // per-stream KV arena: [layer][K|V][Hkv][Lmax][128]; V blocked-transposed [Lmax/32][128][32]
  int n_layers, n_kv_heads, lmax, head_dim;
  __host__ __device__ size_t head_stride() const { return (size_t)lmax * head_dim; }
  __host__ __device__ size_t kv_stride() const { return (size_t)n_kv_heads * lmax * head_dim; }
  __host__ __device__ size_t layer_stride() const { return (size_t)2 * n_kv_heads * lmax * head_dim; }
  __host__ __device__ size_t total() const { return (size_t)n_layers * layer_stride(); }
};

// ---- GEMM (gemm.hip) ----
// optional fused tail of a split-K skinny GEMV (M <= 2): the last-arriving block reduces the slabs and runs the consumer
struct GemvTail {
  int kind = 0;                 // 0 none; 1 residual add + RMSNorm; 2 bias + M-RoPE + KV append
  int32_t* counter = nullptr;   // device word, zero before the launch (the tail resets it)
  // kind 1
  bf16_t* h = nullptr; const bf16_t* norm_w = nullptr; bf16_t* y = nullptr; float eps = 0.f;
  // kind 2
  const bf16_t* bias = nullptr; const bf16_t* cs = nullptr; const bf16_t* sn = nullptr;
  const int32_t* tok_stream = nullptr; const int32_t* tok_pos = nullptr; const int32_t* kv_len = nullptr;
  bf16_t* const* kv_base = nullptr; KvLayout lay = {0, 0, 0, 0}; int layer = 0; bf16_t* q_out = nullptr; int n_q_heads = 0;
};
constexpr int GEMM_EPI_VIT_QK = 6, GEMM_EPI_VIT_V = 7, GEMM_EPI_VIT_QKV = 8;   // GemmArgs::epilogue values of the internal vision q|k / V epilogues (common.h; the public LCC_EPI_* codes are 0-4)
// extra operands of the EPI_VIT_QK / EPI_VIT_V epilogues (vision-tower q|k|v projection with 2-D RoPE and the V transpose fused in)
struct VitQkvEpi {
  const float* cs = nullptr; const float* sn = nullptr;   // QK: fp32 [P, 40]: cos / sin of the 40 rotation pairs of a head (head_dim 80)
  const int32_t* grp_off = nullptr;                       // V: [P / 4]: element offset of patch 4i's key slot inside a (head, channel 0) plane of vt
                                                          //    = (first block of its segment + local index / 32) * 80 * 32 + local index % 32
  bf16_t* vt = nullptr; int total_blocks = 0;             // V blocked-transposed [head][block][80][32]
  int E = 0;                                              // embed dim (QK: N = 2E, V: N = E, QKV: N = 3E with E % 128 == 0)
};
struct GemmArgs {
  const bf16_t* A = nullptr; int lda = 0;       // [M,K]
  const bf16_t* W = nullptr; int ldw = 0;       // [N,K] row-major, or packed fragments (w_packed, ldw ignored)
  int w_packed = 0;                             // 1: [N/16][ceil(K/32)][4][16][8] (see gemm.hip)
  const bf16_t* bias = nullptr;                 // [N] or null
  const bf16_t* residual = nullptr; int ldr = 0;
  bf16_t* C = nullptr; int ldc = 0;             // [M,N] (swiglu: [M,N/2])
  float* partial = nullptr; int nsplit = 0;     // fp32 split-K slabs [nsplit][M][N] instead of C (consumer reduces)
  int M = 0, N = 0, K = 0;
  int epilogue = 0;
  GemvTail tail;
  // fp8 (OCP e4m3) weights: W points at bytes in the PACKED8 order [N/16][K/64][4 g][16 rows][16 k], wscale = fp32 [N]
  // (y[n] = wscale[n] * sum_k x[k] q[n][k]); M > 16 needs dq_scratch = N*K bf16 (exact dequantisation into the bf16 packed
  // order, then the bf16 GEMM with the scale in its epilogue)
  int w_fp8 = 0; const float* wscale = nullptr; bf16_t* dq_scratch = nullptr;
  VitQkvEpi vq;                                 // epilogue == GEMM_EPI_VIT_QK / GEMM_EPI_VIT_V only
};
int gemm_bf16(const GemmArgs& a, hipStream_t st);
bool gemm_vit_qkv_eligible(int M, int E, int K);   // shapes the EPI_VIT_QK / EPI_VIT_V epilogues serve (8-wave kernel: K % 64 == 0; M > 64, M % 4 == 0, E % 32 == 0)
int gemv_num_splits(int N, int K);
int gemm_plan(int M, int N, int K, int epilogue, int nsplit, bool w_fp8);     // host logic: which kernel family would serve this GEMM
int gemm_tiled_num_splits(int M, int N, int K, bool packed_bf16 = false);      // packed_bf16: the small variable-height tiles may serve it (more splits)
void set_gemv_variant(int v);
bool gemm_routes_skinny(int M, int K, bool w_fp8);   // M rows x [N, K] weights take the weight-streaming GEMV kernels (the one predicate: gemm.hip and the engine)
int set_skinny_rows(int rows);     // 16..64: largest M served by the weight-streaming GEMV kernels (returns the previous value)
void set_gemm_variant(int v);
void set_grid_cap(int cap);        // > 0: the MFMA-bound tile kernels (8-wave / 4-wave LDS-DMA GEMMs, vision attention) launch at most `cap` workgroups
int get_grid_cap();                //      and walk their tiles persistently (gemm.hip: g_grid_cap); 0 = one workgroup per tile
int mfma_probe(const bf16_t* A, const bf16_t* B, float* D, hipStream_t st);

// ---- elementwise / normalisation / layout (elementwise.hip) ----
int patchify_norm_u8(const uint8_t* frames, int layout, int T, int H, int W, const float* mean255,
                     const float* std255, bf16_t* out, int ld, hipStream_t st);
int resize_bicubic_aa_u8(const uint8_t* src, int layout, int T, int Hin, int Win, uint8_t* dst, int Hout, int Wout,
                         const int32_t* xmin, const int32_t* xsize, const float* wx, int kx, const int32_t* ymin,
                         const int32_t* ysize, const float* wy, int ky, float* tmp, hipStream_t st);
int cast_f32_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t st);
int layernorm_bf16(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows, int dim, float eps,
                   hipStream_t st);
int rmsnorm_bf16(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int dim, float eps, hipStream_t st);
int add_rmsnorm_bf16(bf16_t* h, const bf16_t* delta_bf16, const float* delta_partial, int nsplit, const bf16_t* w,
                     bf16_t* y, int rows, int dim, float eps, hipStream_t st);
int vit_rope_vt_bf16(bf16_t* qkv, const float* cos, const float* sin, const int32_t* seg_of_patch,
                     const int32_t* seg_start, const int32_t* seg_blk_start, bf16_t* vt, int P, int heads,
                     int total_blocks, hipStream_t st);
int mrope_table(const int32_t* pos3, const float* inv_freq, int S, int sec_t, int sec_h, bf16_t* cos, bf16_t* sin,
                hipStream_t st);
int mrope_table_decode(const int32_t* slots, const int32_t* pos, const float* inv_freq, int B, bf16_t* cos, bf16_t* sin,
                       hipStream_t st);
int swiglu_bf16(const bf16_t* g, const bf16_t* u, bf16_t* out, int64_t n, hipStream_t st);

int rope_kv_append_bf16(const bf16_t* qkv_bf16, const float* qkv_partial, int nsplit, const bf16_t* bias,
                        const bf16_t* cos, const bf16_t* sin, const int32_t* tok_stream, const int32_t* tok_pos,
                        const int32_t* kv_len, bf16_t* const* kv_base, KvLayout lay, int layer, bf16_t* q_out, int S,
                        int n_q_heads, hipStream_t st);
int embed_gather_bf16(const int32_t* ids, const int32_t* indirect, const int32_t* vit_index, const bf16_t* table,
                      const bf16_t* vit_rows, bf16_t* out, int S, int dim, hipStream_t st);
int gather_rows_bf16(const bf16_t* in, const int32_t* rows, bf16_t* out, int n, int dim, hipStream_t st);

// ---- attention (attention.hip) ----
void set_attn_variant(int v);
int get_attn_variant();
// ViT attention on 32x32x16 MFMAs (attn32.hip): groups of 8 x 32 query rows of one segment (grp_seg / grp_q0, 256-row groups)
int attn_vit32_launch(const bf16_t* qkv, const bf16_t* vt, bf16_t* out, const int32_t* grp_seg, const int32_t* grp_q0,
                      const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_groups, int heads,
                      int total_blocks, float scale_log2e, hipStream_t st, int group_rows = 256);
// LLM prefill attention on 32-row tiles / 32x32x16 MFMAs (attn32.hip); partials in the layout of attn_prefill_combine_kernel
// query rows per tile of the 32x32x16 prefill kernel: the most a block can serve (NWAVE * 32 / G with the pair packing, else 32) and what
// the engine builds (LCC_ATTN32_TILE_ROWS caps it)
int attn32_max_tile_rows(int G);
int attn32_tile_rows(int G);
// tile height + key-split count of one 32x32x16 prefill launch (host logic only; attn32.hip)
void attn32_plan(const int* n_new, int n_streams, int max_kv, int G, int n_kv_heads, int cus, int* tile_rows, int* splits);
int attn_prefill32_launch(const bf16_t* q, bf16_t* out, const int32_t* tile_stream, const int32_t* tile_q0, const int32_t* tile_nq,
                          const int32_t* tile_pos0, bf16_t* const* kv_base, KvLayout lay, int layer, int n_tiles, int n_q_heads,
                          int nsplit, float* ws_o, float* ws_ml, float scale_log2e, hipStream_t st);
// tile tables: 32-row query tiles (per-wave kernel); group tables: 128-row groups of one segment (LDS-shared kernel)
int attn_vit_bf16(const bf16_t* qkv, const bf16_t* vt, bf16_t* out, const int32_t* tile_seg, const int32_t* tile_q0,
                  const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_tiles,
                  int heads, int total_blocks, const int32_t* grp_seg, const int32_t* grp_q0, int n_groups, hipStream_t st);
int attn_prefill_bf16(const bf16_t* q, bf16_t* out, const int32_t* tile_stream, const int32_t* tile_q0,
                      const int32_t* tile_nq, const int32_t* tile_pos0, bf16_t* const* kv_base, KvLayout lay,
                      int layer, int n_tiles, int n_q_heads, int tile_rows, int nsplit, int n_rows, float* ws_o, float* ws_ml,
                      hipStream_t st);
// Host-known stream state of a decode step, passed BY VALUE to the decode attention (round 6): the arena pointer and the key count
// (cached keys + the new token) of each of up to 4 rows.  The kernel then has no dependent index loads (slots[b] -> kv_len[slot], kv_base[slot]
// -> K / V: two cross-XCD round trips in front of the first key tile).  The host knows both: lcc_llm_decode counts the steps it enqueues;
// a stream that EOS froze on the device (`done`) has fewer keys than the host's count -- its step output is discarded anyway (the sampler
// leaves a done slot untouched), and the rows past its length are finite bytes of the same arena.
struct AttnDirect { const bf16_t* base[4]; int n[4]; int used = 0; };
int attn_decode_bf16(const bf16_t* q, bf16_t* out, const int32_t* slots, const int32_t* kv_len, bf16_t* const* kv_base,
                     KvLayout lay, int layer, int B, int n_q_heads, int nsplit, float* ws_o, float* ws_ml, hipStream_t st,
                     const AttnDirect* direct = nullptr);
void set_attn_fused_tail(int v);
// fused decode attention: bias + M-RoPE + KV append + attention + split merge in one launch (reads the qkv GEMV's fp32 slabs)
int attn_decode_fused_bf16(const float* qkv_part, int ns_qkv, const bf16_t* bias, const bf16_t* cs, const bf16_t* sn,
                           const int32_t* slots, const int32_t* kv_len, bf16_t* const* kv_base, KvLayout lay, int layer, int B,
                           int n_q_heads, int nsplit, float* ws_o, float* ws_ml, int32_t* counters, bf16_t* out, hipStream_t st);

// ---- sampler (sampler.hip) ----
int seen_set(uint32_t* seen, int words_per_stream, const int32_t* ids, const int32_t* slot_of_id, int n, int indirect,
             const int32_t* done, hipStream_t st);
int sample_greedy(const bf16_t* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream,
                  const int32_t* stream_slot, float repetition_penalty, int thr_token, int use_thr, float thr_value,
                  int eos_token, int eos_token2, int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history, int hist_ld,
                  int32_t* hist_col, float* scores_out, float* ws, hipStream_t st);
// do_sample: temperature -> top-k -> top-p -> multinomial (Philox stream per slot); top_k 0 = off, top_p 1 = off
int sample_topk_topp(const bf16_t* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream, const int32_t* stream_slot,
                     float repetition_penalty, int thr_token, int use_thr, float thr_value, int eos_token, int eos_token2,
                     int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history, int hist_ld, int32_t* hist_col,
                     float* scores_out, float temperature, int top_k, float top_p, uint64_t seed, uint32_t* rng_ctr, hipStream_t st);
// "first use on this device" latch for per-kernel attributes (160-KB dynamic LDS): hipFuncSetAttribute applies to the CURRENT device, so a
// process that drives a second GPU must set it there too (the shipped deployment is one process per GPU).  Host only.
struct DeviceOnce {
  unsigned long long seen = 0;
  bool first() {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (seen & bit) return false;
    seen |= bit;
    return true;
  }
};
int advance_lengths(const int32_t* slots, int32_t* kv_len, int32_t* pos, int B, const int32_t* done, hipStream_t st);
// teacher forcing (tests): the token just sampled for stream b becomes forced[b] (current token + last history column); the forced stream
// also decides where a slot ends (`done`: flags from before this step, updated by a forced EOS)
int force_tokens(const int32_t* slots, const int32_t* forced, int B, int32_t* cur_tok, int32_t* history, int hist_ld, const int32_t* hist_col,
                 int32_t* done, int eos, int eos2, hipStream_t st);

// ---- decode layer pipeline v2 (decode_v2.hip): elementwise stages fused into the weight-streaming GEMVs ----
struct DgArgs {
  const bf16_t* W = nullptr; int M = 0, N = 0, K = 0;
  const float* wscale = nullptr;   // non-null: W = OCP e4m3 bytes in the PACKED8 order [N/16][K/64][4 g][16 rows][16 k] + this fp32 scale per stored row
  const bf16_t* X = nullptr; int ldx = 0;                                      // plain activation rows [M, K]
  const bf16_t* H = nullptr; const float* stats = nullptr; int n_stat = 0; const bf16_t* norm_w = nullptr; float eps = 0.f;  // RMSNorm prologue
  const bf16_t* bias = nullptr; bf16_t* C = nullptr; int ldc = 0;              // bf16 / SwiGLU epilogue
  bf16_t* Hres = nullptr; float* stats_out = nullptr;                          // residual epilogue
  const bf16_t* cs = nullptr; const bf16_t* sn = nullptr; const int32_t* tok_stream = nullptr; const int32_t* kv_len = nullptr;
  bf16_t* const* kv_base = nullptr; KvLayout lay = {0, 0, 0, 0}; int layer = 0; bf16_t* q_out = nullptr; int n_q_heads = 0;   // rope epilogue
};
int decode_step_begin(const int32_t* slots, const int32_t* cur_tok, const int32_t* done, uint32_t* seen, int words, const bf16_t* table,
                      bf16_t* h, float* stats, int dim, const int32_t* pos, const float* inv_freq, bf16_t* cs, bf16_t* sn, int B,
                      hipStream_t st);
int dgemv_qkv_rope(const DgArgs& a, hipStream_t st);      // [RMSNorm] q|k|v Linear (row-permuted decode copy) [bias + M-RoPE + KV append]
int set_resid_waves(int mode);   // 0: 8-wave blocks, 1: 16 waves for bf16 weights with K >= 8192 (default), 2: 16 waves always; returns the old mode
int dgemv_resid(const DgArgs& a, hipStream_t st);         // o_proj / down_proj [residual add in place + per-tile sums of squares]
int dgemv_norm_swiglu(const DgArgs& a, hipStream_t st);   // [RMSNorm] gate/up Linear [SwiGLU]
// Live timing of the dominant kernel (round 6): the NEXT dgemv_norm_swiglu launch of this host thread carries the two events ON THE DISPATCH
// (hipExtLaunchKernel: start = the kernel's begin timestamp, stop = its end -- what rocprofv3 reports) instead of being bracketed by
// hipEventRecord calls, whose pair also times ~3.5 us of dispatch gaps around a 44-us kernel (47.3 vs 43.9 us, profiles/r06).
void dgemv_attach_events_to_next_swiglu(hipEvent_t start, hipEvent_t stop);
// chained launch: down_proj of layer l + q/k/v of layer l+1, the consumer's weights prefetched under the producer (decode_v2.hip)
int dgemv_chain_capacity();
int dgemv_down_qkv(const DgArgs& down, const DgArgs& qkv, unsigned* flag, unsigned target, unsigned* err, hipStream_t st);
int dgemv_norm_bf16(const DgArgs& a, hipStream_t st);     // [final RMSNorm] lm_head

}  // namespace lcc
