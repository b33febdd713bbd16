// LLM side of the engine: Qwen2VLTextModel.forward (HF modeling_qwen2_vl.py:762-844) + lm_head (1320-1323) + GenerationMixin._sample
// (generation/utils.py:2783-2960) as launch sequences: lcc_llm_prefill (packed rows of several streams over their carried KV) and
// lcc_llm_decode (n steps without a host round trip; decode pipeline v2 for 1-2 streams, the weight-streaming GEMV sequence up to 64).
#include "engine_internal.h"

// 1: decode pipeline v2 launches down_proj(l) + q/k/v(l+1) as ONE chained launch where both grids fit the chip at once (decode_v2.hip).
// Measured on MI355X (LiveCC-7B, one stream, no ViT prefetch; profiles/r03/decode_chain_ab.jsonl): bit-identical, but SLOWER --
// 3105-3229 us per decode step against 2989 us for the two launches (the chained kernel 43.6 us vs 25.5 + 10.7 us).  The consumer's
// 33 MB of weights are served at the START of the launch (total HBM bytes are the same), and what the hand-off then exposes after the
// producer is the consumer's whole serial tail (flag -> statistics -> normalise -> LDS -> 14 MFMAs -> reduce -> RoPE epilogue, ~5 us)
// that a stand-alone launch hides under its own weight stream -- as much as the removed kernel boundary was worth.  Kept as a tested
// variant (lcc_debug_set_decode_chain(1)); default off.
int g_decode_chain = 0;

// ------------------------------------------------------------------------------------------------
// LLM
// ------------------------------------------------------------------------------------------------
namespace {
int g_fused_attn = 1;   // decode: 0 three kernels; 1 rope/KV-append + attention fused for multi-stream batches; 2 for every batch
int g_fuse_tails = 0;   // 1: batch-1 decode runs rope/KV-append and residual+RMSNorm as tails of the producing GEMV (last-arriving
                        // block, ticket counter).  Measured on MI355X at 7B shapes: 184 tok/s fused vs 215 tok/s with separate
                        // kernels (the slab write-through + ticket serialises the GEMV's tail), so it stays an opt-in variant.
struct LlmBuffers {
  bf16_t *h, *xn, *qkv, *q, *attn, *act, *cos, *sin, *last_h, *last_xn, *logits, *dq;
  float *partial, *ws_o, *ws_ml, *stats;
};
int carve_llm(lcc_engine* e, LlmBuffers* b) {
  const size_t S = e->lim.max_new_rows, B = e->lim.max_slots, H = e->c.hidden_size, I = e->c.intermediate_size, V = e->c.vocab_size;
  Carver cv; cv.base = e->ws;
  b->h = cv.take<bf16_t>(S * H); b->xn = cv.take<bf16_t>(S * H); b->qkv = cv.take<bf16_t>(S * e->qkvd);
  b->q = cv.take<bf16_t>(S * e->qd); b->attn = cv.take<bf16_t>(S * e->qd); b->act = cv.take<bf16_t>(S * I);
  b->cos = cv.take<bf16_t>(S * 64); b->sin = cv.take<bf16_t>(S * 64);
  b->partial = cv.take<float>(std::max((size_t)MAX_SPLIT * 64 * std::max<size_t>(e->qkvd, H), (size_t)6 * std::min<size_t>(S, 4096) * H));
  b->last_h = cv.take<bf16_t>(B * H); b->last_xn = cv.take<bf16_t>(B * H);
  b->stats = cv.take<float>(16 * (H / 16 + 4));
  b->logits = cv.take<bf16_t>(B * V);
  const size_t nslot = std::max<size_t>(B * e->c.n_kv_heads * 128 * 16, std::min<size_t>(S, 1024) * e->c.n_q_heads * 8);
  b->ws_o = cv.take<float>(nslot * 128); b->ws_ml = cv.take<float>(nslot * 128);
  b->dq = e->c.llm_fp8 ? cv.take<bf16_t>(std::max<size_t>((size_t)e->qkvd * H, 2 * I * H)) : nullptr;
  if (cv.off > e->ws_bytes) return fail(LCC_ERR_STATE, "workspace too small");
  return 0;
}

// the 28 decoder layers over S packed rows; on exit b.h holds the residual stream after the last layer and,
// on the skinny path (S <= 16), b.xn already holds final_norm(h).
struct LayerCtx {
  int S; bool skinny;
  const int32_t *tok_stream, *tok_pos;          // prefill: explicit positions; decode: tok_pos == nullptr
  const int32_t *tile_stream, *tile_q0, *tile_nq, *tile_pos0; int n_tiles, tile_rows, kv_split;  // prefill attention tiles
  const int32_t* slots; int B; int nsplit_attn; int nsplit_attn_fused;  // decode attention
};
int run_layers(lcc_engine* e, const LlmBuffers& b, const LayerCtx& cx, hipStream_t st) {
  const int H = e->c.hidden_size, I = e->c.intermediate_size, S = cx.S;
  const float eps = e->c.rms_eps;
  const int sp_qkv = cx.skinny ? std::min(MAX_SPLIT, gemv_num_splits(e->qkvd, H)) : 0;
  const int sp_o = cx.skinny ? std::min(MAX_SPLIT, gemv_num_splits(H, e->qd)) : 0;
  const int sp_dn = cx.skinny ? std::min(MAX_SPLIT, gemv_num_splits(H, I)) : 0;
  // prefill with few output tiles (N = hidden): split-K slabs, reduced by the fused residual-add + RMSNorm kernel
  // (up to 6 slabs of S x H floats: the slab buffer holds 6 x min(max_new_rows, 4096) x H -- carve_llm -- and S <= max_new_rows)
  const bool w16 = !e->c.llm_fp8;
  const int tp_o = cx.skinny ? 1 : gemm_tiled_num_splits(S, H, e->qd, w16);
  const int tp_dn = cx.skinny ? 1 : gemm_tiled_num_splits(S, H, I, w16);
  // q/k/v of a short prefill (one streaming chunk): split-K slabs consumed by the rope / KV-append kernel (267.5 -> 269.1 tok/s single
  // stream; LCC_PREFILL_QKV_SPLIT=0 restores the bf16 GEMM output)
  static const int qkv_split_on = [] { const char* v = getenv("LCC_PREFILL_QKV_SPLIT"); return v ? atoi(v) : 1; }();
  const int tp_qkv = (cx.skinny || !qkv_split_on || e->c.llm_fp8 || (size_t)6 * std::min<size_t>(e->lim.max_new_rows, 4096) * H <
                      (size_t)8 * S * e->qkvd) ? 1 : gemm_tiled_num_splits(S, e->qkvd, H, true);
  // fp8 weights: the same GemmArgs with the byte pointer, the row scales and the dequantisation scratch of the tiled path
  auto set_w = [&](GemmArgs& g, const bf16_t* w, const float* scale) {
    g.w_packed = 1; g.W = w;
    if (scale != nullptr) { g.w_fp8 = 1; g.wscale = scale; g.dq_scratch = b.dq; }
  };
  LCC_TRY(rmsnorm_bf16(b.h, e->llm[0].in_norm, b.xn, S, H, eps, st));
  // parity instrumentation: tap 0 = embeddings, 2l+1 = residual stream after the attention block of layer l, 2l+2 = after its MLP;
  // an override replaces the INPUT of layer l (teacher forcing per layer: every layer is fed the oracle's hidden state)
  if ((e->llm_taps || e->llm_over) && S > e->llm_tap_rows) return fail(LCC_ERR_STATE, "LLM taps bound for %d rows, call has %d", e->llm_tap_rows, S);
  const size_t tap_stride = (size_t)e->llm_tap_rows * H, tap_bytes = (size_t)S * H * 2;
  if (e->llm_taps) HIP_TRY(hipMemcpyAsync(e->llm_taps, b.h, tap_bytes, hipMemcpyDeviceToDevice, st));
  for (int l = 0; l < e->c.n_layers; ++l) {
    const LlmLayerW& L = e->llm[l];
    const bf16_t* next_norm = (l + 1 < e->c.n_layers) ? e->llm[l + 1].in_norm : e->final_norm;
    GemmArgs g;
    if (e->llm_over) {
      HIP_TRY(hipMemcpyAsync(b.h, e->llm_over + (size_t)l * tap_stride, tap_bytes, hipMemcpyDeviceToDevice, st));
      LCC_TRY(rmsnorm_bf16(b.h, L.in_norm, b.xn, S, H, eps, st));
    }
    // q/k/v projection (+bias) -> M-RoPE -> in-place KV append
    g = GemmArgs(); set_w(g, L.qkv_w, L.qkv_s); g.A = b.xn; g.lda = H; g.ldw = H; g.M = S; g.N = e->qkvd; g.K = H;
    const bool fuse = cx.skinny && S <= 2 && g_fuse_tails && !e->c.llm_fp8;   // batch-1 decode: consumer ops run as GEMV tails
    // batch decode: bias + M-RoPE + KV append + attention + split merge in one launch (attention.hip)
    // Measured on MI355X (tools/bench_kernels.py --attn, 7B heads): one stream 14.2 vs 14.3 us per layer (no gain: the chain is a
    // sequence of dependent memory round trips either way), 8 streams 24.6 vs 29.2 us (6k keys), 37.9 vs 41.9 us (12k keys) --
    // so the fused kernel serves batches with >= 16 (stream, KV head) pairs; g_fused_attn = 2 forces it for every batch.
    const bool fused_attn = !fuse && cx.skinny && cx.tok_pos == nullptr && cx.B * e->c.n_kv_heads <= 256 &&
                            (g_fused_attn == 2 || (g_fused_attn == 1 && cx.B * e->c.n_kv_heads >= 16));
    if (fuse) {
      g.partial = b.partial; g.nsplit = sp_qkv;
      g.tail.kind = 2; g.tail.counter = e->d_counter; g.tail.bias = L.qkv_b; g.tail.cs = b.cos; g.tail.sn = b.sin;
      g.tail.tok_stream = cx.tok_stream; g.tail.tok_pos = cx.tok_pos; g.tail.kv_len = e->d_kv_len; g.tail.kv_base = e->d_kv_base;
      g.tail.lay = e->lay; g.tail.layer = l; g.tail.q_out = b.q; g.tail.n_q_heads = e->c.n_q_heads;
      LCC_TRY(gemm_bf16(g, st));
    } else if (cx.skinny) {
      g.partial = b.partial; g.nsplit = sp_qkv;
      LCC_TRY(gemm_bf16(g, st));
      if (!fused_attn)
        LCC_TRY(rope_kv_append_bf16(nullptr, b.partial, sp_qkv, L.qkv_b, b.cos, b.sin, cx.tok_stream, cx.tok_pos, e->d_kv_len,
                                    e->d_kv_base, e->lay, l, b.q, S, e->c.n_q_heads, st));
    } else if (tp_qkv > 1) {
      // one streaming chunk: N = 4608 is 252 tiles of 64 x 128 (one latency-bound block per CU) -> split K, the fp32 slabs are
      // reduced (+ bias, one bf16 rounding as in the GEMM epilogue) by the rope / KV-append kernel
      g.partial = b.partial; g.nsplit = tp_qkv;
      LCC_TRY(gemm_bf16(g, st));
      LCC_TRY(rope_kv_append_bf16(nullptr, b.partial, tp_qkv, L.qkv_b, b.cos, b.sin, cx.tok_stream, cx.tok_pos, e->d_kv_len,
                                  e->d_kv_base, e->lay, l, b.q, S, e->c.n_q_heads, st));
    } else {
      g.bias = L.qkv_b; g.C = b.qkv; g.ldc = e->qkvd;
      LCC_TRY(gemm_bf16(g, st));
      LCC_TRY(rope_kv_append_bf16(b.qkv, nullptr, 0, nullptr, b.cos, b.sin, cx.tok_stream, cx.tok_pos, e->d_kv_len,
                                  e->d_kv_base, e->lay, l, b.q, S, e->c.n_q_heads, st));
    }
    // attention
    if (fused_attn)
      LCC_TRY(attn_decode_fused_bf16(b.partial, sp_qkv, L.qkv_b, b.cos, b.sin, cx.slots, e->d_kv_len, e->d_kv_base, e->lay, l, cx.B,
                                     e->c.n_q_heads, cx.nsplit_attn_fused, b.ws_o, b.ws_ml, e->d_attn_cnt, b.attn, st));
    else if (cx.tok_pos == nullptr)
      LCC_TRY(attn_decode_bf16(b.q, b.attn, cx.slots, e->d_kv_len, e->d_kv_base, e->lay, l, cx.B, e->c.n_q_heads, cx.nsplit_attn,
                               b.ws_o, b.ws_ml, st));
    else
      LCC_TRY(attn_prefill_bf16(b.q, b.attn, cx.tile_stream, cx.tile_q0, cx.tile_nq, cx.tile_pos0, e->d_kv_base, e->lay, l,
                                cx.n_tiles, e->c.n_q_heads, cx.tile_rows, cx.kv_split, S, b.ws_o, b.ws_ml, st));
    // o_proj + residual + post-attention RMSNorm
    g = GemmArgs(); set_w(g, L.o_w, L.o_s); g.A = b.attn; g.lda = e->qd; g.ldw = e->qd; g.M = S; g.N = H; g.K = e->qd;
    if (fuse) {
      g.partial = b.partial; g.nsplit = sp_o;
      g.tail.kind = 1; g.tail.counter = e->d_counter; g.tail.h = b.h; g.tail.norm_w = L.post_norm; g.tail.y = b.xn; g.tail.eps = eps;
      LCC_TRY(gemm_bf16(g, st));
    } else if (cx.skinny) {
      g.partial = b.partial; g.nsplit = sp_o;
      LCC_TRY(gemm_bf16(g, st));
      LCC_TRY(add_rmsnorm_bf16(b.h, nullptr, b.partial, sp_o, L.post_norm, b.xn, S, H, eps, st));
    } else if (tp_o > 1) {
      g.partial = b.partial; g.nsplit = tp_o;
      LCC_TRY(gemm_bf16(g, st));
      LCC_TRY(add_rmsnorm_bf16(b.h, nullptr, b.partial, tp_o, L.post_norm, b.xn, S, H, eps, st));
    } else {
      g.residual = b.h; g.ldr = H; g.C = b.h; g.ldc = H; g.epilogue = LCC_EPI_RESIDUAL;
      LCC_TRY(gemm_bf16(g, st));
      LCC_TRY(rmsnorm_bf16(b.h, L.post_norm, b.xn, S, H, eps, st));
    }
    if (e->llm_taps) HIP_TRY(hipMemcpyAsync(e->llm_taps + (size_t)(2 * l + 1) * tap_stride, b.h, tap_bytes, hipMemcpyDeviceToDevice, st));
    // SwiGLU MLP
    g = GemmArgs(); set_w(g, L.gate_up_w, L.gate_up_s); g.A = b.xn; g.lda = H; g.ldw = H; g.C = b.act; g.ldc = I; g.M = S; g.N = 2 * I; g.K = H;
    g.epilogue = LCC_EPI_SWIGLU;
    // one sampled launch per decode step (the middle layer): an event pair opens a ~6 us bubble on the stream on each side, which
    // at 28 pairs per step was 8 % of the round-1 step time
    // (decode steps of every batch size: the 17-64-stream path through the GEMM tiles is sampled too)
    const bool prof = e->prof_on && cx.tok_pos == nullptr && l == e->c.n_layers / 2 && 2 * (e->prof_n + 1) <= (int)e->prof_ev.size();
    if (prof) HIP_TRY(hipEventRecord(e->prof_ev[2 * e->prof_n], st));
    LCC_TRY(gemm_bf16(g, st));
    if (prof) { HIP_TRY(hipEventRecord(e->prof_ev[2 * e->prof_n + 1], st)); e->prof_n++; }
    g = GemmArgs(); set_w(g, L.down_w, L.down_s); g.A = b.act; g.lda = I; g.ldw = I; g.M = S; g.N = H; g.K = I;
    if (fuse) {
      g.partial = b.partial; g.nsplit = sp_dn;
      g.tail.kind = 1; g.tail.counter = e->d_counter; g.tail.h = b.h; g.tail.norm_w = next_norm; g.tail.y = b.xn; g.tail.eps = eps;
      LCC_TRY(gemm_bf16(g, st));
    } else if (cx.skinny) {
      g.partial = b.partial; g.nsplit = sp_dn;
      LCC_TRY(gemm_bf16(g, st));
      LCC_TRY(add_rmsnorm_bf16(b.h, nullptr, b.partial, sp_dn, next_norm, b.xn, S, H, eps, st));
    } else if (tp_dn > 1) {
      g.partial = b.partial; g.nsplit = tp_dn;
      LCC_TRY(gemm_bf16(g, st));
      LCC_TRY(add_rmsnorm_bf16(b.h, nullptr, b.partial, tp_dn, (l + 1 < e->c.n_layers) ? next_norm : nullptr, b.xn, S, H, eps, st));
    } else {
      g.residual = b.h; g.ldr = H; g.C = b.h; g.ldc = H; g.epilogue = LCC_EPI_RESIDUAL;
      LCC_TRY(gemm_bf16(g, st));
      if (l + 1 < e->c.n_layers) LCC_TRY(rmsnorm_bf16(b.h, next_norm, b.xn, S, H, eps, st));
    }
    if (e->llm_taps) HIP_TRY(hipMemcpyAsync(e->llm_taps + (size_t)(2 * l + 2) * tap_stride, b.h, tap_bytes, hipMemcpyDeviceToDevice, st));
  }
  if (e->llm_over) {   // overrides[n_layers] = the input of the final norm (isolates final norm + lm_head)
    HIP_TRY(hipMemcpyAsync(b.h, e->llm_over + (size_t)e->c.n_layers * tap_stride, tap_bytes, hipMemcpyDeviceToDevice, st));
    if (cx.skinny) LCC_TRY(rmsnorm_bf16(b.h, e->final_norm, b.xn, S, H, eps, st));
  }
  return 0;
}

int g_decode_path = 1;   // 1: decode pipeline v2 (decode_v2.hip: 6 launches per layer) where eligible; 0: the round-1 launch sequence
bool decode_v2_ok(const lcc_engine* e) {
  if (g_decode_path != 1) return false;
  if ((e->c.hidden_size & 63) || e->c.hidden_size > 8192 || (e->c.intermediate_size & 31) || (e->qd & 31)) return false;
  if (e->c.llm_fp8 && ((e->c.intermediate_size & 63) || (e->qd & 63))) return false;     // fp8: whole 64-k fragments
  for (const LlmLayerW& L : e->llm) if (L.qkv_w_dec == nullptr || (e->c.llm_fp8 && L.qkv_s_dec == nullptr)) return false;
  return true;
}
// the 28 decoder layers of ONE decode step over B rows, v2 launch sequence.  On entry b.h / b.stats / b.cos / b.sin come from
// decode_step_begin; on exit b.h is the residual stream after the last layer and b.stats its per-tile sums of squares (the final
// RMSNorm runs as the prologue of the lm_head GEMV).
int run_decode_layers_v2(lcc_engine* e, const LlmBuffers& b, int B, const int32_t* d_slots, int nsplit_attn, hipStream_t st,
                         const AttnDirect* direct = nullptr) {
  const int H = e->c.hidden_size, I = e->c.intermediate_size;
  const float eps = e->c.rms_eps;
  if (e->llm_over) return fail(LCC_ERR_STATE, "per-layer input overrides are a prefill-only instrument (decode pipeline v2 carries row statistics)");
  if (e->llm_taps && B > e->llm_tap_rows) return fail(LCC_ERR_STATE, "LLM taps bound for %d rows, decode batch has %d", e->llm_tap_rows, B);
  const size_t tap_stride = (size_t)e->llm_tap_rows * H, tap_bytes = (size_t)B * H * 2;
  if (e->llm_taps) HIP_TRY(hipMemcpyAsync(e->llm_taps, b.h, tap_bytes, hipMemcpyDeviceToDevice, st));
  // chained launches: down_proj of layer l and q/k/v of layer l+1 in ONE launch (the consumer's weights stream under the producer's
  // tail: decode_v2.hip).  Only when both grids fit the chip at once, <= 2 streams, <= 127 layers, and no parity taps are bound
  // (a tap copy between the two halves would have to sit inside the launch).
  auto qkv_args = [&](int l) {
    const LlmLayerW& L = e->llm[l];
    DgArgs a; a.W = L.qkv_w_dec; a.wscale = L.qkv_s_dec; a.M = B; a.N = e->qkvd; a.K = H; a.H = b.h; a.stats = b.stats; a.n_stat = H / 16; a.norm_w = L.in_norm;
    a.eps = eps; a.bias = L.qkv_b; a.cs = b.cos; a.sn = b.sin; a.tok_stream = d_slots; a.kv_len = e->d_kv_len; a.kv_base = e->d_kv_base;
    a.lay = e->lay; a.layer = l; a.q_out = b.q; a.n_q_heads = e->c.n_q_heads;
    return a;
  };
  const bool chain = g_decode_chain && !e->c.llm_fp8 && B <= 2 && e->c.n_layers <= 127 && !e->llm_taps && (long)B * H * 2 <= 16 * 1024 &&
                     H / 16 + e->qkvd / 16 <= dgemv_chain_capacity();
  for (int l = 0; l < e->c.n_layers; ++l) {
    const LlmLayerW& L = e->llm[l];
    DgArgs a;
    if (l == 0 || !chain) LCC_TRY(dgemv_qkv_rope(qkv_args(l), st));     // otherwise launched together with the previous layer's down_proj
    LCC_TRY(attn_decode_bf16(b.q, b.attn, d_slots, e->d_kv_len, e->d_kv_base, e->lay, l, B, e->c.n_q_heads, nsplit_attn, b.ws_o, b.ws_ml, st, direct));
    a = DgArgs(); a.W = L.o_w; a.wscale = L.o_s; a.M = B; a.N = H; a.K = e->qd; a.X = b.attn; a.ldx = e->qd; a.Hres = b.h; a.stats_out = b.stats;
    LCC_TRY(dgemv_resid(a, st));
    if (e->llm_taps) HIP_TRY(hipMemcpyAsync(e->llm_taps + (size_t)(2 * l + 1) * tap_stride, b.h, tap_bytes, hipMemcpyDeviceToDevice, st));
    a = DgArgs(); a.W = L.gate_up_w; a.wscale = L.gate_up_s; a.M = B; a.N = 2 * I; a.K = H; a.H = b.h; a.stats = b.stats; a.n_stat = H / 16; a.norm_w = L.post_norm;
    a.eps = eps; a.C = b.act; a.ldc = I;
    const bool prof = e->prof_on && l == e->c.n_layers / 2 && 2 * (e->prof_n + 1) <= (int)e->prof_ev.size();   // one sample per step
    // the sampled launch carries its events on the dispatch (kernel begin / end timestamps: kernels.h)
    if (prof) dgemv_attach_events_to_next_swiglu(e->prof_ev[2 * e->prof_n], e->prof_ev[2 * e->prof_n + 1]);
    LCC_TRY(dgemv_norm_swiglu(a, st));
    if (prof) e->prof_n++;
    a = DgArgs(); a.W = L.down_w; a.wscale = L.down_s; a.M = B; a.N = H; a.K = I; a.X = b.act; a.ldx = I; a.Hres = b.h; a.stats_out = b.stats;
    if (chain && l + 1 < e->c.n_layers) {
      const unsigned target = ++e->chain_epoch[l] * (unsigned)(H / 16);     // monotonic counter: every launch adds H/16 arrivals
      LCC_TRY(dgemv_down_qkv(a, qkv_args(l + 1), e->d_chain + l, target, e->d_chain + 128, st));
    } else {
      LCC_TRY(dgemv_resid(a, st));
    }
    if (e->llm_taps) HIP_TRY(hipMemcpyAsync(e->llm_taps + (size_t)(2 * l + 2) * tap_stride, b.h, tap_bytes, hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

int head_and_sample(lcc_engine* e, const LlmBuffers& b, const bf16_t* xn_rows, int B, const int32_t* d_slots, const lcc_sampling* sp,
                    int step_index, hipStream_t st) {
  const int H = e->c.hidden_size, V = e->c.vocab_size;
  bf16_t* logits = b.logits;
  if (sp && sp->logits_out) logits = (bf16_t*)sp->logits_out + (size_t)step_index * B * V;
  if (xn_rows == nullptr) {   // decode v2: final RMSNorm of b.h as the prologue of the lm_head GEMV
    DgArgs a; a.W = e->lm_head; a.wscale = e->lm_head_s; a.M = B; a.N = V; a.K = H; a.H = b.h; a.stats = b.stats; a.n_stat = H / 16; a.norm_w = e->final_norm;
    a.eps = e->c.rms_eps; a.C = logits; a.ldc = V;
    LCC_TRY(dgemv_norm_bf16(a, st));
  } else {
    GemmArgs g; g.w_packed = 1; g.A = xn_rows; g.lda = H; g.W = e->lm_head; g.ldw = H; g.C = logits; g.ldc = V; g.M = B; g.N = V; g.K = H;
    if (e->lm_head_s != nullptr) { g.w_fp8 = 1; g.wscale = e->lm_head_s; }
    LCC_TRY(gemm_bf16(g, st));
  }
  const float pen = sp ? sp->repetition_penalty : 1.0f;
  const int thr_tok = sp ? sp->thr_token : -1;
  const int use_thr = sp ? sp->use_thr : 0;
  const float thr = sp ? sp->thr_base + sp->thr_step * (float)step_index : 0.f;
  const int eos2 = sp ? sp->eos_token2 : -1;
  // teacher forcing: the forced stream decides where a slot ends, so the model's own pick must not set `done` (a model EOS at
  // a step where the forced stream goes on froze the slot and later forced tokens overwrote its last history column)
  const bool forcing = e->forced != nullptr && B == e->forced_B && step_index < e->forced_steps;
  // bit 0 masks the EOS scores (the caller's min_new_tokens), bit 1 only keeps a picked EOS from setting `done` (sampler.hip): under forcing the
  // processed scores keep the oracle's definition (EOS unmasked unless the caller asked)
  const int sup_eos = ((sp && sp->suppress_eos) ? 1 : 0) | (forcing ? 2 : 0);
  if (sp && sp->do_sample && sp->top_k != 1) {
    LCC_TRY(sample_topk_topp(logits, V, B, V, e->d_seen, e->words, d_slots, pen <= 0.f ? 1.0f : pen, thr_tok, use_thr, thr, sp->eos_token,
                             eos2, sup_eos, e->d_done, e->d_cur_tok, e->d_history, e->lim.max_history, e->d_hist_col,
                             sp->scores_out, sp->temperature, sp->top_k, sp->top_p, sp->seed, e->d_rng_ctr, st));
    if (forcing)
      LCC_TRY(force_tokens(d_slots, e->forced + (size_t)step_index * B, B, e->d_cur_tok, e->d_history, e->lim.max_history, e->d_hist_col, e->d_done,
                           sp->suppress_eos ? -1 : sp->eos_token, sp->suppress_eos ? -1 : eos2, st));
    return 0;
  }
  // top_k == 1 (the released generation_config): the top-k warper leaves one finite score -> the draw IS the argmax
  LCC_TRY(sample_greedy(logits, V, B, V, e->d_seen, e->words, d_slots, pen <= 0.f ? 1.0f : pen, thr_tok, use_thr, thr,
                        sp ? sp->eos_token : -1, eos2, sup_eos, e->d_done, e->d_cur_tok, e->d_history, e->lim.max_history,
                        e->d_hist_col, sp ? sp->scores_out : nullptr, b.ws_ml, st));
  if (forcing) {
    const bool keep_going = sp && sp->suppress_eos;      // force_length: a forced EOS does not end the slot either
    LCC_TRY(force_tokens(d_slots, e->forced + (size_t)step_index * B, B, e->d_cur_tok, e->d_history, e->lim.max_history, e->d_hist_col, e->d_done,
                         keep_going || !sp ? -1 : sp->eos_token, keep_going || !sp ? -1 : eos2, st));
  }
  return 0;
}
}  // namespace

extern "C" int lcc_llm_prefill(lcc_engine* e, int n_streams, const int32_t* slots, const int32_t* n_new, const int32_t* ids,
                               const int32_t* vit_index, const void* vit_embeds, const int32_t* pos3, const lcc_sampling* sp,
                               void* stream) {
  LCC_TRY(ensure_ready(e));
  CallScope in_flight;
  std::lock_guard<std::mutex> lk(e->mu_llm);
  if (n_streams <= 0 || !slots || !n_new || !ids || !pos3) return fail(LCC_ERR_ARG, "null argument");
  if (n_streams > e->lim.max_slots) return fail(LCC_ERR_STATE, "too many streams");
  hipStream_t st = (hipStream_t)stream;
  int S = 0;
  for (int b = 0; b < n_streams; ++b) {
    if (slots[b] < 0 || slots[b] >= e->lim.max_slots || !e->h_kv_base[slots[b]]) return fail(LCC_ERR_STATE, "slot %d not bound", slots[b]);
    if (n_new[b] <= 0) return fail(LCC_ERR_ARG, "stream %d has no new tokens", b);
    if (e->h_kv_len[slots[b]] + n_new[b] + e->lim.max_history > e->lim.max_kv_len)
      return fail(LCC_ERR_STATE, "slot %d: KV capacity %d exceeded (%d cached + %d new + %d generation headroom)", slots[b], e->lim.max_kv_len,
                  e->h_kv_len[slots[b]], n_new[b], e->lim.max_history);
    S += n_new[b];
  }
  if (S > e->lim.max_new_rows) return fail(LCC_ERR_STATE, "%d new rows > max_new_rows %d", S, e->lim.max_new_rows);
  for (int i = 0; i < S; ++i) {
    if (ids[i] < 0 || ids[i] >= e->c.vocab_size) return fail(LCC_ERR_ARG, "token id %d out of range at %d", ids[i], i);
    if (vit_index && vit_index[i] >= 0 && !vit_embeds) return fail(LCC_ERR_ARG, "vit_index set but vit_embeds is null");
  }
  LlmBuffers bf; LCC_TRY(carve_llm(e, &bf));

  // host tables
  std::vector<int32_t> tok_stream(S), tok_pos(S), last_row(n_streams), tile_stream, tile_q0, tile_nq, tile_pos0;
  // 32-row query tiles unless that leaves the GPU mostly idle (a 386-row chunk: 13 tiles x 28 heads = 364 waves)
  // 32-row tiles (NQ = 2) need ~200 VGPRs = one 7-wave block per CU; 16-row tiles run two blocks per CU.  Measured (8 streams x
  // 386 rows against 6k keys): 674 us with 16-row tiles vs 747 us with 32-row tiles, so the wide tile is kept for very large
  // prefills only (e.g. the 8 x 1114-row first turn), where the grid is several waves of blocks either way.
  // attention variant 3 (attn32.hip: 32x32x16 MFMAs, one wave = 32 rows of one head) always takes 32-row tiles.
  const bool mfma32 = get_attn_variant() == 3 && e->c.n_q_heads / e->c.n_kv_heads <= 8;
  // Round 6: a block of that kernel can hold 8 x 32 / G rows -- 36 at 7 heads per KV head, all eight waves busy instead of seven -- and a
  // block's time does not depend on that (it is set by its busiest SIMD).  Tile height and key-split count are planned TOGETHER
  // (attn32.hip: attn32_plan, pinned on the CPU through lcc_debug_attn_plan).
  int tile_rows = mfma32 ? 32 : ((long)((S + 31) / 32) * e->c.n_q_heads >= 6144 ? 32 : 16);
  int max_kv = 0;
  for (int b = 0; b < n_streams; ++b) max_kv = std::max(max_kv, e->h_kv_len[slots[b]] + n_new[b]);
  const int ks_cap = std::max(1, std::min(8, (max_kv / 32) / 8));      // >= 8 key tiles per split
  int ks32 = 1;
  if (mfma32) attn32_plan(n_new, n_streams, max_kv, e->c.n_q_heads / e->c.n_kv_heads, e->c.n_kv_heads, e->cu_count, &tile_rows, &ks32);
  int row = 0;
  for (int b = 0; b < n_streams; ++b) {
    const int past = e->h_kv_len[slots[b]];
    for (int i = 0; i < n_new[b]; ++i) { tok_stream[row + i] = slots[b]; tok_pos[row + i] = past + i; }
    for (int q = 0; q < n_new[b]; q += tile_rows) {
      tile_stream.push_back(slots[b]); tile_q0.push_back(row + q); tile_nq.push_back(std::min(tile_rows, n_new[b] - q)); tile_pos0.push_back(past + q);
    }
    row += n_new[b];
    last_row[b] = row - 1;
  }
  const int n_tiles = (int)tile_stream.size();
  MetaWriter mw; LCC_TRY(meta_begin(e, &mw));
  int32_t *d_ids, *d_vit = nullptr, *d_pos3, *d_tok_stream, *d_tok_pos, *d_last_row, *d_slots, *d_ts, *d_tq, *d_tn, *d_tp;
  bool ok = mw.put(ids, S, &d_ids) && mw.put(pos3, (size_t)3 * S, &d_pos3) && mw.put(tok_stream.data(), S, &d_tok_stream) &&
            mw.put(tok_pos.data(), S, &d_tok_pos) && mw.put(last_row.data(), n_streams, &d_last_row) && mw.put(slots, n_streams, &d_slots) &&
            mw.put(tile_stream.data(), n_tiles, &d_ts) && mw.put(tile_q0.data(), n_tiles, &d_tq) && mw.put(tile_nq.data(), n_tiles, &d_tn) &&
            mw.put(tile_pos0.data(), n_tiles, &d_tp);
  if (ok && vit_index) ok = mw.put(vit_index, S, &d_vit) != nullptr;
  if (!ok) return fail(LCC_ERR_STATE, "meta ring slot too small");
  LCC_TRY(meta_commit(&mw, st));

  // history column restarts at 0 for this generate call; repetition penalty sees every id of the history
  for (int b = 0; b < n_streams; ++b) {
    HIP_TRY(hipMemsetAsync(e->d_hist_col + slots[b], 0, 4, st));
    HIP_TRY(hipMemsetAsync(e->d_done + slots[b], 0, 4, st));
  }
  LCC_TRY(seen_set(e->d_seen, e->words, d_ids, d_tok_stream, S, 0, nullptr, st));
  LCC_TRY(embed_gather_bf16(d_ids, nullptr, d_vit, e->embed, (const bf16_t*)vit_embeds, bf.h, S, e->c.hidden_size, st));
  LCC_TRY(mrope_table(d_pos3, e->inv_freq, S, e->c.mrope_sec_t, e->c.mrope_sec_h, bf.cos, bf.sin, st));

  LayerCtx cx{};
  cx.S = S; cx.skinny = S <= 16; cx.tok_stream = d_tok_stream; cx.tok_pos = d_tok_pos;
  cx.tile_stream = d_ts; cx.tile_q0 = d_tq; cx.tile_nq = d_tn; cx.tile_pos0 = d_tp; cx.n_tiles = n_tiles; cx.tile_rows = tile_rows;
  {  // few query tiles against a long cache (a streaming chunk): also split the keys so that every SIMD gets 2-3 waves
    const long waves = (long)n_tiles * e->c.n_q_heads;
    int ks = (int)std::min<long>(8, 3072 / std::max<long>(waves, 1));
    ks = std::min(ks, (max_kv / 32) / 16);          // >= 16 key tiles per split
    cx.kv_split = (S <= 1024 && ks >= 2) ? ks : 1;
    if (mfma32) {      // the plan made with the tile height above
      static const int forced = [] { const char* v = getenv("LCC_ATTN32_SPLIT"); return v ? atoi(v) : 0; }();
      // a forced split obeys the same bound as the automatic one: the partial buffers hold min(S, 1024) x heads x 8 slots (carve_llm)
      cx.kv_split = (forced > 0 && S <= 1024) ? std::min(forced, ks_cap) : (forced > 0 ? 1 : ks32);
    }
  }
  cx.slots = d_slots; cx.B = n_streams; cx.nsplit_attn = 1;
  LCC_TRY(run_layers(e, bf, cx, st));

  const bf16_t* xn_rows;
  if (cx.skinny) {
    LCC_TRY(gather_rows_bf16(bf.xn, d_last_row, bf.last_xn, n_streams, e->c.hidden_size, st));
    xn_rows = bf.last_xn;
  } else {
    LCC_TRY(gather_rows_bf16(bf.h, d_last_row, bf.last_h, n_streams, e->c.hidden_size, st));
    LCC_TRY(rmsnorm_bf16(bf.last_h, e->final_norm, bf.last_xn, n_streams, e->c.hidden_size, e->c.rms_eps, st));
    xn_rows = bf.last_xn;
  }
  // lengths: the new rows are now in the cache
  row = 0;
  for (int b = 0; b < n_streams; ++b) {
    const int s = slots[b];
    // In-call decode positions continue from the LAST prompt row (+1 on every axis): HF generation/utils.py:975-985 extends
    // position_ids[..., -1:] + 1.  The prompt always ends in text (assistant header), where the three axes are equal; under
    // the transformers-4.5x text-offset rule that row also holds the maximum, i.e. this equals kv_len + rope_delta
    // (Q2VL:1014).  The NEXT call's positions are past_len + i + rope_delta, computed by the host (protocol.positions_with_cache).
    const int last = row + n_new[b] - 1;
    const int mx = std::max(pos3[last], std::max(pos3[S + last], pos3[2 * S + last]));
    e->h_kv_len[s] += n_new[b];
    e->h_pos[s] = mx + 1;
    row += n_new[b];
  }
  {
    MetaWriter mw2; LCC_TRY(meta_begin(e, &mw2));
    std::vector<int32_t> kv(n_streams), ps(n_streams); int32_t *d_kv, *d_ps;
    for (int b = 0; b < n_streams; ++b) { kv[b] = e->h_kv_len[slots[b]]; ps[b] = e->h_pos[slots[b]]; }
    mw2.put(kv.data(), n_streams, &d_kv); mw2.put(ps.data(), n_streams, &d_ps);
    LCC_TRY(meta_commit(&mw2, st));
    for (int b = 0; b < n_streams; ++b) {
      HIP_TRY(hipMemcpyAsync(e->d_kv_len + slots[b], d_kv + b, 4, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpyAsync(e->d_pos + slots[b], d_ps + b, 4, hipMemcpyDeviceToDevice, st));
    }
  }
  LCC_TRY(head_and_sample(e, bf, xn_rows, n_streams, d_slots, sp, 0, st));
  return check_launch("lcc_llm_prefill");
}

extern "C" int lcc_llm_decode(lcc_engine* e, int n_streams, const int32_t* slots, int n_steps, int first_step_index,
                              const lcc_sampling* sp, void* stream) {
  LCC_TRY(ensure_ready(e));
  CallScope in_flight;
  std::lock_guard<std::mutex> lk(e->mu_llm);
  if (n_streams <= 0 || !slots || n_steps < 0) return fail(LCC_ERR_ARG, "bad argument");
  // <= 16 streams: one activation fragment per weight fragment of the weight-streaming GEMVs.  17..64 (bf16 weights): the same kernels with
  // 2-4 activation fragments per weight fragment (gemv_skinny_kernel<..., MG>, round 4); fp8 weights above 16 rows: the 64-row GEMM tiles
  // of the prefill path.  One predicate decides (gemm.hip: gemm_routes_skinny).  Beyond 64 the caller splits.
  if (n_streams > LCC_MAX_DECODE_BATCH)
    return fail(LCC_ERR_SHAPE, "decode batches of more than %d streams are not supported", LCC_MAX_DECODE_BATCH);
  if (n_streams > e->lim.max_new_rows) return fail(LCC_ERR_STATE, "%d streams > max_new_rows %d", n_streams, e->lim.max_new_rows);
  if (n_steps == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int max_len = 0;
  for (int b = 0; b < n_streams; ++b) {
    const int s = slots[b];
    if (s < 0 || s >= e->lim.max_slots || !e->h_kv_base[s]) return fail(LCC_ERR_STATE, "slot %d not bound", s);
    if (e->h_kv_len[s] + n_steps > e->lim.max_kv_len) return fail(LCC_ERR_STATE, "slot %d: KV capacity exceeded", s);
    max_len = std::max(max_len, e->h_kv_len[s] + n_steps);
  }
  if (first_step_index + n_steps > e->lim.max_history) return fail(LCC_ERR_STATE, "history capacity %d exceeded", e->lim.max_history);
  LlmBuffers bf; LCC_TRY(carve_llm(e, &bf));
  MetaWriter mw; LCC_TRY(meta_begin(e, &mw));
  int32_t* d_slots;
  if (!mw.put(slots, n_streams, &d_slots)) return fail(LCC_ERR_STATE, "meta ring slot too small");
  LCC_TRY(meta_commit(&mw, st));
  const int ntile = (max_len + 31) / 32;
  // key tiles per split of the per-wave decode attention (tuning knob LCC_ATTN_TPS, default 4) and the split cap (LCC_ATTN_MAXSPLIT, 64)
  static const int tps = [] { const char* v = getenv("LCC_ATTN_TPS"); return v ? std::max(1, atoi(v)) : 4; }();
  static const int maxsplit = [] { const char* v = getenv("LCC_ATTN_MAXSPLIT"); return v ? std::max(1, std::min(128, atoi(v))) : 64; }();
  const int nsplit = std::max(1, std::min(maxsplit, (ntile + tps - 1) / tps));

  LayerCtx cx{};
  // weight-streaming ("skinny") layer sequence for up to 64 streams: fp32 split-K slabs consumed by rope / add+norm kernels.  fp8 weights
  // keep the round-3 routing above 16 rows (their GEMV multiplies one activation fragment per weight fragment)
  cx.S = n_streams;
  {
    const bool f8 = e->c.llm_fp8 != 0;      // every Linear of a layer must take the same route (gemm.hip: gemm_routes_skinny)
    cx.skinny = gemm_routes_skinny(n_streams, e->c.hidden_size, f8) && gemm_routes_skinny(n_streams, e->qd, f8) && gemm_routes_skinny(n_streams, e->c.intermediate_size, f8);
  } cx.tok_stream = d_slots; cx.tok_pos = nullptr; cx.slots = d_slots; cx.B = n_streams;
  cx.nsplit_attn = nsplit;
  // fused kernel: 4 waves per block; about one block per CU, never less than one key tile per wave
  static const int fused_blocks = [] { const char* v = getenv("LCC_ATTN_FUSED_BLOCKS"); return v ? std::max(64, atoi(v)) : 256; }();
  cx.nsplit_attn_fused = std::max(1, std::min(std::min(32, (ntile + 3) / 4), std::max(1, fused_blocks / (n_streams * e->c.n_kv_heads))));
  // v2 serves batches of one or two streams (measured on MI355X at 7B shapes: 246 vs 242 tokens/s for one stream, 414 vs 410 for
  // two, but 640 vs 655 for four: with more rows the per-block normalisation prologue outweighs the saved launches)
  const bool v2 = decode_v2_ok(e) && n_streams <= 2 && (long)n_streams * e->c.hidden_size <= 16384;
  for (int step = 0; step < n_steps; ++step) {
    const bool prof_step = e->prof_on && (step & 3) == 0 && 2 * (e->step_n + 1) <= (int)e->step_ev.size();   // every 4th step
    if (prof_step) HIP_TRY(hipEventRecord(e->step_ev[2 * e->step_n], st));
    // the token sampled by the previous step (d_cur_tok[slot]) is embedded, appended at kv_len[slot], position pos[slot]
    if (v2) {
      LCC_TRY(decode_step_begin(d_slots, e->d_cur_tok, e->d_done, e->d_seen, e->words, e->embed, bf.h, bf.stats, e->c.hidden_size, e->d_pos,
                                e->inv_freq, bf.cos, bf.sin, n_streams, st));
      // the step's key counts and arena pointers travel by value (kernels.h: AttnDirect; LCC_ATTN_DIRECT=0: index loads on the device)
      static const int direct_on = [] { const char* v = getenv("LCC_ATTN_DIRECT"); return v ? atoi(v) : 1; }();
      AttnDirect dir;
      if (direct_on && n_streams <= 4) {
        for (int bb = 0; bb < n_streams; ++bb) { dir.base[bb] = (const bf16_t*)e->h_kv_base[slots[bb]]; dir.n[bb] = e->h_kv_len[slots[bb]] + step + 1; }
        dir.used = n_streams;
      }
      LCC_TRY(run_decode_layers_v2(e, bf, n_streams, d_slots, nsplit, st, dir.used ? &dir : nullptr));
    } else {
      LCC_TRY(seen_set(e->d_seen, e->words, e->d_cur_tok, d_slots, n_streams, 1, e->d_done, st));
      LCC_TRY(embed_gather_bf16(e->d_cur_tok, d_slots, nullptr, e->embed, nullptr, bf.h, n_streams, e->c.hidden_size, st));
      LCC_TRY(mrope_table_decode(d_slots, e->d_pos, e->inv_freq, n_streams, bf.cos, bf.sin, st));
      LCC_TRY(run_layers(e, bf, cx, st));
      if (!cx.skinny) LCC_TRY(rmsnorm_bf16(bf.h, e->final_norm, bf.xn, n_streams, e->c.hidden_size, e->c.rms_eps, st));
    }
    LCC_TRY(advance_lengths(d_slots, e->d_kv_len, e->d_pos, n_streams, e->d_done, st));
    LCC_TRY(head_and_sample(e, bf, v2 ? nullptr : bf.xn, n_streams, d_slots, sp, first_step_index + step, st));
    if (prof_step) {
      HIP_TRY(hipEventRecord(e->step_ev[2 * e->step_n + 1], st));
      if (e->step_n < (int)e->step_rel.size()) e->step_rel[e->step_n] = step;
      e->step_n++;
    }
  }
  for (int b = 0; b < n_streams; ++b) { e->h_kv_len[slots[b]] += n_steps; e->h_pos[slots[b]] += n_steps; }
  return check_launch("lcc_llm_decode");
}

// parity instrumentation (tests only): see include/livecc_amd.h
extern "C" int lcc_debug_set_llm_taps(lcc_engine* e, void* taps, const void* overrides, int max_rows) {
  if (!e || max_rows < 0 || ((taps || overrides) && max_rows == 0)) return fail(LCC_ERR_ARG, "bad argument");
  if (((uintptr_t)taps | (uintptr_t)overrides) & 15) return fail(LCC_ERR_ALIGN, "tap buffers must be 16-byte aligned");
  e->llm_taps = (bf16_t*)taps; e->llm_over = (const bf16_t*)overrides; e->llm_tap_rows = max_rows;
  return 0;
}
extern "C" int lcc_debug_set_vit_taps(lcc_engine* e, void* taps, const void* overrides, int max_rows) {
  if (!e || max_rows < 0 || ((taps || overrides) && max_rows == 0)) return fail(LCC_ERR_ARG, "bad argument");
  if (((uintptr_t)taps | (uintptr_t)overrides) & 15) return fail(LCC_ERR_ALIGN, "tap buffers must be 16-byte aligned");
  e->vit_taps = (bf16_t*)taps; e->vit_over = (const bf16_t*)overrides; e->vit_tap_rows = max_rows;
  return 0;
}
extern "C" int lcc_debug_set_forced_tokens(lcc_engine* e, const int32_t* dev_tokens, int n_steps, int n_streams) {
  if (!e || n_steps < 0 || n_streams < 0 || (dev_tokens && (n_steps == 0 || n_streams == 0))) return fail(LCC_ERR_ARG, "bad argument");
  e->forced = dev_tokens; e->forced_steps = dev_tokens ? n_steps : 0; e->forced_B = dev_tokens ? n_streams : 0;
  return 0;
}
// process-global knobs: refused (LCC_ERR_STATE, negative for the setters that return the previous value) while a model-level call is in flight
extern "C" int lcc_debug_set_fused_tails(int on) { if (int kg__ = lcc_knob_guard("lcc_debug_set_fused_tails")) return kg__; g_fuse_tails = on ? 1 : 0; return 0; }
extern "C" int lcc_debug_set_decode_chain(int on) { if (int kg__ = lcc_knob_guard("lcc_debug_set_decode_chain")) return kg__; g_decode_chain = on ? 1 : 0; return 0; }
extern "C" int lcc_debug_set_resid_waves(int mode) { if (lcc_knob_guard("lcc_debug_set_resid_waves")) return -1; return set_resid_waves(mode); }
extern "C" int lcc_debug_set_skinny_rows(int rows) { if (lcc_knob_guard("lcc_debug_set_skinny_rows")) return -1; return set_skinny_rows(rows); }
extern "C" int lcc_debug_set_decode_path(int path) {
  if (int kg__ = lcc_knob_guard("lcc_debug_set_decode_path")) return kg__;
  if (path != 0 && path != 1) return fail(LCC_ERR_ARG, "decode path must be 0 (round-1 launch sequence) or 1 (v2)");
  g_decode_path = path;
  return 0;
}
// bit 0: engine uses the fused decode attention for batches of >= 16 (stream, KV head) pairs (default); bit 2: for every batch;
// bit 1: its key splits are merged in-launch (ticket) instead of by a combine launch
extern "C" int lcc_debug_set_fused_attn(int mode) {
  if (int kg__ = lcc_knob_guard("lcc_debug_set_fused_attn")) return kg__;
  g_fused_attn = (mode & 4) ? 2 : (mode & 1);
  set_attn_fused_tail((mode & 2) ? 0 : 1);
  return 0;
}
