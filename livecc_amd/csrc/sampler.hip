// Fused logits processors + greedy sampler (gfx950), one block per stream, one pass over the vocabulary.
//
// Replaces, in HF's order (generation/utils.py:2894-2925, logits processors merged at 1174-1175 + custom list):
//   1. logits.float()                                  (lm_head output is bf16, utils.py:2894)
//   2. RepetitionPenaltyLogitsProcessor                (logits_process.py ~373-415): for every id present in the
//      whole history (prompt, <|video_pad|> ids and generated tokens): s = s<0 ? s*p : s/p.
//      The history is a per-stream bitmap of seen ids (V bits) instead of a gather over >= 24k ids per step.
//   3. ThresholdLogitsProcessor (ref demo/infer.py:10-23): p = softmax(scores)[tok]; if p <= threshold the
//      token's score becomes -inf.  Only the max and the exp-sum of the scores are needed for that one prob.
//   4. argmax (first index on ties, as torch.argmax on CPU returns).
// The new token is written to the per-slot "current token" word (the id the next decode step embeds) and to the
// per-slot history matrix; its bit enters the bitmap only when a later step actually consumes it (seen_set with
// `indirect`), exactly like HF where `past_ids = sequences[:, :-1]` never contains the last generated token
// (ref demo/infer.py:174).  MinNewTokensLength (force_length) = the EOS score is -inf.  Once a slot samples EOS its
// `done` flag freezes all its device counters, so the decode loop needs no host round trip to honour EOS.
#include "common.h"
#include "kernels.h"

namespace lcc {

// indirect == 0: id i = ids[i].  indirect == 1: id i = ids[slot_of_id[i]] (per-slot current token), skipped when done.
__global__ __launch_bounds__(256) void seen_set_kernel(uint32_t* __restrict__ seen, int words, const int32_t* __restrict__ ids,
                                                       const int32_t* __restrict__ slot_of_id, int n, int indirect,
                                                       const int32_t* __restrict__ done) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int slot = slot_of_id[i];
  if (indirect && done != nullptr && done[slot]) return;
  const int id = indirect ? ids[slot] : ids[i];
  atomicOr(seen + (size_t)slot * words + (id >> 5), 1u << (id & 31));
}
int seen_set(uint32_t* seen, int words_per_stream, const int32_t* ids, const int32_t* slot_of_id, int n, int indirect,
             const int32_t* done, hipStream_t st) {
  if (n <= 0) return 0;
  seen_set_kernel<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(seen, words_per_stream, ids, slot_of_id, n, indirect, done);
  return 0;
}

struct Best { float v; int i; };
LCC_DEVICE Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

template <int NT>
__global__ __launch_bounds__(NT) void sample_greedy_kernel(
    const bf16_t* __restrict__ logits, int ld, int V, uint32_t* __restrict__ seen, int words,
    const int32_t* __restrict__ stream_slot, float penalty, int thr_token, int use_thr, float thr_value,
    int eos_token, int suppress_eos, int32_t* __restrict__ done,
    int32_t* __restrict__ out_tokens, int32_t* __restrict__ history, int hist_ld, int32_t* __restrict__ hist_col,
    float* __restrict__ scores_out) {
  __shared__ float s_max[NT / 64], s_sum[NT / 64], s_bv[NT / 64];
  __shared__ int s_bi[NT / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = stream_slot[b];
  if (done != nullptr && done[slot]) return;  // this stream already emitted EOS in this generate call
  const bf16_t* lg = logits + (size_t)b * ld;
  const uint32_t* sb = seen + (size_t)slot * words;
  float* so = scores_out ? scores_out + (size_t)b * V : nullptr;

  Best best = {-INFINITY, 0x7fffffff};
  float mx = -INFINITY, sum = 0.f, thr_score = -INFINITY;
  for (int c = tid; c * 8 < V; c += NT) {  // V % 8 == 0
    const u32x4 q = ld16(lg + c * 8);
    const uint32_t bits = (sb[c >> 2] >> ((c & 3) * 8)) & 0xffu;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = lo2f(q[e]); v[2 * e + 1] = hi2f(q[e]); }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (penalty != 1.0f && ((bits >> e) & 1u)) v[e] = v[e] < 0.f ? v[e] * penalty : v[e] / penalty;
      const int id = c * 8 + e;
      if (suppress_eos && id == eos_token) v[e] = -INFINITY;  // MinNewTokensLengthLogitsProcessor
      if (so) so[id] = v[e];
      // online max / exp-sum over ALL scores (softmax denominator of the threshold processor)
      if (v[e] > mx) { sum = sum * __expf(mx - v[e]) + 1.f; mx = v[e]; }
      else sum += __expf(v[e] - mx);
      if (id == thr_token) thr_score = v[e];
      else best = better(best, Best{v[e], id});
    }
  }
  // block reductions
  float wmx = wave_max(mx);
  sum *= (mx == -INFINITY) ? 0.f : __expf(mx - wmx);
  sum = wave_sum(sum);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other = {__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
    best = better(best, other);
  }
  thr_score = wave_max(thr_score);
  __shared__ float s_thr[NT / 64];
  if (lane == 0) { s_max[wave] = wmx; s_sum[wave] = sum; s_bv[wave] = best.v; s_bi[wave] = best.i; s_thr[wave] = thr_score; }
  __syncthreads();
  if (tid == 0) {
    float M = -INFINITY;
    for (int w = 0; w < NT / 64; ++w) M = fmaxf(M, s_max[w]);
    float tot = 0.f, ts = -INFINITY;
    Best bb = {-INFINITY, 0x7fffffff};
    for (int w = 0; w < NT / 64; ++w) {
      tot += (s_max[w] == -INFINITY) ? 0.f : s_sum[w] * __expf(s_max[w] - M);
      bb = better(bb, Best{s_bv[w], s_bi[w]});
      ts = fmaxf(ts, s_thr[w]);
    }
    int tok = bb.i;
    if (thr_token >= 0) {
      bool suppressed = false;
      if (use_thr) {
        const float p = __expf(ts - M) / tot;
        suppressed = p <= thr_value;
      }
      if (suppressed) { if (so) so[thr_token] = -INFINITY; }
      else tok = better(bb, Best{ts, thr_token}).i;
    }
    out_tokens[slot] = tok;  // per-slot: the id the next decode step embeds
    if (history != nullptr) {
      const int col = hist_col[slot];
      if (col < hist_ld) history[(size_t)slot * hist_ld + col] = tok;
      hist_col[slot] = col + 1;
    }
    if (done != nullptr && tok == eos_token) done[slot] = 1;
  }
}

// ---- two-stage variant: stage 1 spreads the V-wide pass over SAMPLE_NB blocks per stream, stage 2 merges the partials
constexpr int SAMPLE_NB = 32;

__global__ __launch_bounds__(256) void sample_partial_kernel(
    const bf16_t* __restrict__ logits, int ld, int V, const uint32_t* __restrict__ seen, int words,
    const int32_t* __restrict__ stream_slot, float penalty, int thr_token, int eos_token, int suppress_eos,
    const int32_t* __restrict__ done, float* __restrict__ part, float* __restrict__ scores_out) {
  __shared__ float s_max[4], s_sum[4], s_bv[4], s_thr[4];
  __shared__ int s_bi[4];
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = stream_slot[b];
  if (done != nullptr && done[slot]) return;
  const bf16_t* lg = logits + (size_t)b * ld;
  const uint32_t* sb = seen + (size_t)slot * words;
  float* so = scores_out ? scores_out + (size_t)b * V : nullptr;
  const int nch = V / 8, per = (nch + SAMPLE_NB - 1) / SAMPLE_NB;
  const int c0 = blk * per, c1 = min(nch, c0 + per);
  Best best = {-INFINITY, 0x7fffffff};
  float mx = -INFINITY, sum = 0.f, thr_score = -INFINITY;
  for (int c = c0 + tid; c < c1; c += 256) {
    const u32x4 q = ld16(lg + c * 8);
    const uint32_t bits = (sb[c >> 2] >> ((c & 3) * 8)) & 0xffu;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = lo2f(q[e]); v[2 * e + 1] = hi2f(q[e]); }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (penalty != 1.0f && ((bits >> e) & 1u)) v[e] = v[e] < 0.f ? v[e] * penalty : v[e] / penalty;
      const int id = c * 8 + e;
      if (suppress_eos && id == eos_token) v[e] = -INFINITY;
      if (so) so[id] = v[e];
      if (v[e] > mx) { sum = sum * __expf(mx - v[e]) + 1.f; mx = v[e]; }
      else sum += __expf(v[e] - mx);
      if (id == thr_token) thr_score = v[e];
      else best = better(best, Best{v[e], id});
    }
  }
  float wmx = wave_max(mx);
  sum *= (mx == -INFINITY) ? 0.f : __expf(mx - wmx);
  sum = wave_sum(sum);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other = {__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
    best = better(best, other);
  }
  thr_score = wave_max(thr_score);
  if (lane == 0) { s_max[wave] = wmx; s_sum[wave] = sum; s_bv[wave] = best.v; s_bi[wave] = best.i; s_thr[wave] = thr_score; }
  __syncthreads();
  if (tid == 0) {
    float M = -INFINITY, tot = 0.f, ts = -INFINITY;
    Best bb = {-INFINITY, 0x7fffffff};
    for (int w = 0; w < 4; ++w) M = fmaxf(M, s_max[w]);
    for (int w = 0; w < 4; ++w) {
      tot += (s_max[w] == -INFINITY) ? 0.f : s_sum[w] * __expf(s_max[w] - M);
      bb = better(bb, Best{s_bv[w], s_bi[w]});
      ts = fmaxf(ts, s_thr[w]);
    }
    float* p = part + ((size_t)b * SAMPLE_NB + blk) * 8;
    p[0] = M; p[1] = tot; p[2] = bb.v; p[3] = __int_as_float(bb.i); p[4] = ts;
  }
}

__global__ __launch_bounds__(64) void sample_final_kernel(
    const float* __restrict__ part, int V, const int32_t* __restrict__ stream_slot, int thr_token, int use_thr, float thr_value,
    int eos_token, int32_t* __restrict__ done, int32_t* __restrict__ out_tokens, int32_t* __restrict__ history, int hist_ld,
    int32_t* __restrict__ hist_col, float* __restrict__ scores_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int slot = stream_slot[b];
  if (done != nullptr && done[slot]) return;
  const float* p = part + ((size_t)b * SAMPLE_NB + min(lane, SAMPLE_NB - 1)) * 8;
  const bool have = lane < SAMPLE_NB;
  float m = have ? p[0] : -INFINITY, tot = have ? p[1] : 0.f, ts = have ? p[4] : -INFINITY;
  Best bb = {have ? p[2] : -INFINITY, have ? __float_as_int(p[3]) : 0x7fffffff};
  const float M = wave_max(m);
  tot = wave_sum((m == -INFINITY) ? 0.f : tot * __expf(m - M));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other = {__shfl_xor(bb.v, o, 64), __shfl_xor(bb.i, o, 64)};
    bb = better(bb, other);
  }
  ts = wave_max(ts);
  if (lane != 0) return;
  int tok = bb.i;
  if (thr_token >= 0) {
    bool suppressed = false;
    if (use_thr) suppressed = (__expf(ts - M) / tot) <= thr_value;
    if (suppressed) { if (scores_out) scores_out[(size_t)b * V + thr_token] = -INFINITY; }
    else tok = better(bb, Best{ts, thr_token}).i;
  }
  out_tokens[slot] = tok;
  if (history != nullptr) {
    const int col = hist_col[slot];
    if (col < hist_ld) history[(size_t)slot * hist_ld + col] = tok;
    hist_col[slot] = col + 1;
  }
  if (done != nullptr && tok == eos_token) done[slot] = 1;
}

int sample_greedy(const bf16_t* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream,
                  const int32_t* stream_slot, float repetition_penalty, int thr_token, int use_thr, float thr_value,
                  int eos_token, int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history, int hist_ld,
                  int32_t* hist_col, float* scores_out, float* ws, hipStream_t st) {
  if (B <= 0) return 0;
  if ((V & 31) || (ld & 7) || words_per_stream * 32 < V) return LCC_ERR_SHAPE;
  if (ws != nullptr && V >= 8192) {   // ws: B * 32 * 8 floats of scratch
    sample_partial_kernel<<<dim3(SAMPLE_NB, B), dim3(256), 0, st>>>(logits, ld, V, seen, words_per_stream, stream_slot, repetition_penalty,
                                                                   thr_token, eos_token, suppress_eos, done, ws, scores_out);
    sample_final_kernel<<<dim3(B), dim3(64), 0, st>>>(ws, V, stream_slot, thr_token, use_thr, thr_value, eos_token, done, out_tokens,
                                                      history, hist_ld, hist_col, scores_out);
    return 0;
  }
  sample_greedy_kernel<1024><<<dim3(B), dim3(1024), 0, st>>>(logits, ld, V, seen, words_per_stream, stream_slot,
                                                             repetition_penalty, thr_token, use_thr, thr_value, eos_token,
                                                             suppress_eos, done, out_tokens, history, hist_ld, hist_col, scores_out);
  return 0;
}

// per-stream device counters advanced at the end of a decode step (kv length, rope position, history column)
// per-slot device counters of a consumed token (kv length, rope position): frozen once the slot has emitted EOS
__global__ void advance_kernel(const int32_t* slots, int32_t* kv_len, int32_t* pos, int B, const int32_t* done) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int s = slots[i];
  if (done == nullptr || !done[s]) { kv_len[s] += 1; pos[s] += 1; }
}
int advance_lengths(const int32_t* slots, int32_t* kv_len, int32_t* pos, int B, const int32_t* done, hipStream_t st) {
  if (B <= 0) return 0;
  advance_kernel<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(slots, kv_len, pos, B, done);
  return 0;
}

}  // namespace lcc
