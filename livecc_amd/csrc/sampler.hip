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
    int eos_token, int eos_token2, int suppress_eos, int32_t* __restrict__ done,
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
      if ((suppress_eos & 1) && (id == eos_token || id == eos_token2)) v[e] = -INFINITY;  // MinNewTokensLengthLogitsProcessor
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
    if (done != nullptr && !(suppress_eos & 2) && (tok == eos_token || tok == eos_token2)) done[slot] = 1;
  }
}

// ---- two-stage variant: stage 1 spreads the V-wide pass over SAMPLE_NB blocks per stream, stage 2 merges the partials
constexpr int SAMPLE_NB = 32;

__global__ __launch_bounds__(256) void sample_partial_kernel(
    const bf16_t* __restrict__ logits, int ld, int V, const uint32_t* __restrict__ seen, int words,
    const int32_t* __restrict__ stream_slot, float penalty, int thr_token, int eos_token, int eos_token2, int suppress_eos,
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
      if ((suppress_eos & 1) && (id == eos_token || id == eos_token2)) v[e] = -INFINITY;
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
    int eos_token, int eos_token2, int no_done, int32_t* __restrict__ done, int32_t* __restrict__ out_tokens, int32_t* __restrict__ history, int hist_ld,
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
  if (done != nullptr && !no_done && (tok == eos_token || tok == eos_token2)) done[slot] = 1;
}

// suppress_eos: bit 0 = both EOS ids are masked to -inf (MinNewTokensLengthLogitsProcessor); bit 1 = a picked EOS does not set done[slot]
// (teacher forcing: the forced stream decides where a slot ends, and the processed scores keep HF's definition -- EOS unmasked)
int sample_greedy(const bf16_t* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream,
                  const int32_t* stream_slot, float repetition_penalty, int thr_token, int use_thr, float thr_value,
                  int eos_token, int eos_token2, int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history, int hist_ld,
                  int32_t* hist_col, float* scores_out, float* ws, hipStream_t st) {
  if (B <= 0) return 0;
  if ((V & 31) || (ld & 7) || words_per_stream * 32 < V) return LCC_ERR_SHAPE;
  if (ws != nullptr && V >= 8192) {   // ws: B * 32 * 8 floats of scratch
    sample_partial_kernel<<<dim3(SAMPLE_NB, B), dim3(256), 0, st>>>(logits, ld, V, seen, words_per_stream, stream_slot, repetition_penalty,
                                                                   thr_token, eos_token, eos_token2, suppress_eos, done, ws, scores_out);
    sample_final_kernel<<<dim3(B), dim3(64), 0, st>>>(ws, V, stream_slot, thr_token, use_thr, thr_value, eos_token, eos_token2, suppress_eos & 2, done, out_tokens,
                                                      history, hist_ld, hist_col, scores_out);
    return 0;
  }
  sample_greedy_kernel<1024><<<dim3(B), dim3(1024), 0, st>>>(logits, ld, V, seen, words_per_stream, stream_slot,
                                                             repetition_penalty, thr_token, use_thr, thr_value, eos_token, eos_token2,
                                                             suppress_eos, done, out_tokens, history, hist_ld, hist_col, scores_out);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// General sampling (do_sample=True with top_k != 1): HF's order (generation/utils.py:1171-1330, 2894-2925)
//   repetition penalty -> min-new-tokens EOS mask -> custom processors (ThresholdLogitsProcessor) ->
//   TemperatureLogitsWarper (scores / T) -> TopKLogitsWarper (keep scores >= k-th largest) ->
//   TopPLogitsWarper (ascending cumulative softmax mass <= 1 - top_p removed, the largest always kept) ->
//   softmax -> multinomial.
// One block of 1024 threads per stream, no sort and no atomics: the k-th largest score and the top-p cut are found by
// radix-16 descents over the order-preserving integer key of the fp32 score (8 passes of per-thread REGISTER histograms +
// a block reduction each).  Probability masses are accumulated as 2^40-scaled integers, so every sum is independent of the
// summation order: the same seed gives the same tokens on every run.  The draw is an inverse-CDF walk over a fixed
// (thread-major) order of the vocabulary with one Philox4x32-10 uniform per (seed, slot, draw counter).
// The processed score of id i is recomputed from the bf16 logits + the seen bitmap in every pass (cheaper than an fp32
// scratch row: 304 KB of L2-resident logits per pass).
LCC_DEVICE uint32_t order_key(float x) {   // monotone: x < y  <=>  key(x) < key(y)   (-inf lowest; NaN not expected)
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
LCC_DEVICE void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
LCC_DEVICE unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
    v += ((unsigned long long)hi << 32) | lo;
  }
  return v;
}

struct SampleParams {
  const bf16_t* logits; int ld, V; const uint32_t* seen; int words; const int32_t* stream_slot;
  float penalty; int thr_token, use_thr; float thr_value; int eos_token, eos_token2, suppress_eos;
  int32_t* done; int32_t* out_tokens; int32_t* history; int hist_ld; int32_t* hist_col; float* scores_out;
  float temperature; int top_k; float top_p; uint32_t seed_lo, seed_hi; uint32_t* rng_ctr;
};

constexpr int SNT = 1024;                       // threads per block
constexpr double MASS_SCALE = 1099511627776.0;  // 2^40

__global__ __launch_bounds__(SNT) void sample_topk_topp_kernel(SampleParams P) {
  __shared__ float s_f[SNT / 64][3];
  __shared__ unsigned long long s_hist[16][SNT];   // 128 KB: per-thread radix-16 histograms (the block may use all 160 KB of LDS)
  __shared__ unsigned long long s_tot[16];
  __shared__ unsigned long long s_scan[SNT / 64];
  __shared__ uint32_t s_u[4];
  __shared__ unsigned long long s_ull[2];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = P.stream_slot[b];
  if (P.done != nullptr && P.done[slot]) return;
  const bf16_t* lg = P.logits + (size_t)b * P.ld;
  const uint32_t* sb = P.seen + (size_t)slot * P.words;
  const int V = P.V, nch = V / 8;
  const float penalty = P.penalty;
  const int eos1 = (P.suppress_eos & 1) ? P.eos_token : -1, eos2 = (P.suppress_eos & 1) ? P.eos_token2 : -1;

  // processed scores BEFORE the warpers of the 8 ids of chunk c (penalty, EOS mask; `thr_dead` = threshold processor fired)
  auto load8 = [&](int c, int thr_dead, float (&v)[8]) {
    const u32x4 q = ld16(lg + c * 8);
    const uint32_t bits = (sb[c >> 2] >> ((c & 3) * 8)) & 0xffu;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = lo2f(q[e]); v[2 * e + 1] = hi2f(q[e]); }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (penalty != 1.0f && ((bits >> e) & 1u)) v[e] = v[e] < 0.f ? v[e] * penalty : v[e] / penalty;
      const int id = c * 8 + e;
      if (id == eos1 || id == eos2 || id == thr_dead) v[e] = -INFINITY;
    }
  };

  // ---- pass A: max / exp-sum of the processed scores (ThresholdLogitsProcessor's softmax), score of its token ----
  float mx = -INFINITY, sum = 0.f, thr_score = -INFINITY;
  for (int c = tid; c < nch; c += SNT) {
    float v[8];
    load8(c, -1, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (v[e] > mx) { sum = sum * expf(mx - v[e]) + 1.f; mx = v[e]; }
      else if (v[e] != -INFINITY) sum += expf(v[e] - mx);
      if (c * 8 + e == P.thr_token) thr_score = v[e];
    }
  }
  {
    const float wmx = wave_max(mx);
    sum *= (mx == -INFINITY) ? 0.f : expf(mx - wmx);
    sum = wave_sum(sum);
    thr_score = wave_max(thr_score);
    if (lane == 0) { s_f[wave][0] = wmx; s_f[wave][1] = sum; s_f[wave][2] = thr_score; }
    __syncthreads();
    if (tid == 0) {
      float M = -INFINITY, tot = 0.f, ts = -INFINITY;
      for (int w = 0; w < SNT / 64; ++w) M = fmaxf(M, s_f[w][0]);
      for (int w = 0; w < SNT / 64; ++w) { tot += (s_f[w][0] == -INFINITY) ? 0.f : s_f[w][1] * expf(s_f[w][0] - M); ts = fmaxf(ts, s_f[w][2]); }
      int dead = -1;
      if (P.thr_token >= 0 && P.use_thr && (expf(ts - M) / tot) <= P.thr_value) dead = P.thr_token;
      s_u[0] = (uint32_t)dead;
    }
    __syncthreads();
  }
  const int thr_dead = (int)s_u[0];
  const float invT_is_one = (P.temperature == 1.0f) ? 1.f : 0.f;
  auto warp_t = [&](float v) { return invT_is_one != 0.f ? v : v / P.temperature; };   // TemperatureLogitsWarper: scores / T

  // ---- maximum of the tempered scores (softmax shift; the top element is always kept) ----
  float xm = -INFINITY;
  for (int c = tid; c < nch; c += SNT) {
    float v[8];
    load8(c, thr_dead, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) xm = fmaxf(xm, warp_t(v[e]));
  }
  xm = wave_max(xm);
  __syncthreads();
  if (lane == 0) s_f[wave][0] = xm;
  __syncthreads();
  xm = -INFINITY;
  for (int w = 0; w < SNT / 64; ++w) xm = fmaxf(xm, s_f[w][0]);
  auto mass_of = [&](float x) -> unsigned long long {   // 2^40-scaled softmax numerator, exact integer accumulation
    return x == -INFINITY ? 0ull : (unsigned long long)((double)expf(x - xm) * MASS_SCALE);
  };

  // radix-16 descent.  from_top: the key t with  weight(keys > t) < target <= weight(keys >= t)   (k-th largest, weights 1)
  //                    else    : the smallest key t with  weight(keys <= t) > target               (top-p cut, weights = mass)
  // Only keys >= key_floor take part.  Returns t; *below = weight strictly below/above t on the walked side.
  auto descend = [&](bool from_top, bool use_mass, unsigned long long target, uint32_t key_floor) -> uint32_t {
    uint32_t prefix = 0, mask = 0;
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 28 - 4 * pass;
#pragma unroll
      for (int i = 0; i < 16; ++i) s_hist[i][tid] = 0;          // this thread's private column: bank = tid, no conflicts, no atomics
      for (int c = tid; c < nch; c += SNT) {
        float v[8];
        load8(c, thr_dead, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = warp_t(v[e]);
          const uint32_t key = order_key(x);
          if ((key & mask) != prefix || key < key_floor) continue;
          s_hist[(key >> shift) & 15u][tid] += use_mass ? mass_of(x) : 1ull;
        }
      }
      __syncthreads();
      {   // wave w owns bin w (16 waves, 16 bins): 16 columns per lane, then a wave reduction
        unsigned long long t = 0;
#pragma unroll
        for (int j = 0; j < SNT / 64; ++j) t += s_hist[wave][lane + 64 * j];
        t = wave_sum_u64(t);
        if (lane == 0) s_tot[wave] = t;
      }
      __syncthreads();
      if (tid == 0) {
        const unsigned long long* tot = s_tot;
        int chosen = -1;
        unsigned long long cum = 0;
        if (from_top) {
          for (int i = 15; i >= 0; --i) { if (cum + tot[i] >= target) { chosen = i; break; } cum += tot[i]; }
          if (chosen < 0) { chosen = 0; cum -= tot[0]; }            // fewer eligible elements than k: keep everything
        } else {
          for (int i = 0; i < 16; ++i) { if (cum + tot[i] > target) { chosen = i; break; } cum += tot[i]; }
          if (chosen < 0) { for (int i = 15; i >= 0; --i) if (tot[i]) { chosen = i; break; } if (chosen < 0) chosen = 15; cum = target; }
        }
        s_u[1] = (uint32_t)chosen;
        s_ull[0] = target - cum;
      }
      __syncthreads();
      prefix |= s_u[1] << shift;
      mask |= 15u << shift;
      target = s_ull[0];
    }
    return prefix;
  };

  uint32_t key_floor = 0;                                   // keys below it are removed by the warpers
  if (P.top_k > 0 && P.top_k < V) key_floor = descend(true, false, (unsigned long long)P.top_k, 0u);
  if (P.top_p < 1.0f) {
    // total mass of what top-k kept
    unsigned long long z = 0;
    for (int c = tid; c < nch; c += SNT) {
      float v[8];
      load8(c, thr_dead, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float x = warp_t(v[e]); if (order_key(x) >= key_floor) z += mass_of(x); }
    }
    z = wave_sum_u64(z);
    __syncthreads();
    if (lane == 0) s_scan[wave] = z;
    __syncthreads();
    z = 0;
    for (int w = 0; w < SNT / 64; ++w) z += s_scan[w];
    // remove while ascending cumulative probability <= 1 - top_p  (TopPLogitsWarper), i.e. mass <= (1 - top_p) * Z
    const unsigned long long cut = (unsigned long long)((double)(1.0f - P.top_p) * (double)z);
    const uint32_t kmax = order_key(xm);
    if (cut >= z) {
      // no key has mass(keys <= t) > cut (top_p below ~6e-8: 1.0f - top_p rounds to 1): HF removes everything but the last sorted
      // element (min_tokens_to_keep = 1) -- keep the maximum only (the descent's fallback would land on the SMALLEST key of the top bin)
      key_floor = kmax;
    } else {
      const uint32_t kp = descend(false, true, cut, key_floor);
      key_floor = kp > key_floor ? kp : key_floor;
      if (key_floor > kmax) key_floor = kmax;               // min_tokens_to_keep = 1
    }
  }

  // ---- final pass: masses of the kept set in thread-major order, processed scores out, inverse-CDF draw ----
  float* so = P.scores_out ? P.scores_out + (size_t)b * V : nullptr;
  unsigned long long mine = 0;
  for (int c = tid; c < nch; c += SNT) {
    float v[8];
    load8(c, thr_dead, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = warp_t(v[e]);
      const bool keep = order_key(x) >= key_floor;
      if (so) so[c * 8 + e] = keep ? x : -INFINITY;
      if (keep) mine += mass_of(x);
    }
  }
  // exclusive scan over threads: waves first (shuffles), then the 16 wave totals
  unsigned long long incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t lo = __shfl_up((uint32_t)incl, o, 64), hi = __shfl_up((uint32_t)(incl >> 32), o, 64);
    if (lane >= o) incl += ((unsigned long long)hi << 32) | lo;
  }
  __syncthreads();
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  unsigned long long base = 0, total = 0;
  for (int w = 0; w < SNT / 64; ++w) { if (w < wave) base += s_scan[w]; total += s_scan[w]; }
  if (tid == 0) {
    const uint32_t ctr = P.rng_ctr ? P.rng_ctr[slot] : 0u;
    if (P.rng_ctr) P.rng_ctr[slot] = ctr + 1u;
    uint32_t c4[4] = {ctr, (uint32_t)slot, 0u, 0u};
    philox4x32_10(c4, P.seed_lo, P.seed_hi);
    const double u = (double)((((unsigned long long)c4[0] << 32) | c4[1]) >> 11) * (1.0 / 9007199254740992.0);   // [0,1), 53 bits
    unsigned long long t = (unsigned long long)(u * (double)total);
    if (t >= total) t = total ? total - 1 : 0;
    s_ull[1] = t;
    s_u[2] = 0xffffffffu;
  }
  __syncthreads();
  const unsigned long long tgt = s_ull[1];
  const unsigned long long lo_excl = base + incl - mine;
  if (mine > 0 && tgt >= lo_excl && tgt < lo_excl + mine) {   // exactly one thread owns the target
    unsigned long long cum = lo_excl;
    int tok = -1;
    for (int c = tid; c < nch && tok < 0; c += SNT) {
      float v[8];
      load8(c, thr_dead, v);
      for (int e = 0; e < 8; ++e) {
        const float x = warp_t(v[e]);
        if (order_key(x) < key_floor) continue;
        cum += mass_of(x);
        if (cum > tgt) { tok = c * 8 + e; break; }
      }
    }
    s_u[2] = (uint32_t)tok;
  }
  __syncthreads();
  if (tid == 0) {
    int tok = (int)s_u[2];
    if (tok < 0) tok = 0;                       // unreachable for finite scores; keeps the state machine defined
    P.out_tokens[slot] = tok;
    if (P.history != nullptr) {
      const int col = P.hist_col[slot];
      if (col < P.hist_ld) P.history[(size_t)slot * P.hist_ld + col] = tok;
      P.hist_col[slot] = col + 1;
    }
    if (P.done != nullptr && !(P.suppress_eos & 2) && (tok == P.eos_token || tok == P.eos_token2)) P.done[slot] = 1;
  }
}

int sample_topk_topp(const bf16_t* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream, const int32_t* stream_slot,
                     float repetition_penalty, int thr_token, int use_thr, float thr_value, int eos_token, int eos_token2,
                     int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history, int hist_ld, int32_t* hist_col,
                     float* scores_out, float temperature, int top_k, float top_p, uint64_t seed, uint32_t* rng_ctr, hipStream_t st) {
  if (B <= 0) return 0;
  if ((V & 31) || (ld & 7) || words_per_stream * 32 < V) return LCC_ERR_SHAPE;
  if (!(temperature > 0.f) || top_k < 0 || !(top_p > 0.f) || top_p > 1.f) return LCC_ERR_ARG;
  SampleParams P{logits, ld, V, seen, words_per_stream, stream_slot, repetition_penalty, thr_token, use_thr, thr_value, eos_token,
                 eos_token2, suppress_eos, done, out_tokens, history, hist_ld, hist_col, scores_out, temperature, top_k, top_p,
                 (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), rng_ctr};
  sample_topk_topp_kernel<<<dim3(B), dim3(SNT), 0, st>>>(P);
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
// Teacher forcing (test instrument): the forced stream decides everything, including where it ends.  The sampler ran with EOS suppressed
// (engine_llm.hip: head_and_sample), so `done` still holds the flags from BEFORE this step: a slot whose forced stream has ended stays
// frozen (its last history column is not overwritten by later forced tokens), the others take the forced token and end if it is an EOS.
__global__ void force_tokens_kernel(const int32_t* __restrict__ slots, const int32_t* __restrict__ forced, int B, int32_t* __restrict__ cur_tok,
                                    int32_t* __restrict__ history, int hist_ld, const int32_t* __restrict__ hist_col, int32_t* __restrict__ done,
                                    int eos, int eos2) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int slot = slots[b], tok = forced[b];
  if (done != nullptr && done[slot]) return;
  cur_tok[slot] = tok;
  const int col = hist_col[slot] - 1;          // the sampler has just written column hist_col - 1
  if (col >= 0 && col < hist_ld) history[(size_t)slot * hist_ld + col] = tok;
  if (done != nullptr && (tok == eos || (eos2 >= 0 && tok == eos2))) done[slot] = 1;
}
int force_tokens(const int32_t* slots, const int32_t* forced, int B, int32_t* cur_tok, int32_t* history, int hist_ld, const int32_t* hist_col,
                 int32_t* done, int eos, int eos2, hipStream_t st) {
  force_tokens_kernel<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(slots, forced, B, cur_tok, history, hist_ld, hist_col, done, eos, eos2);
  return 0;
}
int advance_lengths(const int32_t* slots, int32_t* kv_len, int32_t* pos, int B, const int32_t* done, hipStream_t st) {
  if (B <= 0) return 0;
  advance_kernel<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(slots, kv_len, pos, B, done);
  return 0;
}

}  // namespace lcc
