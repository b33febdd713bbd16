// Attention kernels of the LiveCC hot path on gfx950 (flash-style, fp32 online softmax, bf16 MFMA).
//
// Replaces (HF modeling_qwen2_vl.py): VisionAttention 342-422 (non-causal, one segment per temporal slice,
// cu_seqlens from vision_utils.py:42-65), Qwen2VLAttention 537-556 (causal, bottom-right aligned, GQA) for
// prefill and for single-token decode, and the DynamicCache concat (cache_utils.py:127-146) which here is an
// in-place append done by rope_kv_append (elementwise.hip).
//
// Formulation ("swapped QK^T"): one wave owns NQ*16 query columns.  For each 32-key tile
//     S^T[key][q] = K[key][:] . Q[q][:]        A operand = K rows (16-byte loads, K is d-contiguous)
//     O^T[d][q]  += V^T[d][key] . P^T[key][q]   A operand = V^T rows (V is stored blocked-transposed,
//                                               [32-key block][d][32], so these are 16-byte loads too)
// In the C/D layout a lane owns ONE query column (l&15) and 4 keys per MFMA, so the softmax row statistics
// (max / sum / rescale factor) are lane-local plus two cross-lane steps (xor 16, 32), P feeds the second
// MFMA directly from registers (the MFMA row->key map is permuted so a lane ends with 8 consecutive keys),
// and no LDS is used at all: every operand goes HBM/L2 -> VGPR -> MFMA once.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace lcc {

template <int D, int NQ>
struct AttnAcc {
  f32x4 o[D / 16][NQ];
  float m[NQ], l[NQ];
  LCC_DEVICE void init() {
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      m[n] = -INFINITY;
      l[n] = 0.f;
#pragma unroll
      for (int d = 0; d < D / 16; ++d) o[d][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
};

template <int D>
struct KFrag {
  static constexpr int KS = (D + 31) / 32;
  u32x4 v[2][KS];
};

// K rows of the tile starting at key kb.  krow(key) returns the row pointer (already clamped by the caller's
// lambda to a readable row); lanes whose d-chunk lies beyond D load zeros (D = 80: chunks 10, 11).
template <int D, class RowPtr>
LCC_DEVICE void load_k(KFrag<D>& f, RowPtr krow, int kb, int li, int g) {
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = kb + (li >> 2) * 8 + kt * 4 + (li & 3);
    const bf16_t* p = krow(key);
#pragma unroll
    for (int ks = 0; ks < KFrag<D>::KS; ++ks) {
      const int d0 = ks * 32 + g * 8;
      f.v[kt][ks] = (d0 < D) ? ld16(p + d0) : (u32x4){0u, 0u, 0u, 0u};
    }
  }
}

template <int D>
LCC_DEVICE void load_v(u32x4 (&vf)[D / 16], const bf16_t* vt_block, int li, int g) {
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) vf[dt] = ld16(vt_block + (dt * 16 + li) * 32 + g * 8);
}

// one 32-key tile: scores, masking, online softmax, PV accumulation
// key_limit[n] : keys with index >= key_limit[n] are masked for query tile n's column of this lane
template <int D, int NQ>
LCC_DEVICE void attn_tile(AttnAcc<D, NQ>& acc, const KFrag<D>& kf, const u32x4 (&vf)[D / 16],
                          const u32x4 (&qf)[NQ][(D + 31) / 32], int kb, int g, const int (&key_limit)[NQ],
                          float scale_log2e, const int (&min_limit)[NQ]) {
  constexpr int KS = (D + 31) / 32;
  f32x4 s[2][NQ];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a = mfma16(as_bf16x8(kf.v[kt][ks]), as_bf16x8(qf[n][ks]), a);
      s[kt][n] = a;
    }
  bf16x8 pf[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    // Row maximum over the RAW scores (scale > 0), then p = exp2(fma(s, scale, -m)): one FMA per score instead of a multiply
    // and a subtract.  Tiles that lie entirely below this wave's smallest key limit (all but the diagonal / last tile of a
    // long key range) skip the per-score compare + select.  (wave-uniform branch; the VALU work per score is what bounds
    // these kernels: 7 head-waves x 32 MFMAs per 32-key tile leave the SIMDs VALU-limited.)
    const bool masked = kb + 32 > __builtin_amdgcn_readfirstlane(min_limit[n]);
    float mx = -INFINITY;
    if (masked) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb + g * 8 + kt * 4 + r;
          const float v = (key < key_limit[n]) ? s[kt][n][r] : -INFINITY;
          s[kt][n][r] = v;
          mx = fmaxf(mx, v);
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][n][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(acc.m[n], mx * scale_log2e);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(acc.m[n] - m_use);  // acc.m = -inf -> 0
    acc.m[n] = m_new;
    float p[8], psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = exp2f(fmaf(s[kt][n][r], scale_log2e, -m_use));   // masked scores are -inf -> 0
        p[kt * 4 + r] = e;
        psum += e;
      }
    acc.l[n] = acc.l[n] * alpha + psum;
    // lazy rescale: once the running maximum has settled (almost every tile of a long key range) alpha is exactly 1.0 for
    // every lane and the D/16 accumulator multiplies -- the bulk of the per-tile VALU work -- are skipped.  x * 1.0f is exact,
    // so the result is bit-identical to always rescaling.  (wave-uniform branch)
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) acc.o[dt][n] *= alpha;
    }
    u32x4 pk = (u32x4){pack2(p[0], p[1]), pack2(p[2], p[3]), pack2(p[4], p[5]), pack2(p[6], p[7])};
    pf[n] = as_bf16x8(pk);
  }
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
    for (int n = 0; n < NQ; ++n) acc.o[dt][n] = mfma16(as_bf16x8(vf[dt]), pf[n], acc.o[dt][n]);
}

// key-tile loop [t0, t1) with a two-tile register ring: while tile t is multiplied the loads of tile t+1 are already in
// flight and those of tile t+2 are issued right after tile t's operands are consumed.  All loads are unconditional
// (tile index clamped to t1-1, so a re-load of the last tile may be issued and ignored): straight-line code, counted waits.
// wave-wide minimum of the per-lane key limits (tiles entirely below it need no mask)
template <int NQ>
LCC_DEVICE void wave_min_limits(const int (&key_limit)[NQ], int (&min_limit)[NQ]) {
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    int v = key_limit[n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    min_limit[n] = v;
  }
}

struct NoHook { LCC_DEVICE void operator()() const {} };
// `after_first_loads` runs once the first two tiles' loads have been issued (e.g. to build the q fragments while they fly)
template <int D, int NQ, class KRow, class VBlk, class Hook = NoHook>
LCC_DEVICE void attn_loop(AttnAcc<D, NQ>& acc, KRow krow, VBlk vblk, int t0, int t1, const u32x4 (&qf)[NQ][(D + 31) / 32],
                          int li, int g, const int (&key_limit)[NQ], float scale_log2e, Hook after_first_loads = Hook()) {
  if (t0 >= t1) { after_first_loads(); return; }
  int min_limit[NQ];
  wave_min_limits<NQ>(key_limit, min_limit);
  const int tl = t1 - 1;
  KFrag<D> ka, kb;
  u32x4 va[D / 16], vb[D / 16];
  load_k<D>(ka, krow, t0 * 32, li, g);
  load_v<D>(va, vblk(t0), li, g);
  load_k<D>(kb, krow, min(t0 + 1, tl) * 32, li, g);
  load_v<D>(vb, vblk(min(t0 + 1, tl)), li, g);
  after_first_loads();
  for (int t = t0; t < t1; t += 2) {
    attn_tile<D, NQ>(acc, ka, va, qf, t * 32, g, key_limit, scale_log2e, min_limit);
    load_k<D>(ka, krow, min(t + 2, tl) * 32, li, g);
    load_v<D>(va, vblk(min(t + 2, tl)), li, g);
    if (t + 1 < t1) {
      attn_tile<D, NQ>(acc, kb, vb, qf, (t + 1) * 32, g, key_limit, scale_log2e, min_limit);
      load_k<D>(kb, krow, min(t + 3, tl) * 32, li, g);
      load_v<D>(vb, vblk(min(t + 3, tl)), li, g);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ViT attention: non-causal inside each temporal slice (segment).  grid = (ceil(n_tiles/4), heads),
// 4 waves per block, one 32-query tile per wave.  qkv is the [P, 3*E] output of the qkv GEMM with q,k
// already rotated in place; vt is the blocked-transposed V written by vit_rope_vt.
// ------------------------------------------------------------------------------------------------
template <int D, int NQ>
__global__ __launch_bounds__(256, 2) void attn_vit_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out,
    const int32_t* __restrict__ tile_seg, const int32_t* __restrict__ tile_q0,
    const int32_t* __restrict__ seg_start, const int32_t* __restrict__ seg_len,
    const int32_t* __restrict__ seg_blk_start, int n_tiles, int heads, int total_blocks, float scale_log2e) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= n_tiles) return;
  const int h = blockIdx.y;
  const int E = heads * D, ld = 3 * E;
  const int sg = tile_seg[tile], q0 = tile_q0[tile];
  const int s0 = seg_start[sg], sl = seg_len[sg];
  constexpr int KS = (D + 31) / 32;

  u32x4 qf[NQ][KS];
  int key_limit[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    const int ql = min(q0 + n * 16 + li, sl - 1);  // clamp: columns beyond the segment are computed but not stored
    const bf16_t* qp = qkv + (size_t)(s0 + ql) * ld + h * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 32 + g * 8;
      qf[n][ks] = (d0 < D) ? ld16(qp + d0) : (u32x4){0u, 0u, 0u, 0u};
    }
    key_limit[n] = sl;
  }
  const bf16_t* kbase = qkv + (size_t)s0 * ld + E + h * D;
  auto krow = [&](int key) { return kbase + (size_t)min(key, sl - 1) * ld; };
  const bf16_t* vbase = vt + ((size_t)h * total_blocks + seg_blk_start[sg]) * (D * 32);

  AttnAcc<D, NQ> acc;
  acc.init();
  auto vblk = [&](int t) { return vbase + (size_t)t * (D * 32); };
  attn_loop<D, NQ>(acc, krow, vblk, 0, (sl + 31) / 32, qf, li, g, key_limit, scale_log2e);
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    float l = acc.l[n];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    const int ql = q0 + n * 16 + li;
    if (ql < sl) {
      bf16_t* op = out + (size_t)(s0 + ql) * E + h * D;
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) {
        f32x4 o = acc.o[dt][n];
        st8(op + dt * 16 + g * 4, (u32x2){pack2(o[0] * inv, o[1] * inv), pack2(o[2] * inv, o[3] * inv)});
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LLM prefill attention: causal (bottom-right aligned), GQA, K/V read from the per-stream cache that
// already contains this call's new keys.  grid = (ceil(n_tiles/4), n_q_heads).  A tile = NQ*16 consecutive
// new tokens of one stream: tile_stream/tile_q0 (row in the packed q buffer)/tile_nq (valid rows)/
// tile_pos0 (cache index of the first row = past_len + offset).
// ------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(256) void attn_prefill_kernel(
    const bf16_t* __restrict__ q, bf16_t* __restrict__ out, const int32_t* __restrict__ tile_stream,
    const int32_t* __restrict__ tile_q0, const int32_t* __restrict__ tile_nq,
    const int32_t* __restrict__ tile_pos0, bf16_t* const* __restrict__ kv_base, KvLayout lay, int layer,
    int n_tiles, int n_q_heads, float scale_log2e, int nsplit, float* __restrict__ ws_o, float* __restrict__ ws_ml) {
  constexpr int D = 128, KS = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= n_tiles) return;
  const int h = blockIdx.y, hk = h / (n_q_heads / lay.n_kv_heads);
  const int split = blockIdx.z;
  const int strm = tile_stream[tile], q0 = tile_q0[tile], nq = tile_nq[tile], pos0 = tile_pos0[tile];
  const int ldq = n_q_heads * D;

  u32x4 qf[NQ][KS];
  int key_limit[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    const int r = min(n * 16 + li, nq - 1);
    const bf16_t* qp = q + (size_t)(q0 + r) * ldq + h * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[n][ks] = ld16(qp + ks * 32 + g * 8);
    key_limit[n] = pos0 + r + 1;  // causal: keys 0..pos (inclusive)
  }
  const bf16_t* base = kv_base[strm] + (size_t)layer * lay.layer_stride();
  const bf16_t* kbase = base + (size_t)hk * lay.head_stride();
  const bf16_t* vbase = base + lay.kv_stride() + (size_t)hk * lay.head_stride();
  const int kv_n = pos0 + nq;  // keys visible to the last row of this tile
  auto krow = [&](int key) { return kbase + (size_t)min(key, kv_n - 1) * D; };

  AttnAcc<D, NQ> acc;
  acc.init();
  auto vblk = [&](int t) { return vbase + (size_t)t * (D * 32); };
  const int ntile = (kv_n + 31) / 32;
  if (nsplit > 1) {   // key range of this split; partial (o, m, l) go to the workspace, attn_prefill_combine merges them
    const int per = (ntile + nsplit - 1) / nsplit;
    attn_loop<D, NQ>(acc, krow, vblk, split * per, min(ntile, (split + 1) * per), qf, li, g, key_limit, scale_log2e);
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      float l = acc.l[n];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const int r = n * 16 + li;
      if (r < nq) {
        const size_t slot = ((size_t)(q0 + r) * n_q_heads + h) * nsplit + split;
        float* op = ws_o + slot * D;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) *reinterpret_cast<f32x4*>(op + dt * 16 + g * 4) = acc.o[dt][n];
        if (g == 0) { ws_ml[slot * 2] = acc.m[n]; ws_ml[slot * 2 + 1] = l; }
      }
    }
    return;
  }
  attn_loop<D, NQ>(acc, krow, vblk, 0, ntile, qf, li, g, key_limit, scale_log2e);
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    float l = acc.l[n];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    const int r = n * 16 + li;
    if (r < nq) {
      bf16_t* op = out + (size_t)(q0 + r) * ldq + h * D;
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) {
        f32x4 o = acc.o[dt][n];
        st8(op + dt * 16 + g * 4, (u32x2){pack2(o[0] * inv, o[1] * inv), pack2(o[2] * inv, o[3] * inv)});
      }
    }
  }
}

// merges the key splits of the prefill kernel: grid = (Hq, S rows), 128 threads = 4 split groups x 32 lanes (4 d each)
__global__ __launch_bounds__(128) void attn_prefill_combine_kernel(const float* __restrict__ ws_o, const float* __restrict__ ws_ml,
                                                                   bf16_t* __restrict__ out, int n_q_heads, int nsplit) {
  constexpr int D = 128, NG = 4;
  __shared__ float sm[NG][32][6];
  const int h = blockIdx.x, row = blockIdx.y, t = threadIdx.x, dl = t & 31, grp = t >> 5;
  const size_t slot0 = ((size_t)row * n_q_heads + h) * nsplit;
  float M = -INFINITY, den = 0.f;
  f32x4 num = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s = grp; s < nsplit; s += NG) {
    const size_t slot = slot0 + s;
    const float m = ws_ml[slot * 2], l = ws_ml[slot * 2 + 1];
    const f32x4 o = *reinterpret_cast<const f32x4*>(ws_o + slot * D + dl * 4);
    const float Mn = fmaxf(M, m);
    const float a = (M == -INFINITY) ? 0.f : exp2f(M - Mn);
    const float w = (m == -INFINITY) ? 0.f : exp2f(m - Mn);
    num = num * a + o * w;
    den = den * a + w * l;
    M = Mn;
  }
  sm[grp][dl][0] = M; sm[grp][dl][1] = den;
  sm[grp][dl][2] = num[0]; sm[grp][dl][3] = num[1]; sm[grp][dl][4] = num[2]; sm[grp][dl][5] = num[3];
  __syncthreads();
  if (grp != 0) return;
  float MM = -INFINITY;
#pragma unroll
  for (int q = 0; q < NG; ++q) MM = fmaxf(MM, sm[q][dl][0]);
  float dd = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    const float m = sm[q][dl][0];
    const float w = (m == -INFINITY) ? 0.f : exp2f(m - MM);
    dd += w * sm[q][dl][1];
    n0 += w * sm[q][dl][2]; n1 += w * sm[q][dl][3]; n2 += w * sm[q][dl][4]; n3 += w * sm[q][dl][5];
  }
  const float inv = 1.f / dd;
  st8(out + ((size_t)row * n_q_heads + h) * D + dl * 4, (u32x2){pack2(n0 * inv, n1 * inv), pack2(n2 * inv, n3 * inv)});
}

// ------------------------------------------------------------------------------------------------
// v2: K/V tiles shared through LDS.  The per-wave kernels above re-read every K/V byte from L2 once per wave (7 heads x
// 25 query tiles for a 386-row chunk against a 12k cache = 4.3 GB per layer: L2-bandwidth bound).  Here one block =
// NWAVE waves that need the SAME keys: the G query heads of one KV group for the same query tile (prefill, MODE 1), or 4
// consecutive query tiles of one head/segment (ViT, MODE 0).  A 32-key tile (K 2*KS + V^T D/16 one-KB fragment pieces)
// is fetched ONCE per block by LDS-DMA (global_load_lds, per-lane source addresses put every piece in MFMA fragment order
// so the reads are linear ds_read_b128) into a 4-stage ring: two tiles stay in flight while one is multiplied.
//   iteration t:  s_waitcnt vmcnt(2*PW)  ->  s_barrier  ->  issue tile t+3  ->  QK^T / softmax / PV on tile t
// ------------------------------------------------------------------------------------------------
LCC_DEVICE void wait_two_tiles_in_flight(int pw) {   // pw = DMA instructions per wave per tile (wave-uniform)
  switch (pw) {
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

__device__ unsigned int lcc_attn_zero_page[256];

template <int D, int NQ, int MODE, int NWAVE>
LCC_DEVICE void attn_shared_body(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out,
    const int32_t* __restrict__ t0a, const int32_t* __restrict__ t1a, const int32_t* __restrict__ t2a,
    const int32_t* __restrict__ t3a, const int32_t* __restrict__ seg_start, const int32_t* __restrict__ seg_len,
    const int32_t* __restrict__ seg_blk_start, bf16_t* const* __restrict__ kv_base, KvLayout lay, int layer,
    int heads, int total_blocks, float scale_log2e, int nsplit, float* __restrict__ ws_o, float* __restrict__ ws_ml,
    const int bx, const int by, const int bz) {       // the block's grid coordinates, or a virtual block of the persistent walk
  constexpr int KS = (D + 31) / 32, KP = 2 * KS, VP = D / 16, NP = KP + VP, NSTAGE = 4;
  extern __shared__ __attribute__((aligned(16))) u32x4 alds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int nwave = NWAVE, pw = (NP + NWAVE - 1) / NWAVE;   // DMA instructions per wave per tile
  const int li = lane & 15, g = lane >> 4;
  const int grp = bx;

  // ---- what this block streams (K rows, V^T blocks, number of keys) and what this wave computes
  const bf16_t* kbase; size_t kstride; const bf16_t* vbase; int nkeys;
  const bf16_t* qrow0; size_t qstride; int nq_valid; bool active; bf16_t* orow0; size_t ostride;
  int key_limit[NQ];
  int part_row0 = 0, part_head = 0;   // (row, head) of this wave's first query row, for split partials
  if (MODE == 0) {          // ViT: t0a = group segment, t1a = first query row of the group inside the segment
    const int h = by, E = heads * D, ld = 3 * E;
    const int sg = t0a[grp], q0 = t1a[grp] + wave * (NQ * 16);
    const int s0 = seg_start[sg], sl = seg_len[sg];
    kbase = q + (size_t)s0 * ld + E + h * D; kstride = ld;
    vbase = vt + ((size_t)h * total_blocks + seg_blk_start[sg]) * (D * 32);
    nkeys = sl;
    active = q0 < sl;
    nq_valid = max(0, min(NQ * 16, sl - q0));
    qrow0 = q + (size_t)(s0 + min(q0, sl - 1)) * ld + h * D; qstride = ld;
    orow0 = out + (size_t)(s0 + q0) * E + h * D; ostride = E;
#pragma unroll
    for (int n = 0; n < NQ; ++n) key_limit[n] = sl;
  } else {                  // prefill: t0a = stream slot, t1a = first row in q, t2a = valid rows, t3a = cache index of row 0
    const int hk = by, G = heads / lay.n_kv_heads, h = hk * G + wave;
    const int strm = t0a[grp], q0 = t1a[grp], nq = t2a[grp], pos0 = t3a[grp];
    const bf16_t* base = kv_base[strm] + (size_t)layer * lay.layer_stride();
    kbase = base + (size_t)hk * lay.head_stride(); kstride = D;
    vbase = base + lay.kv_stride() + (size_t)hk * lay.head_stride();
    nkeys = pos0 + nq;
    active = wave < G;
    nq_valid = nq;
    const int ldq = heads * D;
    qrow0 = q + (size_t)q0 * ldq + min(h, heads - 1) * D; qstride = ldq;
    orow0 = out + (size_t)q0 * ldq + min(h, heads - 1) * D; ostride = ldq;
    part_row0 = q0; part_head = min(h, heads - 1);
#pragma unroll
    for (int n = 0; n < NQ; ++n) key_limit[n] = pos0 + min(n * 16 + li, nq - 1) + 1;
  }
  const int ntile = (nkeys + 31) / 32;
  // key split (grid z): this block handles key tiles [tb, te)
  const int per = (ntile + nsplit - 1) / nsplit;
  const int tb = (nsplit > 1) ? min(ntile, bz * per) : 0;
  const int te = (nsplit > 1) ? min(ntile, tb + per) : ntile;

  u32x4 qf[NQ][KS];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    const int r = min(n * 16 + li, max(nq_valid - 1, 0));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 32 + g * 8;
      qf[n][ks] = (d0 < D) ? ld16(qrow0 + (size_t)r * qstride + d0) : (u32x4){0u, 0u, 0u, 0u};
    }
  }
#pragma unroll
  for (int n = 0; n < NQ; ++n)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) settle_load(qf[n][ks]);     // before the first LDS-DMA piece (common.h: glds16)

  const bf16_t* zp = reinterpret_cast<const bf16_t*>(lcc_attn_zero_page) + lane * 8;
  // Per-lane source pointers of this wave's pieces for key tile 0 and their per-tile strides are computed ONCE: a DMA issue is
  // then one 64-bit multiply-add per piece.  (The first version recomputed piece -> (row, chunk) -> address in a runtime
  // loop for every tile: ~200 of the ~300 VALU instructions per tile, on a kernel the SQ counters show VALU-issue bound.)
  // Keys past `nkeys` in the last tile: the KV cache (MODE 1) is allocated in whole 32-key tiles, so the rows exist and are
  // masked; the ViT qkv buffer (MODE 0) is not, so its last tile clamps the row index.
  const bf16_t* pbase[pw];
  size_t pstride[pw];
  int pkey0[pw], pd0[pw];
#pragma unroll
  for (int j = 0; j < pw; ++j) {
    const int p = min(j * nwave + wave, NP - 1);   // surplus slots re-fetch the last piece (same bytes, same place)
    if (p < KP) {
      const int kt = p / KS, ks = p - kt * KS;
      pkey0[j] = (li >> 2) * 8 + kt * 4 + (li & 3);
      pd0[j] = ks * 32 + g * 8;
      pbase[j] = (pd0[j] < D) ? kbase + (size_t)pkey0[j] * kstride + pd0[j] : zp;
      pstride[j] = (pd0[j] < D) ? (size_t)32 * kstride : 0;
    } else {
      pkey0[j] = -1; pd0[j] = 0;
      pbase[j] = vbase + ((p - KP) * 16 + li) * 32 + g * 8;
      pstride[j] = (size_t)D * 32;
    }
  }
  auto issue = [&](int t) {
    const int tc = min(t, ntile - 1);
    u32x4* sbase = alds + ((t - tb) % NSTAGE) * (NP * 64);
#pragma unroll
    for (int j = 0; j < pw; ++j) {
      const int p = min(j * nwave + wave, NP - 1);
      const bf16_t* src = pbase[j] + (size_t)tc * pstride[j];
      if (MODE == 0 && tc == ntile - 1 && pkey0[j] >= 0 && pd0[j] < D)
        src = kbase + (size_t)min(tc * 32 + pkey0[j], nkeys - 1) * kstride + pd0[j];
      glds16(src, lds_addr(sbase + p * 64));
    }
  };

  AttnAcc<D, NQ> acc;
  acc.init();
  int min_limit[NQ];
  wave_min_limits<NQ>(key_limit, min_limit);
  // Two key tiles per barrier: the ring's four stages form two pair-stages.  After the barrier of pair P (its DMAs have landed,
  // every wave is done with pair P-1) the DMAs of pair P+1 are issued into the stages pair P-1 used and fly under the
  // QK^T / softmax / PV of both tiles of pair P -- half the barriers and waits of the one-tile-per-barrier loop.
  auto compute = [&](int t) {
    const u32x4* s = alds + ((t - tb) % NSTAGE) * (NP * 64);
    KFrag<D> kf;
    u32x4 vf[VP];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf.v[kt][ks] = s[(kt * KS + ks) * 64 + lane];
#pragma unroll
    for (int dt = 0; dt < VP; ++dt) vf[dt] = s[(KP + dt) * 64 + lane];
    attn_tile<D, NQ>(acc, kf, vf, qf, t * 32, g, key_limit, scale_log2e, min_limit);
  };
  issue(tb); issue(tb + 1);
  for (int t = tb; t < te; t += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(t + 2); issue(t + 3);
    if (active) {
      compute(t);
      if (t + 1 < te) compute(t + 1);
    }
  }
  if (!active) return;
  if (MODE == 1 && nsplit > 1) {
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      float l = acc.l[n];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const int r = n * 16 + li;
      if (r < nq_valid) {
        const size_t slot = ((size_t)(part_row0 + r) * heads + part_head) * nsplit + bz;
        float* op = ws_o + slot * D;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) *reinterpret_cast<f32x4*>(op + dt * 16 + g * 4) = acc.o[dt][n];
        if (g == 0) { ws_ml[slot * 2] = acc.m[n]; ws_ml[slot * 2 + 1] = l; }
      }
    }
    return;
  }
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    float l = acc.l[n];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    const int r = n * 16 + li;
    if (r < nq_valid) {
      bf16_t* op = orow0 + (size_t)r * ostride;
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) {
        f32x4 o = acc.o[dt][n];
        st8(op + dt * 16 + g * 4, (u32x2){pack2(o[0] * inv, o[1] * inv), pack2(o[2] * inv, o[3] * inv)});
      }
    }
  }
}

// kernel: one block per (x, y, z) grid point, or -- vgx > 0 (the ViT launches under the grid cap, gemm.hip: g_grid_cap) -- a persistent walk over
// the vgx x vgy virtual blocks
template <int D, int NQ, int MODE, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 2) void attn_shared_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out,
    const int32_t* __restrict__ t0a, const int32_t* __restrict__ t1a, const int32_t* __restrict__ t2a,
    const int32_t* __restrict__ t3a, const int32_t* __restrict__ seg_start, const int32_t* __restrict__ seg_len,
    const int32_t* __restrict__ seg_blk_start, bf16_t* const* __restrict__ kv_base, KvLayout lay, int layer,
    int heads, int total_blocks, float scale_log2e, int nsplit, float* __restrict__ ws_o, float* __restrict__ ws_ml, int vgx, int vgy) {
  if (vgx <= 0) {
    attn_shared_body<D, NQ, MODE, NWAVE>(q, vt, out, t0a, t1a, t2a, t3a, seg_start, seg_len, seg_blk_start, kv_base, lay, layer, heads, total_blocks,
                                         scale_log2e, nsplit, ws_o, ws_ml, blockIdx.x, blockIdx.y, blockIdx.z);
    return;
  }
  const int nvb = vgx * vgy;
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    attn_shared_body<D, NQ, MODE, NWAVE>(q, vt, out, t0a, t1a, t2a, t3a, seg_start, seg_len, seg_blk_start, kv_base, lay, layer, heads, total_blocks,
                                         scale_log2e, nsplit, ws_o, ws_ml, vb % vgx, vb / vgx, 0);
    __syncthreads();     // the next virtual block's DMA ring reuses the stages (every wave returns from the body)
  }
}

// ------------------------------------------------------------------------------------------------
// LLM decode attention: one new token per stream.  The G = Hq/Hkv query heads that share a KV head are the
// query columns of one wave (G <= 16), so every K/V byte is read once per KV head.  grid = (nsplit, Hkv, B),
// one wave per block; split s handles key tiles [s*per, (s+1)*per).  HBM-bound: L*512 bytes per KV head.
// Partial (o, m, l) go to a workspace, attn_decode_combine merges the splits.
// ------------------------------------------------------------------------------------------------
template <bool DIRECT>
// __launch_bounds__(64, 2): with one wave per SIMD allowed (512 registers) hipcc selects the AGPR form of the MFMAs and moves the score / output
// accumulators to and from the vector registers around the softmax -- 88 v_accvgpr_read / write per key tile inside the loop (round-6 audit of
// the .s files, tools/audit_agpr_copies.py); a minimum of two waves per SIMD caps the kernel at 256 registers (it uses 220) and the MFMAs
// take their accumulators in VGPRs.
__global__ __launch_bounds__(64, 2) void attn_decode_kernel(
    const bf16_t* __restrict__ q, const int32_t* __restrict__ slots, const int32_t* __restrict__ kv_len,
    bf16_t* const* __restrict__ kv_base,
    KvLayout lay, int layer, int n_q_heads, int nsplit, float* __restrict__ ws_o, float* __restrict__ ws_ml,
    float scale_log2e, AttnDirect dir) {
  constexpr int D = 128, KS = 4, NQ = 1;
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const int G = n_q_heads / lay.n_kv_heads;
  const int slot_id = DIRECT ? 0 : slots[b];
  // the new token's K/V were appended at index kv_len[slot] by rope_kv_append.  DIRECT: the host's count (kernels.h: AttnDirect)
  const int n = DIRECT ? (b == 0 ? dir.n[0] : (b == 1 ? dir.n[1] : (b == 2 ? dir.n[2] : dir.n[3]))) : kv_len[slot_id] + 1;
  const int ntile = (n + 31) / 32;
  const int per = (ntile + nsplit - 1) / nsplit;
  const int t0 = split * per, t1 = min(ntile, t0 + per);

  u32x4 qf[NQ][KS];
  int key_limit[NQ] = {n};
  {
    const int hq = hk * G + min(li, G - 1);
    const bf16_t* qp = q + ((size_t)b * n_q_heads + hq) * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[0][ks] = ld16(qp + ks * 32 + g * 8);
  }
  const bf16_t* base = (DIRECT ? (b == 0 ? dir.base[0] : (b == 1 ? dir.base[1] : (b == 2 ? dir.base[2] : dir.base[3]))) : kv_base[slot_id]) +
                       (size_t)layer * lay.layer_stride();
  const bf16_t* kbase = base + (size_t)hk * lay.head_stride();
  const bf16_t* vbase = base + lay.kv_stride() + (size_t)hk * lay.head_stride();
  auto krow = [&](int key) { return kbase + (size_t)min(key, n - 1) * D; };

  AttnAcc<D, NQ> acc;
  acc.init();
  auto vblk = [&](int t) { return vbase + (size_t)t * (D * 32); };
  attn_loop<D, NQ>(acc, krow, vblk, t0, t1, qf, li, g, key_limit, scale_log2e);
  float l = acc.l[0];
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (li < G) {
    const size_t slot = (((size_t)b * lay.n_kv_heads + hk) * nsplit + split) * 16 + li;
    float* op = ws_o + slot * D;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) *reinterpret_cast<f32x4*>(op + dt * 16 + g * 4) = acc.o[dt][0];
    if (g == 0) {
      ws_ml[slot * 2 + 0] = acc.m[0];
      ws_ml[slot * 2 + 1] = l;
    }
  }
}

// (Round 6, measured slower and removed: a multi-wave form -- 8 waves per block sharing a key range, merged through LDS, so that 8 / 16 key splits
// instead of ~51 leave few enough partials to merge them in the o-projection's prologue instead of a combine launch.  Decode step 2930 -> 3134 us
// with 8 splits (32 blocks), 3012 with 16, 3390 with 4: a CU pulls ~35 GB/s of K / V however many waves it runs, so the 13.4 MB of a launch
// need all 204 CUs' worth of one-wave blocks -- and with it the 51 partials.  profiles/r06/decode_attn_multiwave_ab.jsonl.)
// grid = (Hq, B), 256 threads = 8 split groups x 32 lanes (4 consecutive d each).  Group q merges splits q, q+8, ...
// with all its loads in flight at once; the 8 partial (m, num, den) triples are merged through LDS.
__global__ __launch_bounds__(256) void attn_decode_combine_kernel(
    const float* __restrict__ ws_o, const float* __restrict__ ws_ml, bf16_t* __restrict__ out, int n_q_heads,
    int n_kv_heads, int nsplit) {
  constexpr int D = 128, NG = 8;
  __shared__ float sm[NG][32][6];
  const int hq = blockIdx.x, b = blockIdx.y, t = threadIdx.x, dl = t & 31, grp = t >> 5;
  const int G = n_q_heads / n_kv_heads, hk = hq / G, j = hq % G;
  const size_t slot0 = (((size_t)b * n_kv_heads + hk) * nsplit) * 16 + j;
  float M = -INFINITY, den = 0.f;
  f32x4 num = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int s = grp; s < nsplit; s += NG) {
    const size_t slot = slot0 + (size_t)s * 16;
    const float m = ws_ml[slot * 2], l = ws_ml[slot * 2 + 1];
    const f32x4 o = *reinterpret_cast<const f32x4*>(ws_o + slot * D + dl * 4);
    const float Mn = fmaxf(M, m);
    const float a = (M == -INFINITY) ? 0.f : exp2f(M - Mn);
    const float w = (m == -INFINITY) ? 0.f : exp2f(m - Mn);
    num = num * a + o * w;
    den = den * a + w * l;
    M = Mn;
  }
  sm[grp][dl][0] = M; sm[grp][dl][1] = den;
  sm[grp][dl][2] = num[0]; sm[grp][dl][3] = num[1]; sm[grp][dl][4] = num[2]; sm[grp][dl][5] = num[3];
  __syncthreads();
  if (grp != 0) return;
  float MM = -INFINITY;
#pragma unroll
  for (int q = 0; q < NG; ++q) MM = fmaxf(MM, sm[q][dl][0]);
  float dd = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    const float m = sm[q][dl][0];
    const float w = (m == -INFINITY) ? 0.f : exp2f(m - MM);
    dd += w * sm[q][dl][1];
    n0 += w * sm[q][dl][2]; n1 += w * sm[q][dl][3]; n2 += w * sm[q][dl][4]; n3 += w * sm[q][dl][5];
  }
  const float inv = 1.f / dd;
  st8(out + ((size_t)b * n_q_heads + hq) * D + dl * 4, (u32x2){pack2(n0 * inv, n1 * inv), pack2(n2 * inv, n3 * inv)});
}

// ------------------------------------------------------------------------------------------------
// LLM decode attention, FUSED form: [bias + M-RoPE + KV append]  ->  attention  ->  [split merge]  in ONE launch.
// Replaces the rope_kv_append -> attn_decode -> attn_decode_combine chain of the batch decode step (three latency-bound
// launches, ~18 us per layer at 7B) by one kernel:
//   * prologue: every wave rebuilds the roped q fragments of its KV head's G query heads straight from the fp32 split-K slabs
//     of the qkv GEMV (+ bias, HF rounding points of Q2VL:180-222), so q never goes through HBM as bf16;
//   * the block that owns the last key tile appends the new token's roped K row and (blocked-transposed) V column to the
//     cache first (plain stores -> vmcnt(0) -> __syncthreads: same-CU visibility; no other block reads that tile);
//   * 4 waves per block share the block's key range; their (o, m, l) are merged through LDS;
//   * nsplit > 1: the block partial is published with write-through (sc1) stores, one relaxed agent-scope ticket per block;
//     the block that draws the last ticket of its (stream, KV head) acquires once and merges the nsplit partials
//     (placement-independent protocol of cdna_hip_programming.md G16; the counter is left at zero for the next launch).
// grid = (nsplit, Hkv, B), 256 threads.  G = Hq / Hkv <= 16.
// ------------------------------------------------------------------------------------------------
template <int NS>
LCC_DEVICE void qkv8_from_slabs(const float* __restrict__ part, size_t slab_stride, size_t off, const bf16_t* __restrict__ bias,
                                int col, float (&v)[8]) {
  f32x4 pa[NS], pb[NS];
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) {
    const float* pp = part + (size_t)sp * slab_stride + off + col;
    pa[sp] = *reinterpret_cast<const f32x4*>(pp);
    pb[sp] = *reinterpret_cast<const f32x4*>(pp + 4);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp)
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += pa[sp][e]; v[4 + e] += pb[sp][e]; }
  const u32x4 bq = ld16(bias + col);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = rbf(v[2 * e] + lo2f(bq[e])); v[2 * e + 1] = rbf(v[2 * e + 1] + hi2f(bq[e])); }
}

// rotate one (x[c0..c0+8), x[c0+64..c0+72)) pair of a head with the token's cos/sin row (64 bf16 each)
LCC_DEVICE void rope_pair8(const float (&x1)[8], const float (&x2)[8], const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn,
                           int c0, u32x4& lo, u32x4& hi) {
  const u32x4 cq = ld16(cs + c0), sq = ld16(sn + c0);
  float o1[8], o2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float c = (e & 1) ? hi2f(cq[e >> 1]) : lo2f(cq[e >> 1]);
    const float sv = (e & 1) ? hi2f(sq[e >> 1]) : lo2f(sq[e >> 1]);
    o1[e] = rbf(x1[e] * c) + rbf(-x2[e] * sv);
    o2[e] = rbf(x2[e] * c) + rbf(x1[e] * sv);
  }
  lo = (u32x4){pack2(o1[0], o1[1]), pack2(o1[2], o1[3]), pack2(o1[4], o1[5]), pack2(o1[6], o1[7])};
  hi = (u32x4){pack2(o2[0], o2[1]), pack2(o2[2], o2[3]), pack2(o2[4], o2[5]), pack2(o2[6], o2[7])};
}

LCC_DEVICE void store_sc1_f32x4(float* p, f32x4 v) {   // write-through (sc1) 16 bytes as two 8-byte agent-scope atomic stores
  unsigned long long* p8 = reinterpret_cast<unsigned long long*>(p);
  const unsigned long long lo = (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32);
  const unsigned long long hi = (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32);
  __hip_atomic_store(p8, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p8 + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NS, int TAIL, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void attn_decode_fused_kernel(
    const float* __restrict__ qkv_part, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ cs_tab,
    const bf16_t* __restrict__ sn_tab, const int32_t* __restrict__ slots, const int32_t* __restrict__ kv_len,
    bf16_t* const* __restrict__ kv_base, KvLayout lay, int layer, int B, int n_q_heads, int nsplit,
    float* __restrict__ ws_o, float* __restrict__ ws_ml, int32_t* __restrict__ counters, bf16_t* __restrict__ out,
    float scale_log2e) {
  constexpr int D = 128, KS = 4, NQ = 1;
  // LDS: per wave, per query column (16): 128 o values + m + l (row stride 528 B keeps the 16-byte o accesses aligned);
  // the first 512 bytes double as the staging area of the new token's K row / V column and later as the merge flag
  extern __shared__ __attribute__((aligned(16))) float sm_dyn[];     // NW * 16 * (D + 4) floats (8 waves: 67.6 KB, above the static limit)
  float (*sm)[16][D + 4] = reinterpret_cast<float (*)[16][D + 4]>(sm_dyn);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const int hkv = lay.n_kv_heads, G = n_q_heads / hkv;
  const int ld = (n_q_heads + 2 * hkv) * D;
  const size_t slab_stride = (size_t)B * ld, row_off = (size_t)b * ld;
  const int slot_id = slots[b];
  const int nrow = kv_len[slot_id];        // cached keys; the new token's K/V take cache index nrow
  const int ntile = (nrow + 31) / 32;      // tiles of CACHED keys, split over the blocks; the new key is a register tile
  const int per = (ntile + nsplit - 1) / nsplit;
  const int t0 = split * per, t1 = min(ntile, t0 + per);
  const bf16_t* cs = cs_tab + (size_t)b * 64;
  const bf16_t* sn = sn_tab + (size_t)b * 64;
  bf16_t* base = kv_base[slot_id] + (size_t)layer * lay.layer_stride();
  bf16_t* knew = reinterpret_cast<bf16_t*>(&sm[0][0][0]);   // [128] roped K row of the new token
  bf16_t* vnew = knew + D;                                   // [128] V row of the new token

  // ---- split 0 appends the new token's K/V (stores are fire-and-forget: nobody reads them in this launch) and stages them
  //      in LDS for its own wave 0, which scores the new key from registers ----
  if (split == 0) {
    const int t = threadIdx.x;
    if (t < 8) {            // K: 8 (c0, c0+64) pairs
      const int c0 = t * 8;
      float x1[8], x2[8];
      qkv8_from_slabs<NS>(qkv_part, slab_stride, row_off, bias, (n_q_heads + hk) * D + c0, x1);
      qkv8_from_slabs<NS>(qkv_part, slab_stride, row_off, bias, (n_q_heads + hk) * D + c0 + 64, x2);
      u32x4 lo, hi;
      rope_pair8(x1, x2, cs, sn, c0, lo, hi);
      bf16_t* dst = base + (size_t)hk * lay.head_stride() + (size_t)nrow * D;
      st16(dst + c0, lo);
      st16(dst + c0 + 64, hi);
      st16(knew + c0, lo);
      st16(knew + c0 + 64, hi);
    } else if (t >= 64 && t < 80) {   // V: 16 chunks of 8 d
      const int c0 = (t - 64) * 8;
      float x[8];
      qkv8_from_slabs<NS>(qkv_part, slab_stride, row_off, bias, (n_q_heads + hkv + hk) * D + c0, x);
      bf16_t* dst = base + lay.kv_stride() + (size_t)hk * lay.head_stride() + ((size_t)(nrow >> 5) * D + c0) * 32 + (nrow & 31);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const bf16_t h = f2bf(x[e]); dst[e * 32] = h; vnew[c0 + e] = h; }
    }
  }

  // ---- attention over this wave's share of the block's cached key tiles; the roped q fragments of the KV head's query heads
  //      (lane li -> head hk*G + min(li, G-1)) are rebuilt from the slabs while the first K/V tiles are in flight ----
  u32x4 qf[NQ][KS];
  auto build_q = [&]() {
    const int hq = hk * G + min(li, G - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c0 = ks * 32 + g * 8;
      float x1[8], x2[8];
      qkv8_from_slabs<NS>(qkv_part, slab_stride, row_off, bias, hq * D + c0, x1);
      qkv8_from_slabs<NS>(qkv_part, slab_stride, row_off, bias, hq * D + c0 + 64, x2);
      rope_pair8(x1, x2, cs, sn, c0, qf[0][ks], qf[0][ks + 2]);
    }
  };
  const int nt = max(t1 - t0, 0), pw = (nt + NW - 1) / NW;
  const int w0 = t0 + wave * pw, w1 = min(t1, w0 + pw);
  const bf16_t* kbase = base + (size_t)hk * lay.head_stride();
  const bf16_t* vbase = base + lay.kv_stride() + (size_t)hk * lay.head_stride();
  auto krow = [&](int key) { return kbase + (size_t)max(min(key, nrow - 1), 0) * D; };
  auto vblk = [&](int t) { return vbase + (size_t)t * (D * 32); };
  int key_limit[NQ] = {nrow};
  AttnAcc<D, NQ> acc;
  acc.init();
  attn_loop<D, NQ>(acc, krow, vblk, w0, w1, qf, li, g, key_limit, scale_log2e, build_q);

  if (split == 0) {
    __syncthreads();          // knew / vnew staged (block-uniform branch)
    if (wave == 0) {
      // the new key as a one-key register tile: A-operand row 0 of the K fragment, column 0 of the V^T fragment
      KFrag<D> kf;
      u32x4 vf[D / 16];
      const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kf.v[0][ks] = (li == 0) ? *reinterpret_cast<const u32x4*>(knew + ks * 32 + g * 8) : z;
        kf.v[1][ks] = z;
      }
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) vf[dt] = (g == 0) ? (u32x4){(unsigned)vnew[dt * 16 + li], 0u, 0u, 0u} : z;
      const int one[NQ] = {1};
      attn_tile<D, NQ>(acc, kf, vf, qf, 0, g, one, scale_log2e, one);
    }
    __syncthreads();          // staging area is reused by the merge below
  }
  float l = acc.l[0];
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) *reinterpret_cast<f32x4*>(&sm[wave][li][dt * 16 + g * 4]) = acc.o[dt][0];
  if (g == 0) { sm[wave][li][D] = acc.m[0]; sm[wave][li][D + 1] = l; }
  __syncthreads();

  // ---- merge the NW waves: item = (query head j, 4 consecutive d) ----
  const bool single = nsplit == 1;
  for (int item = threadIdx.x; item < G * 32; item += NW * 64) {
    const int j = item >> 5, d4 = (item & 31) * 4;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, sm[w][j][D]);
    float den = 0.f;
    f32x4 num = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float m = sm[w][j][D];
      const float wt = (m == -INFINITY) ? 0.f : exp2f(m - M);
      den += wt * sm[w][j][D + 1];
      num += *reinterpret_cast<const f32x4*>(&sm[w][j][d4]) * wt;
    }
    const size_t slot = (((size_t)b * hkv + hk) * nsplit + split) * 16 + j;
    if (single) {
      const float inv = 1.f / den;
      st8(out + ((size_t)b * n_q_heads + hk * G + j) * D + d4, (u32x2){pack2(num[0] * inv, num[1] * inv), pack2(num[2] * inv, num[3] * inv)});
    } else if (TAIL == 1) {     // plain partials, merged by attn_decode_combine_kernel (next launch)
      *reinterpret_cast<f32x4*>(ws_o + slot * D + d4) = num;
      if ((item & 31) == 0) { ws_ml[slot * 2] = M; ws_ml[slot * 2 + 1] = den; }
    } else {
      store_sc1_f32x4(ws_o + slot * D + d4, num);
      if ((item & 31) == 0) {
        unsigned long long ml = (unsigned long long)__float_as_uint(M) | ((unsigned long long)__float_as_uint(den) << 32);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(ws_ml + slot * 2), ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (single || TAIL == 1) return;

  // ---- publish / last-arriver merge ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                    // also: everyone is done reading sm
  float* flag = &sm[0][0][0];
  if (threadIdx.x == 0) {
    int32_t* cnt = counters + b * hkv + hk;
    const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == nsplit - 1;
    if (last) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    flag[0] = last ? 1.f : 0.f;
  }
  __syncthreads();
  if (flag[0] == 0.f) return;
  for (int item = threadIdx.x; item < G * 32; item += NW * 64) {
    const int j = item >> 5, d4 = (item & 31) * 4;
    const size_t slot0 = (((size_t)b * hkv + hk) * nsplit) * 16 + j;
    float M = -INFINITY, den = 0.f;
    f32x4 num = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nsplit; s0 += 4) {
      float m4[4], l4[4];
      f32x4 o4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {      // 4 partials in flight
        const size_t slot = slot0 + (size_t)min(s0 + u, nsplit - 1) * 16;
        m4[u] = ws_ml[slot * 2];
        l4[u] = ws_ml[slot * 2 + 1];
        o4[u] = *reinterpret_cast<const f32x4*>(ws_o + slot * D + d4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (s0 + u < nsplit) {
          const float Mn = fmaxf(M, m4[u]);
          const float a = (M == -INFINITY) ? 0.f : exp2f(M - Mn);
          const float wt = (m4[u] == -INFINITY) ? 0.f : exp2f(m4[u] - Mn);
          num = num * a + o4[u] * wt;
          den = den * a + wt * l4[u];
          M = Mn;
        }
      }
    }
    const float inv = 1.f / den;
    st8(out + ((size_t)b * n_q_heads + hk * G + j) * D + d4, (u32x2){pack2(num[0] * inv, num[1] * inv), pack2(num[2] * inv, num[3] * inv)});
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline float scale_l2e(int d) { return 1.4426950408889634f / sqrtf((float)d); }
// 0: per-wave kernels everywhere (operands straight from L2);  1: prefill = LDS-shared kernel + key split, ViT = per-wave
// kernel;  2: LDS-shared for both;  3 (default since round 3): 2 with the LLM prefill on 32-row tiles / 32x32x16 MFMAs (attn32.hip:
// 373 vs 522 us at 8 x 386 rows x 6k keys, 1.36 vs 1.93 ms for a 4,096-row piece against 20k keys).  (With the per-tile DMA address math hoisted out of the key loop the shared ViT
// kernel went from 75 us -- slower than the 64-us per-wave kernel -- to on par for one stream and +1.7 % end to end at 8.)
static int g_attn_variant = 3;
void set_attn_variant(int v) { g_attn_variant = v; }
int get_attn_variant() { return g_attn_variant; }

template <class Kern>
static void set_lds_attr(Kern k, size_t bytes) {
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int attn_vit_bf16(const bf16_t* qkv, const bf16_t* vt, bf16_t* out, const int32_t* tile_seg, const int32_t* tile_q0,
                  const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_tiles,
                  int heads, int total_blocks, const int32_t* grp_seg, const int32_t* grp_q0, int n_groups, hipStream_t st) {
  if (n_tiles <= 0) return 0;
  if (g_attn_variant >= 2 && n_groups > 0) {
    constexpr size_t lds = (size_t)4 * (6 + 5) * 1024;
    static DeviceOnce once;
    if (once.first()) { set_lds_attr(attn_shared_kernel<80, 2, 0, 4>, lds); set_lds_attr(attn_shared_kernel<80, 1, 0, 8>, lds); }
    // a group = 128 query rows of one head: 4 waves x 32 rows, or 8 waves x 16 rows.  The per-wave chain QK^T -> softmax -> PV is
    // latency-bound; with few blocks on the chip (one stream's 2-frame chunk: 12 groups x 16 heads = 192 blocks, one wave per SIMD)
    // two shorter chains per SIMD overlap where one long one cannot: 254.4 -> 258.3 tok/s single stream without the ViT prefetch
    // (neutral with it: the other stream's waves already fill the gaps).  LCC_VIT_ATTN_WAVES=4|8 forces one.
    static const int forced = [] { const char* v = getenv("LCC_VIT_ATTN_WAVES"); return v ? atoi(v) : 0; }();
    const int waves = forced ? forced : ((long)n_groups * heads <= 512 ? 8 : 4);
    const int cap = get_grid_cap();
    const bool capped = cap > 0 && (long)n_groups * heads > cap;
    const int vgx = capped ? n_groups : 0, vgy = capped ? heads : 0;
    const dim3 grid = capped ? dim3(cap) : dim3(n_groups, heads);
    if (waves == 8)
      attn_shared_kernel<80, 1, 0, 8><<<grid, dim3(512), lds, st>>>(
          qkv, vt, out, grp_seg, grp_q0, nullptr, nullptr, seg_start, seg_len, seg_blk_start, nullptr, KvLayout{0, 1, 32, 80}, 0,
          heads, total_blocks, scale_l2e(80), 1, nullptr, nullptr, vgx, vgy);
    else
      attn_shared_kernel<80, 2, 0, 4><<<grid, dim3(256), lds, st>>>(
          qkv, vt, out, grp_seg, grp_q0, nullptr, nullptr, seg_start, seg_len, seg_blk_start, nullptr, KvLayout{0, 1, 32, 80}, 0,
          heads, total_blocks, scale_l2e(80), 1, nullptr, nullptr, vgx, vgy);
    return 0;
  }
  attn_vit_kernel<80, 2><<<dim3((n_tiles + 3) / 4, heads), dim3(256), 0, st>>>(
      qkv, vt, out, tile_seg, tile_q0, seg_start, seg_len, seg_blk_start, n_tiles, heads, total_blocks,
      scale_l2e(80));
  return 0;
}

int attn_prefill_bf16(const bf16_t* q, bf16_t* out, const int32_t* tile_stream, const int32_t* tile_q0,
                      const int32_t* tile_nq, const int32_t* tile_pos0, bf16_t* const* kv_base, KvLayout lay,
                      int layer, int n_tiles, int n_q_heads, int tile_rows, int nsplit, int n_rows, float* ws_o, float* ws_ml,
                      hipStream_t st) {
  if (n_tiles <= 0) return 0;
  if (lay.head_dim != 128 || (lay.lmax & 31) || lay.n_kv_heads < 1 || n_q_heads % lay.n_kv_heads) return LCC_ERR_SHAPE;
  if (nsplit > 1 && (!ws_o || !ws_ml || nsplit > 16)) return LCC_ERR_ARG;
  const int G = n_q_heads / lay.n_kv_heads;
  // tile_rows: 16 or 32 for every kernel; the 32x32x16 kernel also takes the taller tiles its pair packing can fill (36 rows at G = 7)
  const bool tall32 = g_attn_variant == 3 && G <= 8 && tile_rows > 32 && tile_rows <= attn32_max_tile_rows(G);
  if (tile_rows != 16 && tile_rows != 32 && !tall32) return LCC_ERR_SHAPE;
  const int S = nsplit > 1 ? nsplit : 1;
  // Default for prefill: the LDS-shared kernel (the G heads of a KV group fetch every K/V tile once: G x less L2/TA traffic,
  // which bounds the per-wave kernel at ~17 TB/s of 64-byte row segments) TOGETHER with the key split (which gives every SIMD
  // 2-3 waves for the dependent MFMA->softmax->MFMA chain).  g_attn_variant 0 forces the per-wave kernel.
  g_launch_counts[LC_LAST_PREFILL_NSPLIT] = S;
  if (S > 1) g_launch_counts[LC_ATTN_PREFILL_COMBINE]++;
  if (g_attn_variant == 3 && tile_rows >= 32 && G <= 8) {   // 32x32x16 MFMA kernel (attn32.hip)
    g_launch_counts[LC_ATTN_PREFILL_MFMA32]++;
    if (int rc = attn_prefill32_launch(q, out, tile_stream, tile_q0, tile_nq, tile_pos0, kv_base, lay, layer, n_tiles, n_q_heads, S, ws_o,
                                       ws_ml, scale_l2e(128), st)) return rc;
    if (S > 1) attn_prefill_combine_kernel<<<dim3(n_q_heads, n_rows), dim3(128), 0, st>>>(ws_o, ws_ml, out, n_q_heads, S);
    return 0;
  }
  if (g_attn_variant != 0 && G >= 2 && G <= 8) {
    g_launch_counts[LC_ATTN_PREFILL_SHARED]++;
    constexpr size_t lds = (size_t)4 * 16 * 1024;
    const dim3 grid(n_tiles, lay.n_kv_heads, S);
#define LCC_ATTN_SH(GW)                                                                                                          \
  case GW: {                                                                                                                     \
    static DeviceOnce once;                                                                                                    \
    if (once.first()) { set_lds_attr(attn_shared_kernel<128, 1, 1, GW>, lds); set_lds_attr(attn_shared_kernel<128, 2, 1, GW>, lds); } \
    if (tile_rows == 32)                                                                                                         \
      attn_shared_kernel<128, 2, 1, GW><<<grid, dim3(GW * 64), lds, st>>>(q, nullptr, out, tile_stream, tile_q0, tile_nq, tile_pos0, nullptr, \
                                                                          nullptr, nullptr, kv_base, lay, layer, n_q_heads, 0, scale_l2e(128), S, ws_o, ws_ml, 0, 0); \
    else                                                                                                                         \
      attn_shared_kernel<128, 1, 1, GW><<<grid, dim3(GW * 64), lds, st>>>(q, nullptr, out, tile_stream, tile_q0, tile_nq, tile_pos0, nullptr, \
                                                                          nullptr, nullptr, kv_base, lay, layer, n_q_heads, 0, scale_l2e(128), S, ws_o, ws_ml, 0, 0); \
  } break;
    switch (G) {
      LCC_ATTN_SH(2) LCC_ATTN_SH(3) LCC_ATTN_SH(4) LCC_ATTN_SH(5) LCC_ATTN_SH(6) LCC_ATTN_SH(7) LCC_ATTN_SH(8)
    }
#undef LCC_ATTN_SH
    if (S > 1) attn_prefill_combine_kernel<<<dim3(n_q_heads, n_rows), dim3(128), 0, st>>>(ws_o, ws_ml, out, n_q_heads, S);
    return 0;
  }
  g_launch_counts[LC_ATTN_PREFILL_PER_WAVE]++;
  if (tile_rows == 32)
    attn_prefill_kernel<2><<<dim3((n_tiles + 3) / 4, n_q_heads, S), dim3(256), 0, st>>>(
        q, out, tile_stream, tile_q0, tile_nq, tile_pos0, kv_base, lay, layer, n_tiles, n_q_heads, scale_l2e(128), S, ws_o, ws_ml);
  else  // few query rows (a streaming chunk against a long cache): 16-row tiles double the number of waves
    attn_prefill_kernel<1><<<dim3((n_tiles + 3) / 4, n_q_heads, S), dim3(256), 0, st>>>(
        q, out, tile_stream, tile_q0, tile_nq, tile_pos0, kv_base, lay, layer, n_tiles, n_q_heads, scale_l2e(128), S, ws_o, ws_ml);
  if (S > 1) attn_prefill_combine_kernel<<<dim3(n_q_heads, n_rows), dim3(128), 0, st>>>(ws_o, ws_ml, out, n_q_heads, S);
  return 0;
}

int attn_decode_bf16(const bf16_t* q, bf16_t* out, const int32_t* slots, const int32_t* kv_len, bf16_t* const* kv_base,
                     KvLayout lay, int layer, int B, int n_q_heads, int nsplit, float* ws_o, float* ws_ml, hipStream_t st,
                     const AttnDirect* direct) {
  if (B <= 0) return 0;
  if (lay.head_dim != 128 || (lay.lmax & 31) || n_q_heads / lay.n_kv_heads > 16) return LCC_ERR_SHAPE;
  g_launch_counts[LC_ATTN_DECODE]++; g_launch_counts[LC_ATTN_DECODE_COMBINE]++; g_launch_counts[LC_LAST_DECODE_NSPLIT] = nsplit;
  if (direct != nullptr && direct->used == B && B <= 4)
    attn_decode_kernel<true><<<dim3(nsplit, lay.n_kv_heads, B), dim3(64), 0, st>>>(
        q, slots, kv_len, kv_base, lay, layer, n_q_heads, nsplit, ws_o, ws_ml, scale_l2e(128), *direct);
  else
    attn_decode_kernel<false><<<dim3(nsplit, lay.n_kv_heads, B), dim3(64), 0, st>>>(
        q, slots, kv_len, kv_base, lay, layer, n_q_heads, nsplit, ws_o, ws_ml, scale_l2e(128), AttnDirect{});
  attn_decode_combine_kernel<<<dim3(n_q_heads, B), dim3(256), 0, st>>>(ws_o, ws_ml, out, n_q_heads,
                                                                       lay.n_kv_heads, nsplit);
  return 0;
}

// 0: the last-arriving block of each (stream, KV head) merges the key splits in the same launch (ticket + acquire);
// 1: plain partials + attn_decode_combine_kernel as a second launch
static int g_attn_fused_tail = 1;
void set_attn_fused_tail(int v) { g_attn_fused_tail = v; }

int attn_decode_fused_bf16(const float* qkv_part, int ns_qkv, const bf16_t* bias, const bf16_t* cs, const bf16_t* sn,
                           const int32_t* slots, const int32_t* kv_len, bf16_t* const* kv_base, KvLayout lay, int layer, int B,
                           int n_q_heads, int nsplit, float* ws_o, float* ws_ml, int32_t* counters, bf16_t* out, hipStream_t st) {
  if (B <= 0) return 0;
  if (lay.head_dim != 128 || (lay.lmax & 31) || n_q_heads % lay.n_kv_heads || n_q_heads / lay.n_kv_heads > 16) return LCC_ERR_SHAPE;
  if (nsplit < 1 || nsplit > 64 || ns_qkv < 1 || ns_qkv > 8) return LCC_ERR_ARG;
  // LCC_ATTN_FUSED_WAVES=8: 512-thread blocks (two waves per SIMD, half the key tiles per wave); only with the separate combine launch
  static const int waves = [] { const char* v = getenv("LCC_ATTN_FUSED_WAVES"); return (v && atoi(v) == 8) ? 8 : 4; }();
  const int tail = g_attn_fused_tail;
  const int nw = (waves == 8 && tail != 0) ? 8 : 4;
  const dim3 grid(nsplit, lay.n_kv_heads, B), blk(nw * 64);
  const size_t lds = (size_t)nw * 16 * (128 + 4) * sizeof(float);
  g_launch_counts[tail == 0 ? LC_ATTN_DECODE_FUSED_MERGE : LC_ATTN_DECODE_FUSED]++; g_launch_counts[LC_LAST_DECODE_NSPLIT] = nsplit;
  if (tail != 0 && nsplit > 1) g_launch_counts[LC_ATTN_DECODE_COMBINE]++;
#define LCC_ADF(NS)                                                                                                              \
  case NS:                                                                                                                       \
    if (tail == 0)                                                                                                               \
      attn_decode_fused_kernel<NS, 0><<<grid, blk, lds, st>>>(qkv_part, bias, cs, sn, slots, kv_len, kv_base, lay, layer, B,     \
                                                              n_q_heads, nsplit, ws_o, ws_ml, counters, out, scale_l2e(128));    \
    else if (nw == 8) {                                                                                                          \
      static DeviceOnce attr;                                                                                                  \
      if (attr.first()) { set_lds_attr(attn_decode_fused_kernel<NS, 1, 8>, lds); }                                         \
      attn_decode_fused_kernel<NS, 1, 8><<<grid, blk, lds, st>>>(qkv_part, bias, cs, sn, slots, kv_len, kv_base, lay, layer, B,  \
                                                                 n_q_heads, nsplit, ws_o, ws_ml, counters, out, scale_l2e(128)); \
    } else                                                                                                                       \
      attn_decode_fused_kernel<NS, 1><<<grid, blk, lds, st>>>(qkv_part, bias, cs, sn, slots, kv_len, kv_base, lay, layer, B,     \
                                                              n_q_heads, nsplit, ws_o, ws_ml, counters, out, scale_l2e(128));    \
    break;
  switch (ns_qkv) {
    LCC_ADF(1) LCC_ADF(2) LCC_ADF(3) LCC_ADF(4) LCC_ADF(5) LCC_ADF(6) LCC_ADF(7) LCC_ADF(8)
  }
#undef LCC_ADF
  if (tail != 0 && nsplit > 1)
    attn_decode_combine_kernel<<<dim3(n_q_heads, B), dim3(256), 0, st>>>(ws_o, ws_ml, out, n_q_heads, lay.n_kv_heads, nsplit);
  return 0;
}

}  // namespace lcc
