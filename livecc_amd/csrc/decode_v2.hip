// Decode layer pipeline v2 (M <= 16 rows = one new token per stream): the 9-launch decoder layer of round 1
//
//     qkv GEMV -> rope_kv_append -> attn_decode -> combine -> o GEMV -> add_rmsnorm -> gate/up GEMV -> down GEMV -> add_rmsnorm
//
// becomes 6 launches by moving every elementwise stage into the GEMV that produces or consumes its data:
//
//     [RMSNorm] qkv GEMV [bias + M-RoPE + KV append] -> attn_decode -> combine -> o GEMV [residual add + sum of squares]
//       -> [RMSNorm] gate/up GEMV [SwiGLU] -> down GEMV [residual add + sum of squares]
//
// Measured on MI355X (rocprofv3 kernel trace of the round-1 path, LiveCC-7B, one stream): the three removed launches
// (rope_kv_append 4.7 us, add_rmsnorm 4.1 + 4.7 us) move < 100 KB each -- pure launch + dependent-round-trip latency, 13 % of a
// 104-us layer.  What makes the fusion free of cross-block reductions:
//   * o_proj and down_proj run WITHOUT an inter-block K split (8 waves per block split K inside the block and merge through
//     LDS), so a block owns 16 finished rows of the residual stream: it adds the residual in place (HF rounding: Linear output
//     -> bf16, residual + output -> bf16, Q2VL:594-612) and emits the sum of squares of its 16 new values.  The RMSNorm row
//     statistic of the NEXT GEMV is then 224 floats (H/16) per row instead of a re-read of split-K slabs.
//   * the consumer GEMV builds its activation fragments on the fly: x = bf16(w * bf16(h * rsqrt(mean(h^2) + eps))) (Q2VL:96-110)
//     from the h and norm-weight fragments it needs for its own K range -- the normalised row is never written to memory.  The
//     weight loads of the first pipeline stage are issued BEFORE that prologue, so its L2 round trips hide under the HBM latency.
//   * q/k/v: one block owns a 16-row tile that holds BOTH rotation partners of 8 channels (rows d..d+7 and d+64..d+71 of one
//     head: the decode copy of the qkv weight is row-permuted at load time, weights.py `qkv_w_dec`), so bias + M-RoPE (HF's bf16
//     op sequence, Q2VL:180-222) + the in-place KV append (cache_utils.py:127-146) run in the epilogue from registers.
// All of it is HBM-bound weight streaming (same packed fragment order, nontemporal 16-byte loads, 2-stage software pipeline as
// gemv_skinny_kernel); MFMA is only the multiply unit.  Algorithmic bytes per layer are unchanged (466 MB at 7B).
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace lcc {

__device__ unsigned int lcc_zero_page_v2[64];  // 256 zero bytes: activation operand of absent K chunks

enum { DG_PRO_PLAIN = 0, DG_PRO_NORM = 1 };
enum { DG_EPI_BF16 = 0, DG_EPI_SWIGLU = 1, DG_EPI_RESID = 2, DG_EPI_ROPE = 3 };

// ---- chained launches (round 3): two dependent GEMVs in ONE launch.  The consumer's blocks (q/k/v of layer l+1) are resident next to
// the producer's (down_proj of layer l), request their weights -- which do not depend on the producer -- right away, and only then
// wait for the producer's blocks to have published the residual stream: the HBM stream does not drain at the hand-off (the
// "prefetch-credit" of MI355X_MICROARCH.md) and one kernel boundary per layer disappears.  Hand-off = Guideline 16, recipe R1:
// the producer's epilogue stores are write-through (relaxed agent-scope 8-byte atomic stores = `global_store_dwordx2 sc0 sc1`), the
// storing wave drains them (vmcnt(0)), one lane adds 1 to a monotonic counter; the consumer polls that ONE word relaxed from one lane
// (s_sleep between polls), then reads the published bytes with agent-scope (sc1, L1-bypassing) loads.  Placement-independent; the
// spin is bounded (a block that gives up raises *err and the engine fails the call).  Deadlock-free by construction: the launch is
// only used when producer + consumer blocks fit the chip at once, and producer blocks never wait.
enum { DG_CHAIN_NONE = 0, DG_CHAIN_SIGNAL = 1, DG_CHAIN_WAIT = 2 };
struct DgChain { unsigned* flag; unsigned target; unsigned* err; int delay_us; };   // delay_us: no polling for about that long (the
// producer streams its weights for a known time: 288 lanes polling one word across the fabric from the first microsecond on cost the
// producer a third of its bandwidth -- 43.6 us for the chained launch against 25.5 + 10.7 us for the two launches)

LCC_DEVICE u32x4 ld16_agent(const void* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (u32x4){(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}
LCC_DEVICE f32x4 ld16f_agent(const void* p) { return __builtin_bit_cast(f32x4, ld16_agent(p)); }
LCC_DEVICE void st8_agent(void* p, u32x2 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// x fragment of 8 consecutive k for activation row m: PRO_NORM = bf16(w * bf16(h * r))
LCC_DEVICE u32x4 norm_frag(u32x4 hv, u32x4 wv, float r) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = lo2f(wv[e]) * rbf(lo2f(hv[e]) * r);
    const float b = hi2f(wv[e]) * rbf(hi2f(hv[e]) * r);
    o[e] = pack2(a, b);
  }
  return o;
}

// Occupancy decides whether the whole grid is resident at once (one round) or a few blocks start a second round that pays the
// prologue latency again behind an almost idle memory system (7B shapes): gate/up = 1184 blocks of 4 waves need 5 blocks per CU
// (<= 96 VGPRs; at 97 the kernel ran 1024 + 160); q/k/v = 288 blocks of 8 waves need 2 per CU (<= 128 VGPRs; at 140: 256 + 32).
// MR = activation rows (streams) whose RMSNorm statistic every wave reduces in the prologue: 2 (the engine's v2 batches) or 4.
template <int NTILE, int PRO, int EPI, int NW, int MR>
constexpr int dgemv_min_waves_per_simd() {
  return MR > 2 ? 1 : ((NTILE == 2 && PRO == DG_PRO_NORM && NW == 4) ? 5 : ((EPI == DG_EPI_ROPE && NW == 8) ? 4 : 1));
}

template <int NTILE, int PRO, int EPI, int NW, int UNR, int MR, int CHAIN, bool W8 = false>
LCC_DEVICE void dgemv_body(const DgArgs& a, const DgChain& ch, const int bid, f32x4 (*red)[NTILE][64], bf16_t* s_x) {
  static_assert(EPI != DG_EPI_SWIGLU || NTILE == 2, "swiglu needs the gate and the up tile in one block");
  static_assert(EPI == DG_EPI_SWIGLU || NTILE == 1, "one tile per block");
  static_assert(CHAIN != DG_CHAIN_SIGNAL || EPI == DG_EPI_RESID, "the producer of a chained launch publishes the residual stream");
  // PRO_NORM: the activation rows are built ONCE per block in LDS (M * K <= 16384 elements) and the MFMA fragments are
  // read from there.  (The first version normalised every fragment in registers: 16 lanes of a wave computed the same 8 values and
  // every one of the N/16 blocks repeated it -- +6.7 us on the gate/up GEMV, +24 us on lm_head.)
  constexpr bool XLDS = PRO != DG_PRO_PLAIN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int M = a.M, N = a.N, K = a.K;
  const int n0 = bid * (NTILE * 16);
  const int nchunk = (K + 63) >> 6, K32 = (K + 31) >> 5;
  const int xm = min(li, M - 1);
  const bf16_t* zp = reinterpret_cast<const bf16_t*>(lcc_zero_page_v2) + g * 8;
  // fp8 weights (W8): a 64-k chunk of a 16-row tile is ONE 1-KB fragment (lane (g, row) owns the 16 bytes k = c*64 + g*16 .. +15), fed
  // to two MFMAs (bytes 0-7, 8-15) whose activation fragments follow the same k assignment: x[c*64 + g*16 + h*8 .. +8]
  const bf16_t* xp = PRO == DG_PRO_PLAIN ? a.X + (size_t)xm * a.ldx + (W8 ? g * 16 : g * 8) : nullptr;
  const bf16_t* wp[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
    wp[t] = W8 ? reinterpret_cast<const bf16_t*>(reinterpret_cast<const uint8_t*>(a.W) + (size_t)((n0 >> 4) + t) * nchunk * 1024 + lane * 16)
               : a.W + (size_t)((n0 >> 4) + t) * K32 * 512 + lane * 8;

  f32x4 acc[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  struct Stage { u32x4 w[UNR][W8 ? 1 : 2][NTILE]; u32x4 x[XLDS ? 1 : UNR][2]; };
  auto load_w = [&](int c0, Stage& s) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int cc = c0 + u * NW;
      if (W8) {
        const int ccl = min(cc, nchunk - 1);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) s.w[u][0][t] = __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)ccl * 512));   // 1 KB per chunk
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kbc = min(2 * cc + h, K32 - 1);
#pragma unroll
          for (int t = 0; t < NTILE; ++t) s.w[u][W8 ? 0 : h][t] = __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)kbc * 512));
        }
      }
    }
  };
  auto load_x = [&](int c0, Stage& s) {
    if (XLDS) return;                      // activations come from LDS
#pragma unroll
    for (int u = 0; u < (XLDS ? 1 : UNR); ++u) {
      const int cc = c0 + u * NW;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kb = 2 * cc + h;
        const bool ok = cc < nchunk && (W8 || kb < K32);      // wave-uniform: an absent block multiplies by a zero activation
        const int kbc = min(kb, K32 - 1);
        s.x[u][h] = ld16(ok ? (W8 ? xp + min(cc, nchunk - 1) * 64 + h * 8 : xp + kbc * 32) : zp);
      }
    }
  };
  auto mma = [&](const Stage& s, int c0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 xf;
        if (XLDS) {
          const int cc = c0 + u * NW, kb = 2 * cc + h;
          const int xo = W8 ? cc * 64 + g * 16 + h * 8 : kb * 32 + g * 8;
          xf = (cc < nchunk && (W8 || kb < K32)) ? *reinterpret_cast<const u32x4*>(s_x + (size_t)xm * K + xo) : (u32x4){0u, 0u, 0u, 0u};
        } else {
          xf = s.x[XLDS ? 0 : u][h];
        }
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
          acc[t] = mfma16(W8 ? fp8x8_to_bf16x8(s.w[u][0][t][2 * h], s.w[u][0][t][2 * h + 1]) : as_bf16x8(s.w[u][W8 ? 0 : h][t]), as_bf16x8(xf), acc[t]);
      }
  };

  constexpr int STEP = UNR * NW;
  Stage sa, sb;
  int c = wave;

  // ---- everything the kernel will need from the previous launch is requested up front, in ONE burst with the first two weight
  //      stages: vmcnt retires loads in issue order, so a small load issued after the weight stream would only return behind it
  //      (a 14-iteration dependent loop over the row statistics at this point cost 8 us per block in the first version) ----
  // PRO_NORM: every wave reads the n_stat per-tile sums of squares of each row itself (n_stat / 4 <= 128 16-byte pieces: one or two
  // per lane, summed by a butterfly -- the same fixed order in every wave of every block), and the block's threads share the
  // 8-element pieces of the residual rows and of the norm weight that are normalised into LDS
  constexpr int T = NW * 64, XP = 2;      // XP row pieces per thread are requested up front (all of them for one stream)
  if (CHAIN == DG_CHAIN_WAIT) {
    // consumer of a chained launch: the weights first (they do not depend on the producer), then wait for the producer's blocks
    __builtin_amdgcn_sched_barrier(0);
    load_w(c, sa);
    load_w(c + STEP, sb);
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x == 0) {
      for (int i = 0; i < ch.delay_us; ++i) __builtin_amdgcn_s_sleep(36);      // ~1 us each (64 x 36 cycles): memory-silent wait
      int spins = 0;
      while ((int)(__hip_atomic_load(ch.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ch.target) < 0) {
        if (++spins > 200000) { __hip_atomic_fetch_or(ch.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(36);
      }
    }
    __syncthreads();
  }
  const int n4 = a.n_stat >> 2;
  const int kp = K >> 3, xpieces = PRO == DG_PRO_NORM ? M * kp : 0;
  f32x4 pv[PRO == DG_PRO_NORM ? MR : 1][2];
  u32x4 hv[PRO == DG_PRO_NORM ? XP : 1], nv[PRO == DG_PRO_NORM ? XP : 1];
  if (PRO == DG_PRO_NORM) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = lane + 64 * u;           // unconditional (clamped) loads: a branch here would serialise them
        const float* sp = a.stats + (size_t)min(m, M - 1) * a.n_stat + 4 * (i < n4 ? i : 0);
        pv[m][u] = CHAIN == DG_CHAIN_WAIT ? ld16f_agent(sp) : *reinterpret_cast<const f32x4*>(sp);
      }
#pragma unroll
    for (int u = 0; u < XP; ++u) {
      const int p = (int)threadIdx.x + u * T, pc = p < xpieces ? p : 0;
      hv[u] = CHAIN == DG_CHAIN_WAIT ? ld16_agent(a.H + (size_t)pc * 8) : ld16(a.H + (size_t)pc * 8);   // rows are contiguous: [M][K]
      nv[u] = ld16(a.norm_w + (size_t)(pc % kp) * 8);
    }
  }
  // epilogue operands of wave 0 (residual row / bias, cos, sin and the KV slot): in flight during the whole weight stream
  u32x2 e_h = {0u, 0u}, e_b = {0u, 0u}, e_c = {0u, 0u}, e_s = {0u, 0u};
  int e_pos = 0;
  bf16_t* e_base = nullptr;
  if (wave == 0) {
    if (EPI == DG_EPI_RESID) {
      const int n = n0 + g * 4;
      if (li < M && n < N) e_h = ld8(a.Hres + (size_t)li * N + n);
    } else if (EPI == DG_EPI_ROPE) {
      const int tile = n0 >> 4, head = tile >> 3, j = tile & 7;
      const int dc = j * 8 + (g & 1) * 4, d = (g >> 1) * 64 + dc;
      e_b = ld8(a.bias + head * 128 + d);
      e_c = ld8(a.cs + (size_t)xm * 64 + dc);
      e_s = ld8(a.sn + (size_t)xm * 64 + dc);
      const int strm = a.tok_stream[xm];
      e_pos = a.kv_len[strm];
      e_base = a.kv_base[strm];
    }
  }
  // keep the small loads AHEAD of the weight stream in the memory queue (vmcnt retires in issue order): the compiler otherwise
  // sinks some of them behind the weight loads and then has to wait for the whole first stage before the prologue can run
  __builtin_amdgcn_sched_barrier(0);
  if (CHAIN != DG_CHAIN_WAIT) {
    load_w(c, sa);
    load_w(c + STEP, sb);
  }
  load_x(c, sa);
  load_x(c + STEP, sb);
  __builtin_amdgcn_sched_barrier(0);
  if (PRO == DG_PRO_NORM) {
    float rr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float sacc = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (lane + 64 * u < n4) sacc += (pv[m][u][0] + pv[m][u][1]) + (pv[m][u][2] + pv[m][u][3]);
      rr[m] = rsqrtf(wave_sum(sacc) / (float)K + a.eps);
    }
    auto row_r = [&](int m) { return m == 0 ? rr[0] : (m == 1 ? rr[1] : (m == 2 ? rr[2] : rr[3])); };
#pragma unroll
    for (int u = 0; u < XP; ++u) {
      const int p = (int)threadIdx.x + u * T;
      if (p < xpieces) *reinterpret_cast<u32x4*>(s_x + (size_t)p * 8) = norm_frag(hv[u], nv[u], row_r(p / kp));
    }
    for (int p = (int)threadIdx.x + XP * T; p < xpieces; p += T)       // batches of several streams: the remaining pieces
      *reinterpret_cast<u32x4*>(s_x + (size_t)p * 8) = norm_frag(CHAIN == DG_CHAIN_WAIT ? ld16_agent(a.H + (size_t)p * 8) : ld16(a.H + (size_t)p * 8),
                                                                 ld16(a.norm_w + (size_t)(p % kp) * 8), row_r(p / kp));
    __syncthreads();
  }
  for (; c < nchunk; c += 2 * STEP) {
    mma(sa, c);
    if (c + 2 * STEP < nchunk) { load_w(c + 2 * STEP, sa); load_x(c + 2 * STEP, sa); }   // wave-uniform: no loads past the last chunk
    if (c + STEP < nchunk) mma(sb, c + STEP);
    if (c + 3 * STEP < nchunk) { load_w(c + 3 * STEP, sb); load_x(c + 3 * STEP, sb); }
  }

  // cross-wave reduction through LDS; wave 0 runs the epilogue.  D'[n][m]: lane (m = li, g) holds 4 consecutive n = g*4 ..
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NTILE; ++t) red[wave - 1][t][lane] = acc[t];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) acc[t] += red[w][t][lane];
    if (W8) acc[t] *= *reinterpret_cast<const f32x4*>(a.wscale + min(n0 + t * 16 + g * 4, N - 4));   // per-stored-row dequantisation scale
  }

  if (EPI == DG_EPI_BF16) {
    if (li >= M) return;
    const int n = n0 + g * 4;
    if (n >= N) return;
    float v[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
    if (a.bias != nullptr) {
      const u32x2 b = ld8(a.bias + n);
      v[0] += lo2f(b.x); v[1] += hi2f(b.x); v[2] += lo2f(b.y); v[3] += hi2f(b.y);
    }
    st8(a.C + (size_t)li * a.ldc + n, (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])});
  } else if (EPI == DG_EPI_SWIGLU) {   // tile 0 = 16 gate rows, tile 1 = the 16 matching up rows (weights.py interleave_gate_up)
    if (li >= M || n0 >= N) return;
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = silu_bf16(rbf(acc[0][q])) * rbf(acc[NTILE - 1][q]);
    st8(a.C + (size_t)li * a.ldc + n0 / 2 + g * 4, (u32x2){pack2(o[0], o[1]), pack2(o[2], o[3])});
  } else if (EPI == DG_EPI_RESID) {
    // h = bf16(h + bf16(Linear output)); stats_out[m][tile] = sum of squares of the 16 new values of this tile
    const int n = n0 + g * 4;
    float ss = 0.f;
    if (li < M && n < N) {
      bf16_t* hp = a.Hres + (size_t)li * N + n;
      const u32x2 hv = e_h;
      const float v0 = rbf(lo2f(hv.x) + rbf(acc[0][0])), v1 = rbf(hi2f(hv.x) + rbf(acc[0][1]));
      const float v2 = rbf(lo2f(hv.y) + rbf(acc[0][2])), v3 = rbf(hi2f(hv.y) + rbf(acc[0][3]));
      if (CHAIN == DG_CHAIN_SIGNAL) st8_agent(hp, (u32x2){pack2(v0, v1), pack2(v2, v3)});
      else st8(hp, (u32x2){pack2(v0, v1), pack2(v2, v3)});
      ss = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (g == 0 && li < M) {
      float* sp = a.stats_out + (size_t)li * (N >> 4) + (n0 >> 4);
      if (CHAIN == DG_CHAIN_SIGNAL) __hip_atomic_store(reinterpret_cast<unsigned*>(sp), __float_as_uint(ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *sp = ss;
    }
    if (CHAIN == DG_CHAIN_SIGNAL) {       // wave 0 is the only storing wave: drain its write-through stores, then ONE lane publishes
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(ch.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {   // DG_EPI_ROPE: tile rows 0-7 = channels d0..d0+7 of a head, rows 8-15 = their rotation partners d0+64..
    constexpr int D = 128;
    const int tile = n0 >> 4, head = tile >> 3, j = tile & 7;
    const int hkv = a.lay.n_kv_heads, nq = a.n_q_heads;
    const int half = g >> 1;                          // 0: first half of the head (d < 64), 1: second half
    const int dc = j * 8 + (g & 1) * 4;               // channel inside the half, 4 consecutive
    const int d = half * 64 + dc;
    float x[4], xo[4];
    {
      const u32x2 b = e_b;                           // HF: Linear output = bf16(acc + bias)
      x[0] = rbf(acc[0][0] + lo2f(b.x)); x[1] = rbf(acc[0][1] + hi2f(b.x));
      x[2] = rbf(acc[0][2] + lo2f(b.y)); x[3] = rbf(acc[0][3] + hi2f(b.y));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xo[q] = __shfl_xor(x[q], 32, 64);      // partner channel (d +- 64) of the same token row
    if (li >= M) return;
    const int pos = e_pos;                           // the new token's cache index
    bf16_t* base = e_base + (size_t)a.layer * a.lay.layer_stride();
    if (head < nq + hkv) {
      const u32x2 cq = e_c, sq = e_s;
      const float cv[4] = {lo2f(cq.x), hi2f(cq.x), lo2f(cq.y), hi2f(cq.y)};
      const float sv[4] = {lo2f(sq.x), hi2f(sq.x), lo2f(sq.y), hi2f(sq.y)};
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)   // out = bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)); rotate_half = (-x2, x1)
        o[q] = half == 0 ? rbf(x[q] * cv[q]) + rbf(-xo[q] * sv[q]) : rbf(x[q] * cv[q]) + rbf(xo[q] * sv[q]);
      bf16_t* dst = head < nq ? a.q_out + ((size_t)li * nq + head) * D
                              : base + (size_t)(head - nq) * a.lay.head_stride() + (size_t)pos * D;
      st8(dst + d, (u32x2){pack2(o[0], o[1]), pack2(o[2], o[3])});
    } else {     // V: no rotation; cache is blocked-transposed [Lmax/32][128][32]
      const int hv = head - nq - hkv;
      bf16_t* dst = base + a.lay.kv_stride() + (size_t)hv * a.lay.head_stride() + ((size_t)(pos >> 5) * D + d) * 32 + (pos & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q * 32] = f2bf(x[q]);
    }
  }
}

template <int NTILE, int PRO, int EPI, int NW, int UNR, int MR = 4, bool W8 = false>
__global__ __launch_bounds__(NW * 64, (dgemv_min_waves_per_simd<NTILE, PRO, EPI, NW, MR>())) void dgemv_kernel(DgArgs a) {
  __shared__ f32x4 red[NW - 1][NTILE][64];
  extern __shared__ __attribute__((aligned(16))) bf16_t s_x[];   // PRO_NORM: M * K bf16 (dynamic: 7 KB for one stream at 7B)
  dgemv_body<NTILE, PRO, EPI, NW, UNR, MR, DG_CHAIN_NONE, W8>(a, DgChain{nullptr, 0u, nullptr, 0}, blockIdx.x, red, s_x);
}

// down_proj of layer l (blocks [0, nb_down): producer, publishes the residual stream + its tile statistics) and the q/k/v GEMV of
// layer l+1 (the remaining blocks: consumer) in ONE launch.  512-thread blocks, <= 128 VGPRs: two blocks per CU, so all
// nb_down + nb_qkv blocks of LiveCC-7B (224 + 288 = 512 = 2 x 256 CUs) are resident at once (the host checks the capacity).
// The producer part runs UNR = 3 (120 VGPRs; the stand-alone kernel's UNR = 4 needs 152).
template <int MR>
__global__ __launch_bounds__(512, 4) void dgemv_down_qkv_kernel(DgArgs down, DgArgs qkv, DgChain ch, int nb_down) {
  __shared__ f32x4 red[7][1][64];
  extern __shared__ __attribute__((aligned(16))) bf16_t s_x[];
  if ((int)blockIdx.x < nb_down) dgemv_body<1, DG_PRO_PLAIN, DG_EPI_RESID, 8, 3, 4, DG_CHAIN_SIGNAL>(down, ch, blockIdx.x, red, s_x);
  else dgemv_body<1, DG_PRO_NORM, DG_EPI_ROPE, 8, 4, MR, DG_CHAIN_WAIT>(qkv, ch, (int)blockIdx.x - nb_down, red, s_x);
}

// ------------------------------------------------------------------------------------------------------------------------
// step prologue: one block per stream does what were four launches (seen_set, embed_gather, mrope_table_decode, and the
// first layer's RMSNorm statistic): marks the consumed token in the stream's seen-id bitmap (repetition penalty), gathers its
// embedding row into the residual stream with the per-tile sums of squares, and builds the cos/sin row of its position.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_step_begin_kernel(
    const int32_t* __restrict__ slots, const int32_t* __restrict__ cur_tok, const int32_t* __restrict__ done, uint32_t* __restrict__ seen,
    int words, const bf16_t* __restrict__ table, bf16_t* __restrict__ h, float* __restrict__ stats, int dim,
    const int32_t* __restrict__ pos, const float* __restrict__ inv_freq, bf16_t* __restrict__ cs, bf16_t* __restrict__ sn) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int slot = slots[b];
  const int id = cur_tok[slot];
  if (t == 0 && !(done != nullptr && done[slot])) atomicOr(seen + (size_t)slot * words + (id >> 5), 1u << (id & 31));
  if (t < 64) {
    const float ang = inv_freq[t] * (float)pos[slot];
    cs[b * 64 + t] = f2bf(cosf(ang));
    sn[b * 64 + t] = f2bf(sinf(ang));
  }
  const bf16_t* src = table + (size_t)id * dim;
  for (int c = t; c * 8 < dim; c += 256) {      // 8 channels per thread; a 16-channel tile = 2 neighbouring threads
    const u32x4 q = ld16(src + c * 8);
    st16(h + (size_t)b * dim + c * 8, q);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float lo = lo2f(q[e]), hi = hi2f(q[e]); s += lo * lo + hi * hi; }
    s += __shfl_xor(s, 1, 64);
    if ((c & 1) == 0) stats[(size_t)b * (dim >> 4) + (c >> 1)] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------
int decode_step_begin(const int32_t* slots, const int32_t* cur_tok, const int32_t* done, uint32_t* seen, int words, const bf16_t* table,
                      bf16_t* h, float* stats, int dim, const int32_t* pos, const float* inv_freq, bf16_t* cs, bf16_t* sn, int B,
                      hipStream_t st) {
  if (B <= 0) return 0;
  if (dim & 15) return LCC_ERR_SHAPE;
  decode_step_begin_kernel<<<dim3(B), dim3(256), 0, st>>>(slots, cur_tok, done, seen, words, table, h, stats, dim, pos, inv_freq, cs, sn);
  return 0;
}

static int dg_check(const DgArgs& a, int pro, int epi) {
  g_launch_counts[LC_DGEMV_V2]++;
  if (a.M < 1 || a.M > 16 || (a.N & 15) || (a.K & 31) || a.W == nullptr) return LCC_ERR_SHAPE;
  if (a.wscale != nullptr && (a.K & 63)) return LCC_ERR_SHAPE;       // fp8 weights: whole 64-k fragments
  if (pro == DG_PRO_PLAIN && (a.X == nullptr || (a.ldx & 7))) return LCC_ERR_ARG;
  if (pro == DG_PRO_NORM && (a.H == nullptr || a.stats == nullptr || a.norm_w == nullptr || a.n_stat != (a.K >> 4) || (a.n_stat & 3) ||
                             a.M > 4 || a.M * a.K > 16384 || a.n_stat > 512)) return LCC_ERR_ARG;
  if (epi == DG_EPI_RESID && (a.Hres == nullptr || a.stats_out == nullptr)) return LCC_ERR_ARG;
  if ((epi == DG_EPI_BF16 || epi == DG_EPI_SWIGLU) && a.C == nullptr) return LCC_ERR_ARG;
  if (epi == DG_EPI_SWIGLU && (a.N & 31)) return LCC_ERR_SHAPE;
  if (epi == DG_EPI_ROPE && (a.bias == nullptr || a.cs == nullptr || a.sn == nullptr || a.tok_stream == nullptr || a.kv_len == nullptr ||
                             a.kv_base == nullptr || a.q_out == nullptr || a.lay.head_dim != 128 || (a.lay.lmax & 31) ||
                             a.N != (a.n_q_heads + 2 * a.lay.n_kv_heads) * 128)) return LCC_ERR_ARG;
  return 0;
}

// [RMSNorm] q/k/v Linear [bias + M-RoPE + KV append]: W = the row-permuted decode copy of the fused q|k|v weight
int dgemv_qkv_rope(const DgArgs& a, hipStream_t st) {
  if (int rc = dg_check(a, DG_PRO_NORM, DG_EPI_ROPE)) return rc;
  if (a.wscale != nullptr) {     // fp8 weights (half the bytes per chunk: UNR 4 is 4 KB in flight per wave and stage)
    if (a.M <= 2) dgemv_kernel<1, DG_PRO_NORM, DG_EPI_ROPE, 8, 4, 2, true><<<dim3(a.N / 16), dim3(512), (size_t)a.M * a.K * 2, st>>>(a);
    else dgemv_kernel<1, DG_PRO_NORM, DG_EPI_ROPE, 8, 4, 4, true><<<dim3(a.N / 16), dim3(512), (size_t)a.M * a.K * 2, st>>>(a);
    return 0;
  }
  if (a.M <= 2) dgemv_kernel<1, DG_PRO_NORM, DG_EPI_ROPE, 8, 4, 2><<<dim3(a.N / 16), dim3(512), (size_t)a.M * a.K * 2, st>>>(a);
  else dgemv_kernel<1, DG_PRO_NORM, DG_EPI_ROPE, 8, 4, 4><<<dim3(a.N / 16), dim3(512), (size_t)a.M * a.K * 2, st>>>(a);
  return 0;
}
static int g_resid_waves = [] { const char* v = getenv("LCC_RESID_WAVES16"); return v ? atoi(v) : 1; }();
int set_resid_waves(int mode) { const int old = g_resid_waves; g_resid_waves = mode < 0 ? 0 : (mode > 2 ? 2 : mode); return old; }
// o_proj / down_proj: x plain, residual add in place + per-tile sums of squares
int dgemv_resid(const DgArgs& a, hipStream_t st) {
  if (int rc = dg_check(a, DG_PRO_PLAIN, DG_EPI_RESID)) return rc;
  // A block owns 16 rows x the whole K, so its waves' dependent chain of load stages is K / (64 NW UNR) round trips long: with 8 waves
  // the 7B down_proj (K = 18944) costs 12.7-14.6 us before / after its stream (tools/bench_dgemv_intercept.py: t = fixed + bytes / rate).
  // 1024-thread blocks (16 waves, UNR 3: 120 VGPRs at 4 waves per SIMD) halve that chain: 27.7 -> 25.1 us stand-alone, decode step
  // 2977 -> 2935 us, 270.5 -> 274.1 tokens/s (profiles/r03/dgemv_resid_16waves.txt); o_proj (K = 3584: 7 chunks per wave already) does
  // not gain (2971 us), and neither do fp8 weights (7B: 2036 vs 2017 us per step, 72B: 13.73 vs 13.49 ms -- half the bytes per chunk).
  // Mode (lcc_debug_set_resid_waves; initial value from LCC_RESID_WAVES16): 0 = 8 waves always, 1 (default) = 16 waves for bf16 weights
  // with K >= 8192, 2 = 16 waves for every call.
  const int w16 = g_resid_waves;
  const bool wide = w16 == 2 || (w16 == 1 && a.K >= 8192 && a.wscale == nullptr);
  if (a.wscale != nullptr) {
    if (wide) dgemv_kernel<1, DG_PRO_PLAIN, DG_EPI_RESID, 16, 4, 4, true><<<dim3(a.N / 16), dim3(1024), 0, st>>>(a);
    else dgemv_kernel<1, DG_PRO_PLAIN, DG_EPI_RESID, 8, 4, 4, true><<<dim3(a.N / 16), dim3(512), 0, st>>>(a);
    return 0;
  }
  if (wide) dgemv_kernel<1, DG_PRO_PLAIN, DG_EPI_RESID, 16, 3><<<dim3(a.N / 16), dim3(1024), 0, st>>>(a);
  else dgemv_kernel<1, DG_PRO_PLAIN, DG_EPI_RESID, 8, 4><<<dim3(a.N / 16), dim3(512), 0, st>>>(a);
  return 0;
}
// chained launch: down_proj of one layer + q/k/v of the next (see dgemv_down_qkv_kernel).  `blocks_capacity` = co-resident 512-thread
// blocks of that kernel on this device (dgemv_chain_capacity); LCC_ERR_STATE when the grid does not fit (the caller then launches the
// two GEMVs separately).  *flag is a monotonic counter: the consumer waits for it to reach `target`.
int dgemv_chain_capacity() {
  static int cap = -1;
  if (cap < 0) {
    int per_cu = 0, cus = 0, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dgemv_down_qkv_kernel<2>, 512, 16 * 1024) != hipSuccess) per_cu = 0;
    cap = std::max(0, std::min(per_cu, 2)) * cus;
  }
  return cap;
}
int dgemv_down_qkv(const DgArgs& down, const DgArgs& qkv, unsigned* flag, unsigned target, unsigned* err, hipStream_t st) {
  if (int rc = dg_check(down, DG_PRO_PLAIN, DG_EPI_RESID)) return rc;
  if (int rc = dg_check(qkv, DG_PRO_NORM, DG_EPI_ROPE)) return rc;
  if (flag == nullptr || err == nullptr) return LCC_ERR_ARG;
  const int nb_down = down.N / 16, nb_qkv = qkv.N / 16;
  // M <= 2 only: the 3-4 row variant of the q/k/v part needs 140 VGPRs and would spill under the two-blocks-per-CU register budget
  if (qkv.M > 2 || down.M != qkv.M || nb_down + nb_qkv > dgemv_chain_capacity() || (size_t)qkv.M * qkv.K * 2 > 16 * 1024) return LCC_ERR_STATE;
  // silent wait before the first poll: ~80 % of the producer's streaming time at 5.3 TB/s (LCC_CHAIN_DELAY_US overrides)
  static const int forced = [] { const char* v = getenv("LCC_CHAIN_DELAY_US"); return v ? atoi(v) : -1; }();
  const double prod_us = (double)down.N * down.K * 2.0 / 5.3e6;
  const DgChain ch{flag, target, err, forced >= 0 ? forced : (int)(0.8 * prod_us)};
  dgemv_down_qkv_kernel<2><<<dim3(nb_down + nb_qkv), dim3(512), (size_t)qkv.M * qkv.K * 2, st>>>(down, qkv, ch, nb_down);
  return 0;
}
static thread_local hipEvent_t g_swiglu_ev0 = nullptr, g_swiglu_ev1 = nullptr;
void dgemv_attach_events_to_next_swiglu(hipEvent_t start, hipEvent_t stop) { g_swiglu_ev0 = start; g_swiglu_ev1 = stop; }
// [RMSNorm] gate/up Linear [SwiGLU]
int dgemv_norm_swiglu(const DgArgs& a, hipStream_t st) {
  if (int rc = dg_check(a, DG_PRO_NORM, DG_EPI_SWIGLU)) return rc;
  hipEvent_t ev0 = g_swiglu_ev0, ev1 = g_swiglu_ev1;
  g_swiglu_ev0 = g_swiglu_ev1 = nullptr;
  if (ev0 != nullptr && ev1 != nullptr && a.wscale == nullptr && a.M <= 2) {      // the profiled launch of the bf16 one-/two-stream kernel
    hipExtLaunchKernelGGL((dgemv_kernel<2, DG_PRO_NORM, DG_EPI_SWIGLU, 4, 1, 2>), dim3(a.N / 32), dim3(256), (uint32_t)((size_t)a.M * a.K * 2), st, ev0, ev1, 0u, a);
    return 0;
  }
  if (ev0 != nullptr && ev1 != nullptr) {      // other instantiations: bracket with plain records (the events must be recorded either way)
    (void)hipEventRecord(ev0, st);
    const int rc = dgemv_norm_swiglu(a, st);
    (void)hipEventRecord(ev1, st);
    return rc;
  }
  if (a.wscale != nullptr) {     // fp8: two chunks per stage keep the bytes in flight of the bf16 kernel
    if (a.M <= 2) dgemv_kernel<2, DG_PRO_NORM, DG_EPI_SWIGLU, 4, 2, 2, true><<<dim3(a.N / 32), dim3(256), (size_t)a.M * a.K * 2, st>>>(a);
    else dgemv_kernel<2, DG_PRO_NORM, DG_EPI_SWIGLU, 4, 2, 4, true><<<dim3(a.N / 32), dim3(256), (size_t)a.M * a.K * 2, st>>>(a);
    return 0;
  }
  if (a.M <= 2) dgemv_kernel<2, DG_PRO_NORM, DG_EPI_SWIGLU, 4, 1, 2><<<dim3(a.N / 32), dim3(256), (size_t)a.M * a.K * 2, st>>>(a);
  else dgemv_kernel<2, DG_PRO_NORM, DG_EPI_SWIGLU, 4, 1, 4><<<dim3(a.N / 32), dim3(256), (size_t)a.M * a.K * 2, st>>>(a);
  return 0;
}
// [final RMSNorm] lm_head
int dgemv_norm_bf16(const DgArgs& a, hipStream_t st) {
  if (int rc = dg_check(a, DG_PRO_NORM, DG_EPI_BF16)) return rc;
  if (a.wscale != nullptr) {
    if (a.M <= 2) dgemv_kernel<1, DG_PRO_NORM, DG_EPI_BF16, 4, 2, 2, true><<<dim3(a.N / 16), dim3(256), (size_t)a.M * a.K * 2, st>>>(a);
    else dgemv_kernel<1, DG_PRO_NORM, DG_EPI_BF16, 4, 2, 4, true><<<dim3(a.N / 16), dim3(256), (size_t)a.M * a.K * 2, st>>>(a);
    return 0;
  }
  if (a.M <= 2) dgemv_kernel<1, DG_PRO_NORM, DG_EPI_BF16, 4, 1, 2><<<dim3(a.N / 16), dim3(256), (size_t)a.M * a.K * 2, st>>>(a);
  else dgemv_kernel<1, DG_PRO_NORM, DG_EPI_BF16, 4, 1, 4><<<dim3(a.N / 16), dim3(256), (size_t)a.M * a.K * 2, st>>>(a);
  return 0;
}

}  // namespace lcc
