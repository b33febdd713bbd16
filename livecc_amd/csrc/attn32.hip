// LLM prefill attention, second generation: 32-row query tiles on the 32x32x16 bf16 MFMA, K/V tiles shared by the G query heads of
// one KV head through an LDS-DMA ring.  Replaces HF modeling_qwen2_vl.py:537-556 (Qwen2VLAttention core: causal, bottom-right aligned,
// GQA) for prefill rows, same contract as attn_shared_kernel<128, NQ, 1, G> in attention.hip (which stays as variant 2).
//
// Why (profiles/r03/pmc_attn_summary.json, 8 streams x 386 rows x 6k keys on attn_shared_kernel<128,1,1,7>): 25 % MFMA busy, 8.3 VALU
// instructions per MFMA, 16 ds_read_b128 per 16 MFMAs of 16 cycles each -- per CU and 32-key tile the LDS reads (112 KB / 256 B/clk
// = 437 clk), the MFMAs (448 clk) and the softmax VALU work cost about the same and only partly overlap: 0.54 PF.  One wave now owns
// 32 query rows of one head:
//   * S^T[32 keys][32 queries] = K . Q^T  is 8 MFMAs of 32 cycles (k = 16 each), O^T[128 d][32 queries] += V^T . P^T another 8: the
//     same 16 ds_read_b128 per tile now feed 512 MFMA cycles instead of 256 -- LDS traffic per flop halves;
//   * in the 32x32 C/D layout a lane owns ONE query column (l & 31) and 16 keys, so the row maximum / sum are 15 in-lane steps + ONE
//     cross-lane step (xor 32) instead of 7 + 2 per 8 scores;
//   * the K rows of a tile are fetched in a permuted order (bits 2 and 3 of the row index swapped), so that the 16 scores a lane
//     ends up with are exactly the 2 x 8 consecutive keys its B-operand slots of the two P.V MFMAs need: P goes from the softmax
//     registers into the MFMA with 8 v_cvt_pk and no cross-lane traffic at all.
// Layouts (gfx950 v_mfma_f32_32x32x16_bf16): A lane l = A[i = l & 31][k = (l >> 5) * 8 + e]; B lane l = B[k = (l >> 5) * 8 + e][j = l & 31];
// C/D lane l, r = 0..15: D[i = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3)][j = l & 31].
// With key(i) = tile * 32 + swap23(i): score register r of lane-half hh holds key offset 16 * (r >> 3) + 8 * hh + (r & 7).
#include <cstdlib>
#include <type_traits>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace lcc {

typedef __attribute__((ext_vector_type(16))) float f32x16;

LCC_DEVICE f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
LCC_DEVICE int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
LCC_DEVICE float vmax(float a, float b) { return __builtin_fmaxf(a, b); }
LCC_DEVICE float vmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // hipcc emits v_max3_f32
// max over the two lanes l and l ^ 32 with ONE v_permlane32_swap (instead of a ds_bpermute round trip through the LDS crossbar)
LCC_DEVICE float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
LCC_DEVICE float xor32_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ unsigned int lcc_attn32_zero_page[256];
__device__ unsigned int lcc_attn32_ones_page[256] = {[0 ... 255] = 0x3F803F80u};      // 1 KB of bf16 1.0: the V^T rows 80..95 of the tower kernel (LM)

// (Round 6, measured null and removed: a static `s_setprio 1` for the younger or for the older half of the waves -- MI355X_MICROARCH "Two
// waves per SIMD" item 4 -- 416-421 us either way at 8 x 386 rows x 6.2k keys, tower 20.8 ms either way: profiles/r06/attn_static_prio_ab.jsonl.)

// ------------------------------------------------------------------------------------------------------------------------------
// The key loop shared by the LLM prefill kernel (D = 128) and the ViT kernel (D = 80): software pipeline, one key tile per REGION.
// The softmax of tile t (vector pipe: ~70 VALU instructions) is issued next to the MFMAs that do not depend on it -- the P . V of tile
// t-1 and the K . Q^T of tile t+1 (matrix pipe: 16 x 32 cycles at D = 128) -- in ONE scheduling region, interleaved by
// sched_group_barrier: a wave always has matrix AND vector work in flight.  (With one tile at a time per wave the two waves of a SIMD
// ran the same phase in lock step behind the block barrier: the first version of the D = 128 kernel, 0.57 PF at 8 x 386 rows x 6k
// keys against 0.54 for the 16x16x32 kernel; pipelined: 0.76 PF, 1.36 vs 1.93 ms for a 4,096-row piece against 20k keys.)
//
// Ring of NSTAGE = 10 tile stages = five pair-stages (the register budget allows one block per CU anyway).  At the top of the
// iteration of pair P = (t, t+1): pairs P-1 (its second tile still owes its P.V), P and P+1 have landed (counted vmcnt: only pair P+2,
// issued one iteration ago, may still be in flight; barrier: every wave's pieces), every wave is done with pair P-2, whose stages
// receive pair P+3.  The iteration enters with sc_a = K.Q^T of tile t and leaves with sc_a = K.Q^T of tile t+2.
// LDS image of a stage: KP = D/16 K pieces (32 keys x 16 d each, rows in swap23 order) then VP = 2 * ceil(D/32) V^T pieces
// (32 d rows x 16 keys each), every piece 1 KB in MFMA fragment (lane) order.
//   mlim: tiles reaching past it need the mask (causal diagonal / end of the key range); lim: this lane's effective key limit.
//
// (Round 5, measured null and removed: reading the next region's first three V^T fragments at the end of the region before it -- no LDS
// round trip in front of a region's first MFMAs -- 377-382 us either way at 8 x 386 rows x 6.2k keys, profiles/r05/attn_pipe_ab.txt.)
// (Round 6, measured and removed -- the first two were tests of "the kernel is bound by the staging of its K / V tiles":
//   * ROW-MAJOR stage images (every LDS-DMA instruction copies one contiguous KB of the cache -- 4 K rows / 16 V^T rows -- into an XOR-swizzled
//     [32][256 B] / [D][64 B] image, conflict-free fragment reads, same bits) instead of the fragment-order pieces below, whose DMA gathers 64
//     separate 16-byte chunks: 8 x 386 rows 427.8 / 429.0 us vs 431.3 / 420.4, one-shot piece 690 / 692 vs 687 / 685, tower 20.80-20.88 ms vs
//     20.52 (8 K pieces instead of 5 there): a null; profiles/r06/attn_rowmajor_staging_ab.jsonl, tower_rowmajor_staging_ab.jsonl.
//   * one wave per SIMD (the 7 heads of a KV head as two 4-wave blocks with a ring each): 675 vs 420 us,
//     profiles/r06/attn_one_wave_per_simd_ab.jsonl.  NOT a clean test, found afterwards: the 4-wave instantiation was compiled in the AGPR form
//     (see attn_gqa32_kernel) and issued 329 instructions per region against the 8-wave kernel's 185 (of which 32 v_pk_mul sit in the skipped
//     rescale branch: 16 MFMAs, 16 fragment reads, 83 vector and ~38 scalar instructions per region and wave).  A 4-wave block with 1.8 x the
//     instructions took 0.8 of the 8-wave block's time, so a wave alone on its SIMD runs its stream ~2.2 x faster than one that shares it:
//     the two waves of a SIMD do not overlap their matrix and vector work, they mostly take turns.
//   * PING-PONG phases ("two waves of a SIMD in the same phase"): a region issued as a matrix phase (its 16 / 11 MFMAs with
//     their fragment reads) and a vector phase (the softmax), in OPPOSITE order on the two waves that share a SIMD (w and w + 4; both phases
//     of region t depend only on region t - 1; two copies of the key loop under a wave-uniform branch, 235 VGPRs, same bits) instead of the
//     instruction-by-instruction interleave below: 8 x 386 rows 429-432 vs 382-387 us (+11 %), a 4,096-row piece 708 vs 655 (+8 %), one
//     chunk 81.5 vs 74.0, tower 20.59 vs 20.38 ms; profiles/r06/attn_pingpong_phases_ab.jsonl.  The interleave already overlaps a wave's own
//     MFMAs with its own vector instructions; separating them loses that and the other wave does not make up for it.  Removed.)
// LM (round 6, the tower kernel only): the softmax denominator comes out of the P . V MFMAs -- the V^T rows 80..95 of the third d-tile do not
// exist at d = 80 and are fed with ONES, so O^T rows 80..95 accumulate sum_k bf16(P) under the same lazy rescale as O -- instead of 16 v_add +
// the l_run update per region (VERDICT r5 next #2b).  l then sums the ROUNDED probabilities the numerator uses (fp32 accumulation either way).
template <int D, int PW, bool LM, class Issue>
LCC_DEVICE void attn32_key_loop(const u32x4* alds, Issue issue, const u32x4 (&qf)[D / 16], int tb, int te, int mlim, int lim, float scale_log2e,
                                bool active, int lane, int hh, f32x16 (&o)[(D + 31) / 32], float& m_run, float& l_run) {
  constexpr int KP = D / 16, DT = (D + 31) / 32, VP = 2 * DT, NP = KP + VP, NSTAGE = 10;
  auto stage_of = [&](int t) { return alds + ((t - tb) % NSTAGE) * (NP * 64); };
  bf16x8 p_prev[2];                                     // P of the previous tile, its P . V still pending
  p_prev[0] = as_bf16x8((u32x4){0u, 0u, 0u, 0u});
  p_prev[1] = p_prev[0];
  const u32x4* v_prev = alds;                           // V stage of that tile (while nothing is pending P = 0: any landed stage will do)
  f32x16 sc_a, sc_b;

  auto region = [&](auto masked_tag, int t, f32x16& sc_cur, f32x16& sc_nxt) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const u32x4* vs = v_prev;
    const u32x4* kn = stage_of(t + 1);
    // matrix pipe: P.V of tile t-1 ...
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) o[dt] = mfma32(as_bf16x8(vs[(KP + dt * 2 + ss) * 64 + lane]), p_prev[ss], o[dt]);
    // ... and K.Q^T of tile t+1
#pragma unroll
    for (int r = 0; r < 16; ++r) sc_nxt[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KP; ++ks) sc_nxt = mfma32(as_bf16x8(kn[ks * 64 + lane]), as_bf16x8(qf[ks]), sc_nxt);
    // vector pipe: softmax of tile t
    const int kb = t * 32;
    if (MASKED) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc_cur[r] = (kb + 16 * (r >> 3) + 8 * hh + (r & 7)) < lim ? sc_cur[r] : -INFINITY;
    }
    float mx = vmax3(sc_cur[0], sc_cur[1], sc_cur[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = vmax3(mx, sc_cur[r], sc_cur[r + 1]);
    mx = vmax(mx, sc_cur[15]);
    mx = xor32_max(mx);
    const float m_new = vmax(m_run, mx * scale_log2e);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);       // m_run = -inf -> 0
    m_run = m_new;
    float p[16], psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(fmaf(sc_cur[r], scale_log2e, -m_use));   // masked scores are -inf -> 0
      if (!LM) psum += p[r];
    }
    if (!LM) l_run = l_run * alpha + psum;
    bf16x8 pn[2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
      pn[ss] = as_bf16x8((u32x4){pack2(p[8 * ss + 0], p[8 * ss + 1]), pack2(p[8 * ss + 2], p[8 * ss + 3]),
                                 pack2(p[8 * ss + 4], p[8 * ss + 5]), pack2(p[8 * ss + 6], p[8 * ss + 7])});
    // keep the exponentials and the packing of P INSIDE this region: P is only consumed by the next region's MFMAs, and LLVM otherwise
    // sinks its computation behind the rescale branch below, where no matrix work is left to hide it
    {
      u32x4 w0 = as_u32x4(pn[0]), w1 = as_u32x4(pn[1]);
      asm volatile("" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]), "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]), "+v"(l_run));
      pn[0] = as_bf16x8(w0); pn[1] = as_bf16x8(w1);
    }
    // issue order: a few fragment reads ahead, then per MFMA one more read and a handful of vector instructions
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < NP - 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, ((MASKED ? 7 : 5) - (LM ? 1 : 0)) * 16 / NP, 0);
    }
    // region boundary: the accumulators now hold tiles <= t-1 at the OLD maximum; bring them to the new one before tile t's P.V.
    // Lazy: once the running maximum has settled alpha is exactly 1.0 in every lane and the multiplies are skipped (x * 1.0f is
    // exact: bit-identical to always rescaling)
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    p_prev[0] = pn[0]; p_prev[1] = pn[1];
    v_prev = stage_of(t);
  };

  issue(tb); issue(tb + 1); issue(tb + 2); issue(tb + 3); issue(tb + 4); issue(tb + 5);
  for (int t = tb; t < te; t += 2) {
    if (PW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (PW == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (PW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(t + 6); issue(t + 7);
    if (!active) continue;
    if (t == tb) {
      const u32x4* k0 = stage_of(tb);
#pragma unroll
      for (int r = 0; r < 16; ++r) sc_a[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KP; ++ks) sc_a = mfma32(as_bf16x8(k0[ks * 64 + lane]), as_bf16x8(qf[ks]), sc_a);
    }
    if ((t + 1) * 32 > mlim) region(std::true_type{}, t, sc_a, sc_b); else region(std::false_type{}, t, sc_a, sc_b);
    if ((t + 2) * 32 > mlim) region(std::true_type{}, t + 1, sc_b, sc_a); else region(std::false_type{}, t + 1, sc_b, sc_a);
  }
  if (active && te > tb) {     // the last tile's P.V (an empty key range never filled the ring: nothing to flush, and 0 x garbage = NaN)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) o[dt] = mfma32(as_bf16x8(v_prev[(KP + dt * 2 + ss) * 64 + lane]), p_prev[ss], o[dt]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA of this block may land after it has left the CU
}

// ------------------------------------------------------------------------------------------------------------------------------
// LLM prefill (causal, GQA): grid = (query tiles, KV heads, key splits); NWAVE waves: wave w < G computes query head hk * G + w, every
// wave feeds the DMA ring.  Tile tables as attn_prefill_kernel: stream slot, first row in q, valid rows (<= 32), cache index of row 0.
// Round 6, two changes of the work mapping (both leave every (row, head)'s arithmetic untouched: bit-identical outputs):
//  * XCD-chunked block order (`xcd_chunks`): the grid is 1-D; work item = (key split, KV head, query tile) with the query tile fastest, and
//    the items are dealt to the 8 XCDs in CONTIGUOUS chunks (block b runs on XCD b % 8 -- observed placement, used for speed only): the
//    query tiles of one stream that read the SAME keys of a (KV head, split) then run on ONE XCD and share its L2.  Before, blockIdx.x
//    = query tile put the 13 tiles of a 386-row chunk on 8 different XCDs, i.e. every K / V byte was pulled into 8 L2s (8 x 386 rows x
//    6.5k keys: 1.4 GB of L2 fills per launch = 3.4 TB/s, all of it LDS-DMA latency in front of the ring's vmcnt waits).
//  * (row, head) pair packing (`pack`): a wave's 32 query columns are 32 consecutive (head, row) PAIRS of the tile's nq x G pairs, head-major
//    (pair p = head_local * nq + row), instead of "wave w = head w, column = row".  With nq = 32 that is the same thing; a ragged tile
//    (386 rows = 12 x 32 + 2) packs its 2 x 7 = 14 pairs into ONE wave instead of running seven waves with two live columns each -- 1/7
//    of the MFMA work for 1/13 of a chunk's blocks.
// __launch_bounds__(.., 2): never the AGPR form of the MFMAs -- the 4-wave instantiation (G <= 4) was built with 512 registers allowed and
// copied O and the scores between AGPRs and VGPRs in every region (144 v_accvgpr_read / write on top of 115 vector instructions; round-6 audit,
// tools/audit_agpr_copies.py); 236 VGPRs now, no copies.
template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 2) void attn_gqa32_kernel(
    const bf16_t* __restrict__ q, bf16_t* __restrict__ out, const int32_t* __restrict__ tile_stream,
    const int32_t* __restrict__ tile_q0, const int32_t* __restrict__ tile_nq, const int32_t* __restrict__ tile_pos0,
    bf16_t* const* __restrict__ kv_base, KvLayout lay, int layer, int heads, float scale_log2e, int nsplit,
    float* __restrict__ ws_o, float* __restrict__ ws_ml, int n_tiles, int xcd_chunks, int pack) {
  constexpr int D = 128, KP = 8, VP = 8, NP = KP + VP, NSTAGE = 10, PW = (NP + NWAVE - 1) / NWAVE;
  extern __shared__ __attribute__((aligned(16))) u32x4 alds[];      // NSTAGE x NP pieces of 1 KB in MFMA fragment (lane) order
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, hh = lane >> 5;
  const int n_items = n_tiles * lay.n_kv_heads * nsplit;
  int item = blockIdx.x;
  if (xcd_chunks) item = (int)(blockIdx.x & 7) * ((n_items + 7) >> 3) + (int)(blockIdx.x >> 3);
  if (item >= n_items) return;                           // (the grid is rounded up to 8 chunks; the whole block leaves before any barrier)
  const int grp = item % n_tiles, hk = (item / n_tiles) % lay.n_kv_heads, split = item / (n_tiles * lay.n_kv_heads);
  const int G = heads / lay.n_kv_heads;
  const int strm = tile_stream[grp], q0 = tile_q0[grp], nq = tile_nq[grp], pos0 = tile_pos0[grp];
  // this lane's (head, row) pair
  const int pairs = nq * G;
  const int pidx = min(wave * 32 + col, pairs - 1);      // clamped: surplus columns compute a valid pair and are not stored
  const int hl = pack ? pidx / nq : min(wave, G - 1);
  const int qr = pack ? pidx - hl * nq : min(col, nq - 1);
  const int h = hk * G + hl;
  const bool active = pack ? wave * 32 < pairs : wave < G;
  const bool valid = pack ? wave * 32 + col < pairs : (wave < G && col < nq);
  const bf16_t* base = kv_base[strm] + (size_t)layer * lay.layer_stride();
  const bf16_t* kbase = base + (size_t)hk * lay.head_stride();
  const bf16_t* vbase = base + lay.kv_stride() + (size_t)hk * lay.head_stride();
  const int nkeys = pos0 + nq, ntile = (nkeys + 31) / 32;
  const int per = (ntile + nsplit - 1) / nsplit;
  const int tb = nsplit > 1 ? min(ntile, split * per) : 0;
  const int te = nsplit > 1 ? min(ntile, tb + per) : ntile;
  const int ldq = heads * D;
  const int key_limit = pos0 + qr + 1;                   // causal, bottom-right aligned: keys 0 .. pos (inclusive)

  // Q^T fragments (B operand of K . Q^T), resident for the whole key loop: Q[row qr][d = ks * 16 + hh * 8 .. + 8]
  u32x4 qf[KP];
  {
    const bf16_t* qp = q + (size_t)(q0 + qr) * ldq + h * D + hh * 8;
#pragma unroll
    for (int ks = 0; ks < KP; ++ks) qf[ks] = ld16(qp + ks * 16);
#pragma unroll
    for (int ks = 0; ks < KP; ++ks) settle_load(qf[ks]);     // before the first LDS-DMA piece (common.h: glds16)
  }

  // DMA sources of this wave's pieces for key tile 0 + per-tile strides (computed once: a DMA issue is one multiply-add per piece).
  // The KV cache is allocated in whole 32-key tiles, so the rows of the last (partial) tile exist and are masked.
  const bf16_t* pbase[PW];
  size_t pstride[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int p = min(j * NWAVE + wave, NP - 1);         // surplus slots re-fetch the last piece (same bytes, same place)
    if (p < KP) {                                        // K piece ks = p: rows in swap23 order, 16 d per lane-half pair
      pbase[j] = kbase + (size_t)swap23(col) * D + p * 16 + hh * 8;
      pstride[j] = (size_t)32 * D;
    } else {                                             // V^T piece (dt, s): V^T[d = dt * 32 + col][keys s * 16 + hh * 8 .. + 8]
      const int dt = (p - KP) >> 1, s = (p - KP) & 1;
      pbase[j] = vbase + (size_t)(dt * 32 + col) * 32 + s * 16 + hh * 8;
      pstride[j] = (size_t)D * 32;
    }
  }
  auto issue = [&](int t) {
    const int tc = min(t, ntile - 1);
    u32x4* sbase = alds + ((t - tb) % NSTAGE) * (NP * 64);
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int p = min(j * NWAVE + wave, NP - 1);
      glds16((pbase[j] + (size_t)tc * pstride[j]), lds_addr(sbase + p * 64));
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  int min_limit = key_limit;                             // wave-wide minimum of the key limits: tiles entirely below it need no mask
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) min_limit = min(min_limit, __shfl_xor(min_limit, off, 64));
  attn32_key_loop<D, PW, false>(alds, issue, qf, tb, te, min(min_limit, te * 32), min(key_limit, te * 32), scale_log2e, active, lane, hh, o, m_run, l_run);
  if (!active) return;

  float l = xor32_sum(l_run);
  if (nsplit > 1) {      // partial (o, m, l) of this key split; attn_prefill_combine_kernel merges them
    if (valid) {
      const size_t slot = ((size_t)(q0 + qr) * heads + h) * nsplit + split;
      float* op = ws_o + slot * D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          *reinterpret_cast<f32x4*>(op + dt * 32 + 8 * r4 + 4 * hh) =
              (f32x4){o[dt][4 * r4], o[dt][4 * r4 + 1], o[dt][4 * r4 + 2], o[dt][4 * r4 + 3]};
      if (hh == 0) { ws_ml[slot * 2] = m_run; ws_ml[slot * 2 + 1] = l; }
    }
    return;
  }
  const float inv = 1.f / l;
  if (valid) {
    bf16_t* op = out + (size_t)(q0 + qr) * ldq + h * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        st8(op + dt * 32 + 8 * r4 + 4 * hh, (u32x2){pack2(o[dt][4 * r4] * inv, o[dt][4 * r4 + 1] * inv),
                                                   pack2(o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv)});
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// ViT attention (Q2VL:375-417: non-causal inside each temporal slice, d = 80) on the same key loop.  grid = (groups of NWAVE x 32
// query rows of ONE segment, heads): wave w owns the 32 rows q0 + 32 w of the group, all waves share the segment's K / V^T tiles of
// this head through the ring.  d = 80 is 5 k-steps of the 32x32x16 MFMA for K.Q^T (no padding; the 16x16x32 kernel pads 80 -> 96)
// and 3 d-tiles for O^T (the third half empty: its V^T rows 80..95 come from a zero page).  qkv = [P, 3E] with q, k rotated in place,
// vt = V blocked-transposed [head][32-key block][80][32] (vit_rope_vt_kernel); keys past the segment end are masked, and the K rows of
// the last (partial) tile are clamped into the segment (the next segment's rows are NOT part of this attention).
// (the 4-wave instantiation keeps one wave per SIMD and the AGPR form: capped at 256 registers it spills 52 bytes; it is an opt-in variant,
// LCC_VIT32_MIN_BLOCKS4, and its 118 accumulator copies per region are what made it lose to the 8-wave form in rounds 3 and 6)
template <int NWAVE, bool LM>
__global__ __launch_bounds__(NWAVE * 64, 2) void attn_vit32_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, const int32_t* __restrict__ grp_seg,
    const int32_t* __restrict__ grp_q0, const int32_t* __restrict__ seg_start, const int32_t* __restrict__ seg_len,
    const int32_t* __restrict__ seg_blk_start, int heads, int total_blocks, float scale_log2e, int n_groups, int xcd_chunks) {
  constexpr int D = 80, KP = 5, DT = 3, VP = 6, NP = KP + VP, NSTAGE = 10, PW = (NP + NWAVE - 1) / NWAVE;
  extern __shared__ __attribute__((aligned(16))) u32x4 alds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, hh = lane >> 5;
  // one (group, head) per block, or a persistent walk under the grid cap (gemm.hip: g_grid_cap): virtual block vb = head * n_groups + group.
  // xcd_chunks (round 6): the virtual blocks are dealt to the 8 XCDs in contiguous chunks (block b runs on XCD b % 8: speed only), so the
  // ~6 row groups that read the same segment's K / V of one head share ONE L2 instead of landing on six
  const int nvb = n_groups * heads, c8 = (nvb + 7) >> 3, vlimit = xcd_chunks ? 8 * c8 : nvb;
  for (int vb0 = blockIdx.x; vb0 < vlimit; vb0 += gridDim.x) {
  const int vb = xcd_chunks ? (vb0 & 7) * c8 + (vb0 >> 3) : vb0;
  if (vb >= nvb) break;
  const int h = vb / n_groups, grp = vb - h * n_groups, E = heads * D, ld = 3 * E;
  const int sg = grp_seg[grp], q0 = grp_q0[grp] + wave * 32;
  const int s0 = seg_start[sg], sl = seg_len[sg];
  const bf16_t* kbase = qkv + (size_t)s0 * ld + E + h * D;
  const bf16_t* vbase = vt + ((size_t)h * total_blocks + seg_blk_start[sg]) * (D * 32);
  const int nkeys = sl, ntile = (nkeys + 31) / 32;
  const bool active = q0 < sl;
  const int nq = max(0, min(32, sl - q0));
  const int qrow = min(q0 + min(col, max(nq - 1, 0)), sl - 1);      // clamped: inactive waves / surplus columns read a valid row

  u32x4 qf[KP];
  {
    const bf16_t* qp = qkv + (size_t)(s0 + qrow) * ld + h * D + hh * 8;
#pragma unroll
    for (int ks = 0; ks < KP; ++ks) qf[ks] = ld16(qp + ks * 16);
#pragma unroll
    for (int ks = 0; ks < KP; ++ks) settle_load(qf[ks]);     // before the first LDS-DMA piece (common.h: glds16)
  }
  // DMA sources: ONE 32-bit element offset per piece and lane from the wave-uniform base of the piece (K: kbase, V^T: vbase) -- the piece
  // index is wave-uniform, so base and per-tile stride stay in SGPRs (round 6: the 64-bit pointer + stride + row per piece of the first
  // version cost 5 VGPRs per piece in a kernel that sits at the 256-register limit).  V^T rows 80..95 of the third d-tile do not exist:
  // their lanes fetch row 79 again (finite values; the output rows >= 80 are never stored).
  const int krow = swap23(col);
  unsigned poff[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int p = min(j * NWAVE + wave, NP - 1);
    if (p < KP) {
      poff[j] = (unsigned)(krow * ld + p * 16 + hh * 8);
    } else {
      const int dt = (p - KP) >> 1, s = (p - KP) & 1, d = min(dt * 32 + col, D - 1);
      poff[j] = (unsigned)(d * 32 + s * 16 + hh * 8);
    }
  }
  auto issue = [&](int t) {
    const int tc = min(t, ntile - 1);
    u32x4* sbase = alds + (t % NSTAGE) * (NP * 64);
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int p = min(j * NWAVE + wave, NP - 1);
      const bool isk = p < KP;                           // wave-uniform
      const bf16_t* src = (isk ? kbase + (size_t)tc * (32 * ld) : vbase + (size_t)tc * (D * 32)) + poff[j];
      if (isk && tc == ntile - 1)                        // last tile: rows past the segment end are clamped (and masked)
        src = kbase + (size_t)min(tc * 32 + krow, nkeys - 1) * ld + p * 16 + hh * 8;
      if (LM && p >= KP + 4 && col >= 16)                // V^T rows 80..95 (third d-tile, upper half): ones -> O^T rows 80..95 = sum of P
        src = reinterpret_cast<const bf16_t*>(lcc_attn32_ones_page) + lane * 8;
      glds16(src, lds_addr(sbase + p * 64));
    }
  };
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  attn32_key_loop<D, PW, LM>(alds, issue, qf, 0, ntile, sl, sl, scale_log2e, active, lane, hh, o, m_run, l_run);
  const float l = LM ? o[2][8] : xor32_sum(l_run);       // LM: row 80 + 4 hh of O^T, every one of rows 80..95 holds the sum over ALL keys
  if (active && col < nq) {
    const float inv = 1.f / l;
    bf16_t* op = out + (size_t)(s0 + q0 + col) * E + h * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = dt * 32 + 8 * r4 + 4 * hh;
        if (d < D)
          st8(op + d, (u32x2){pack2(o[dt][4 * r4] * inv, o[dt][4 * r4 + 1] * inv), pack2(o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv)});
      }
  }
  if (vb0 + (int)gridDim.x < vlimit) __syncthreads();     // the next (group, head)'s DMA ring reuses the stages
  }
}

// LCC_VIT32_LSUM_MFMA (A/B, read once): 1 = the tower kernel takes its softmax denominator from the P . V MFMAs (LM above), 0 = vector adds
static int vit32_lsum_mfma() {
  static const int v = [] { const char* e = getenv("LCC_VIT32_LSUM_MFMA"); return e ? atoi(e) : 1; }();
  return v;
}

template <int NWAVE, bool LM>
static void vit32_launch_t(const bf16_t* qkv, const bf16_t* vt, bf16_t* out, const int32_t* grp_seg, const int32_t* grp_q0, const int32_t* seg_start,
                           const int32_t* seg_len, const int32_t* seg_blk_start, int n_groups, int heads, int total_blocks, float scale_log2e, hipStream_t st) {
  constexpr size_t lds = (size_t)10 * 11 * 1024;
  static DeviceOnce once;   // per instantiation
  if (once.first()) (void)hipFuncSetAttribute((const void*)attn_vit32_kernel<NWAVE, LM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const long nvb = (long)n_groups * heads;
  const int cap = get_grid_cap();
  static const int xcd = [] { const char* v = getenv("LCC_ATTN32_XCD"); return v ? atoi(v) : 1; }();      // A/B: 0 = virtual block = physical block
  const long nb = (cap > 0 && nvb > cap) ? cap : (xcd ? (nvb + 7) / 8 * 8 : nvb);      // (the cap is a multiple of 8: gemm.hip set_grid_cap)
  attn_vit32_kernel<NWAVE, LM><<<dim3((unsigned)nb), dim3(NWAVE * 64), lds, st>>>(
      qkv, vt, out, grp_seg, grp_q0, seg_start, seg_len, seg_blk_start, heads, total_blocks, scale_log2e, n_groups, xcd);
}
int attn_vit32_launch(const bf16_t* qkv, const bf16_t* vt, bf16_t* out, const int32_t* grp_seg, const int32_t* grp_q0,
                      const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_groups, int heads,
                      int total_blocks, float scale_log2e, hipStream_t st, int group_rows) {
  if (n_groups <= 0) return 0;
  if (group_rows != 256 && group_rows != 128) return LCC_ERR_ARG;
  // group_rows 256: 8 waves x 32 rows per block; 128: 4 waves (one per SIMD) -- twice the blocks for a grid that does not fill the chip
#define LCC_VIT32_GO(NW, LMV) vit32_launch_t<NW, LMV>(qkv, vt, out, grp_seg, grp_q0, seg_start, seg_len, seg_blk_start, n_groups, heads, total_blocks, scale_log2e, st)
  if (vit32_lsum_mfma()) { if (group_rows == 256) LCC_VIT32_GO(8, true); else LCC_VIT32_GO(4, true); }
  else { if (group_rows == 256) LCC_VIT32_GO(8, false); else LCC_VIT32_GO(4, false); }
#undef LCC_VIT32_GO
  g_launch_counts[LC_ATTN_VIT32] += 1;
  return 0;
}

// Query rows per tile.  With the (row, head) pair packing a block serves nq x G pairs in NWAVE x 32 columns, so a tile may hold up to
// NWAVE * 32 / G rows: 36 at G = 7 (252 of the 256 columns busy -- with 32-row tiles the eighth wave of every block idles, and a block's time
// is that of its busiest SIMD either way), 42 at G = 6, 32 at G = 8 or 4.  LCC_ATTN32_TILE_ROWS (A/B, read once): 32 = the round-3..6 tiles.
static int attn32_pack_on() {
  static const int pack = [] { const char* v = getenv("LCC_ATTN32_PACK"); return v ? atoi(v) : 1; }();
  return pack;
}
int attn32_max_tile_rows(int G) {
  if (G < 1 || G > 8 || !attn32_pack_on()) return 32;
  return std::min(64, ((G <= 4 ? 4 : 8) * 32) / G);
}
int attn32_tile_rows(int G) {
  static const int forced = [] { const char* v = getenv("LCC_ATTN32_TILE_ROWS"); return v ? atoi(v) : 0; }();
  const int mx = attn32_max_tile_rows(G);
  return forced >= 16 ? std::min(forced, mx) : mx;
}

// Tile height and key-split count of one prefill launch, planned TOGETHER (host logic only: lcc_debug_attn_plan pins it on the CPU).
// One 8-wave block per CU and (tile, KV head, split); a block's time is proportional to its keys and does not depend on how many of its
// waves work.  Cost of a candidate (tile rows, k splits) in units of one unsplit block's time:
//     ceil(tiles x kv_heads x k / cus) / k          rounds of blocks on the chip x keys per block
//   + k x (0.01 + 200 / max_kv)                     what a split costs: its fp32 partials written and merged (~4 us per split against ~0.02 us
//                                                   per key of a block), and a small preference for fewer splits between near-ties
// k <= 8 with >= 8 key tiles per split, and only for S <= 1024 rows (the partial buffers of carve_llm).  The model ranks every measured
// configuration of profiles/r06/attn_tall_tiles_ab.jsonl / attn_tall_tiles_splits_probe.jsonl in the measured order:
//   1 chunk  (386 rows, 6.2k keys): (36, 5) 65.4 us < (36, 4) 72.2 ~ (32, 4) 73.4 << (32, 1) 181
//   2 chunks: (32, 2) 113.7 < (36, 2) 116 < (36, 8) 125.8 ~ (36, 4) 127.4 < (32, 7) 129 < (36, 3) 152.7 < (32, 1) 176
//             (rounds 3-6 maximised "filled fraction of the last round" instead and ran 2-chunk batches at (32, 7): +13 %)
//   3 chunks: (36, 3) 163 < (32, 1) 179 < (36, 4) 190 < (32, 2) 206 < (36, 2) 214  (the engine runs (32, 1) there: 1,158 rows > 1,024)
//   4 / 8 chunks (no split above 1,024 rows): (32, 1) = (36, 1) within 1 %: ties keep 32 rows; 8 first turns: (36, 1) 132 < (32, 1) 140 (4 vs 5 rounds)
void attn32_plan(const int* n_new, int n_streams, int max_kv, int G, int n_kv_heads, int cus, int* tile_rows, int* splits) {
  long S = 0;
  for (int b = 0; b < n_streams; ++b) S += n_new[b];
  const int ks_cap = S <= 1024 ? std::max(1, std::min(8, (max_kv / 32) / 8)) : 1;
  const float per_split = 0.01f + 200.f / (float)std::max(max_kv, 32);
  const int tall = attn32_tile_rows(G);
  float best = 1e30f;
  *tile_rows = 32; *splits = 1;
  for (int pass = 0; pass < (tall > 32 ? 2 : 1); ++pass) {           // 32-row tiles first: they win ties
    const int rows = pass == 0 ? 32 : tall;
    long tiles = 0;
    for (int b = 0; b < n_streams; ++b) tiles += (n_new[b] + rows - 1) / rows;
    const long base = tiles * n_kv_heads;
    for (int k = 1; k <= ks_cap; ++k) {
      const float cost = (float)((base * k + cus - 1) / cus) / (float)k + per_split * (float)k;
      if (cost < best - 1e-6f) { best = cost; *tile_rows = rows; *splits = k; }
    }
  }
}

// launcher: tiles of <= attn32_max_tile_rows(G) rows; the caller (attention.hip: attn_prefill_bf16) runs the split merge
int attn_prefill32_launch(const bf16_t* q, bf16_t* out, const int32_t* tile_stream, const int32_t* tile_q0, const int32_t* tile_nq,
                          const int32_t* tile_pos0, bf16_t* const* kv_base, KvLayout lay, int layer, int n_tiles, int n_q_heads,
                          int nsplit, float* ws_o, float* ws_ml, float scale_log2e, hipStream_t st) {
  const int G = n_q_heads / lay.n_kv_heads;
  if (G < 1 || G > 8 || lay.head_dim != 128) return LCC_ERR_SHAPE;
  constexpr size_t lds = (size_t)10 * 16 * 1024;
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)attn_gqa32_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)attn_gqa32_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int S = nsplit > 1 ? nsplit : 1;
  // LCC_ATTN32_XCD / LCC_ATTN32_PACK (A/B, read once): 0 = the round-3..5 work mapping (blockIdx = tile / KV head / split; wave = head)
  static const int xcd = [] { const char* v = getenv("LCC_ATTN32_XCD"); return v ? atoi(v) : 1; }();
  const int pack = attn32_pack_on();
  const long n_items = (long)n_tiles * lay.n_kv_heads * S;
  const dim3 grid((unsigned)(xcd ? ((n_items + 7) / 8) * 8 : n_items));
  if (G <= 4)
    attn_gqa32_kernel<4><<<grid, dim3(256), lds, st>>>(q, out, tile_stream, tile_q0, tile_nq, tile_pos0, kv_base, lay, layer, n_q_heads,
                                                       scale_log2e, S, ws_o, ws_ml, n_tiles, xcd, pack);
  else
    attn_gqa32_kernel<8><<<grid, dim3(512), lds, st>>>(q, out, tile_stream, tile_q0, tile_nq, tile_pos0, kv_base, lay, layer, n_q_heads,
                                                       scale_log2e, S, ws_o, ws_ml, n_tiles, xcd, pack);
  return 0;
}

}  // namespace lcc
