// bf16 GEMMs of the LiveCC hot path on gfx950 (MI355X):  C[M,N] = A[M,K] * W[N,K]^T  (+bias, +epilogue)
//
// W keeps the nn.Linear layout [N,K] (K contiguous) so that both MFMA operands are read as 8 contiguous
// bf16 (16 bytes) per lane.  fp32 accumulation, one rounding to bf16 at the points where HF rounds.
//
//  * gemm_big_kernel     : M >= 17, packed W, K % 64 == 0 (LLM prefill, ViT).  MFMA-bound.  8 waves, BM x 256 x 64 tile
//                          (BM 256 / 192 / 128), LDS-DMA for both operands, optional fp8-e4m3 W converted after the LDS read.
//  * gemm_vh_kernel      : the same pipeline over row tiles of 256 / 272 / 288 rows (no ragged last row tile; round 5).
//  * gemm_tall_kernel    : one block row covers all of M (256 < M <= 448: one streaming chunk's gate/up).
//  * gemm_glds_kernel /   : shapes with few 256-column tiles, row-major W, K % 64 != 0.  64x128x64 (128x128x64) tile, 4 waves,
//    gemm_tiled_kernel     LDS-DMA 3-stage ring / register-staged double buffer with an XOR swizzle, XCD-aware block remap.
//  * gemv_skinny_kernel  : M <= 64 (decode, lm_head; MG = ceil(M / 16) activation fragments per weight fragment).  HBM-bound weight streaming.  4 waves per block share 16 (or 32) W rows,
//                          each wave a slice of K; W fragments go straight from HBM into the MFMA operand registers (no LDS
//                          round trip - the operand is used once), split-K partial sums as fp32 slabs that the consumer reduces.
//  * gemv_w8_kernel      : the same for fp8-e4m3 weights + fp32 row scales (PACKED8 order), e4m3 -> bf16 exactly in registers.
//
// Replaces: every nn.Linear / Conv3d of HF modeling_qwen2_vl.py on the path
//   (PatchEmbed 251-274, VisionAttention.qkv/proj 349-350, VisionMlp 293-301, PatchMerger 277-290,
//    Qwen2VLAttention q/k/v/o_proj 501-504, Qwen2MLP 453-466, lm_head 1218/1323).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "tail_ops.h"

namespace lcc {

__device__ unsigned int lcc_zero_page[256];  // 1 KB of zeros: operand source of absent K chunks (address select, no branch)

// Grid cap (round 5): while it is set (engine_vit.hip sets it around a vision-tower call that runs on the side stream UNDER another turn's
// decode steps), the MFMA-bound tile kernels launch at most `cap` workgroups and walk their tiles persistently (tile = block, block +
// grid, ...), so that they occupy `cap` of the 256 CUs instead of all of them.  Why: a resident 8-wave GEMM block (2 x 208-216 VGPRs per
// SIMD, 128-140 KB of LDS) leaves room for ONE weight-streaming decode wave per SIMD where five fit on a free CU -- with the tower spread
// over every CU the HBM-bound decode kernels lose their memory-level parallelism and the "overlapped" tower simply adds its own duration
// to the decode steps (8 streams: 4.88 vs 3.68 ms per step while a 20-ms tower runs; 1 stream: 3.23 vs 2.95 ms).  Half the CUs stream the
// weights at the full HBM rate.  A multiple of 8 (virtual block id and physical block id then agree on the XCD).
// thread_local: the cap belongs to the host thread that is inside lcc_vit_encode -- a second thread launching prefill GEMMs
// meanwhile is not capped, and two engines with different caps do not race.
static thread_local int g_grid_cap = 0;
void set_grid_cap(int cap) { g_grid_cap = cap <= 0 ? 0 : std::max(8, cap & ~7); }
int get_grid_cap() { return g_grid_cap; }
static inline unsigned capped_grid(long nblk) { return (unsigned)((g_grid_cap > 0 && nblk > g_grid_cap) ? g_grid_cap : nblk); }

// epilogue shared by the tiled kernels.  acc[i][j][r] = C[mbase + i*16 + li][nbase + j*16 + g*4 + r] (swapped operands).
template <int EPI, int MT, int NT, int AM = MT, int AN = NT>   // the first MT x NT tiles of an AM x AN accumulator array
LCC_DEVICE void tile_epilogue(const f32x4 (&acc)[AM][AN], int mbase, int nbase, int ocbase, int li, int g,
                              const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int ldr,
                              bf16_t* __restrict__ C, int ldc, int M, int N, float* __restrict__ partial,
                              const float* __restrict__ wscale) {
  // per-output-row dequantisation scale of fp8 weights (1.0 for bf16 weights: x * 1.0 is exact)
  f32x4 sc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) sc[j] = (f32x4){1.f, 1.f, 1.f, 1.f};
  if (wscale != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j) sc[j] = *reinterpret_cast<const f32x4*>(wscale + min(nbase + j * 16 + g * 4, N - 4));
  }
  // Loads (bias: once per column group; residual: NT per row tile) are issued unconditionally from clamped addresses and
  // back to back, so that they overlap instead of one L2 round trip per (row tile, column group); only the stores are
  // predicated on the M / N edges.
  u32x2 bv[NT];
  if (EPI != EPI_PARTIAL && EPI != EPI_SWIGLU) {
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = (u32x2){0u, 0u};
    if (bias != nullptr) {
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = ld8(bias + min(nbase + j * 16 + g * 4, N - 4));
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mbase + i * 16 + li;
    const bool mok = m < M;
    const int mc = min(m, M - 1);
    if (EPI == EPI_PARTIAL) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = nbase + j * 16 + g * 4;
        if (mok && n < N) *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * M + m) * N + n) = acc[i][j] * sc[j];
      }
    } else if (EPI == EPI_SWIGLU) {
#pragma unroll
      for (int j = 0; j < NT; j += 2) {
        const int n = nbase + j * 16 + g * 4;  // column in the interleaved [gate16|up16] space
        const int oc = ocbase + (j / 2) * 16 + g * 4;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float gate = rbf(acc[i][j][r] * sc[j][r]), up = rbf(acc[i][j + 1][r] * sc[j + 1][r]);
          o[r] = silu_bf16(gate) * up;
        }
        if (mok && n < N) st8(C + (size_t)m * ldc + oc, (u32x2){pack2(o[0], o[1]), pack2(o[2], o[3])});
      }
    } else {
      u32x2 rv[NT];
      if (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int j = 0; j < NT; ++j) rv[j] = ld8(residual + (size_t)mc * ldr + min(nbase + j * 16 + g * 4, N - 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = nbase + j * 16 + g * 4;
        float v[4];
        v[0] = rbf(acc[i][j][0] * sc[j][0] + lo2f(bv[j].x));
        v[1] = rbf(acc[i][j][1] * sc[j][1] + hi2f(bv[j].x));
        v[2] = rbf(acc[i][j][2] * sc[j][2] + lo2f(bv[j].y));
        v[3] = rbf(acc[i][j][3] * sc[j][3] + hi2f(bv[j].y));
        if (EPI == EPI_QUICK_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = quick_gelu_bf16(v[r]);
        } else if (EPI == EPI_GELU_ERF) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf_bf16(v[r]);
        } else if (EPI == EPI_RESIDUAL) {
          v[0] += lo2f(rv[j].x); v[1] += hi2f(rv[j].x); v[2] += lo2f(rv[j].y); v[3] += hi2f(rv[j].y);
        }
        if (mok && n < N) st8(C + (size_t)m * ldc + n, (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])});
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// EPI_VIT_QK / EPI_VIT_V (round 4): the vision tower's q|k|v projection writes what its attention kernels read -- q and k rotated (2-D
// RoPE, HF modeling_qwen2_vl.py:225-248) at their natural columns of C, V blocked-transposed into vt[head][32-key block][80][32] -- instead
// of a plain [P, 3E] tensor that vit_rope_vt_kernel re-reads and re-writes (52 us per tower block at 8 streams, 179 MB of traffic).
// Two launches of gemm_big_kernel over the `qkv_w_rope` copy of the weight (weights.py: vit_qkv_rope_row_permutation):
//
//  * q|k (N = 2E, EPI_VIT_QK): inside q and inside k every 32 stored rows are [16 first-half channels | their 16 rotation partners],
//    first-half channels enumerated head-major (f = head * 40 + c, c < 40), so a lane's accumulators acc[i][j] / acc[i][j + 1] hold
//    (x[c .. c+3], x[c+40 .. c+43]) of one head: the rotation is register-local (40 % 4 == 0: a lane's four channels never straddle a
//    head).  The block's cos / sin rows (BM x 40 fp32 each, one contiguous piece of the tables) are DMA'd into the LDS the k-loop has
//    finished with: the first version loaded them per (row tile, column pair) from global memory and hipcc put an s_waitcnt vmcnt(0)
//    -- which on gfx9 also waits for the previous iteration's STORES -- into each of the 16 iterations (+15 us on a 137-us GEMM).
//  * V (N = E, EPI_VIT_V): the MFMA operands are swapped (activations as the A operand), so a lane holds 4 consecutive PATCHES of one
//    channel -- 8 contiguous bytes of vt, four lanes = 32 contiguous bytes, the store granularity of the plain epilogue -- and no
//    transpose is needed.  Same products, same k order: same bits.  grp_off[p / 4] = element offset of patch p's key slot inside a
//    (head, channel 0) plane of vt; segment lengths are multiples of 4 (H, W multiples of 28), so a 4-patch group never straddles a
//    segment or a 32-key block.
// ------------------------------------------------------------------------------------------------
template <int MT, int NT>
LCC_DEVICE void vit_qk_epilogue(const f32x4 (&acc)[MT][NT], int mbase, int nbase, int mrow0, int li, int g, const bf16_t* __restrict__ bias,
                                bf16_t* __restrict__ C, int ldc, int M, int N, int E, const float* lcs, const float* lsn) {
  u32x2 b1[NT / 2], b2[NT / 2];
#pragma unroll
  for (int jp = 0; jp < NT / 2; ++jp) {
    b1[jp] = (u32x2){0u, 0u}; b2[jp] = (u32x2){0u, 0u};
    const int n = min(nbase + jp * 32, N - 32);
    if (bias != nullptr) { b1[jp] = ld8(bias + n + g * 4); b2[jp] = ld8(bias + n + 16 + g * 4); }
  }
#pragma unroll
  for (int jp = 0; jp < NT / 2; ++jp) {
    const int j = jp * 2;
    const int n = nbase + j * 16;            // first column of a 32-column group (wave-uniform)
    if (n >= N) continue;
    const int which = n >= E ? 1 : 0;
    const int f = ((n - which * E) >> 5) * 16 + g * 4;
    const int h = f / 40, c = f - h * 40;
    bf16_t* out = C + which * E + h * 80 + c;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = mbase + i * 16 + li;
      const f32x4 cv = *reinterpret_cast<const f32x4*>(lcs + (mrow0 + i * 16 + li) * 40 + c);
      const f32x4 sv = *reinterpret_cast<const f32x4*>(lsn + (mrow0 + i * 16 + li) * 40 + c);
      float x1[4], x2[4], o1[4], o2[4];
      x1[0] = rbf(acc[i][j][0] + lo2f(b1[jp].x)); x1[1] = rbf(acc[i][j][1] + hi2f(b1[jp].x));
      x1[2] = rbf(acc[i][j][2] + lo2f(b1[jp].y)); x1[3] = rbf(acc[i][j][3] + hi2f(b1[jp].y));
      x2[0] = rbf(acc[i][j + 1][0] + lo2f(b2[jp].x)); x2[1] = rbf(acc[i][j + 1][1] + hi2f(b2[jp].x));
      x2[2] = rbf(acc[i][j + 1][2] + lo2f(b2[jp].y)); x2[3] = rbf(acc[i][j + 1][3] + hi2f(b2[jp].y));
#pragma unroll
      for (int r = 0; r < 4; ++r) vit_rope_pair(x1[r], x2[r], cv[r], sv[r], o1[r], o2[r]);
      if (m < M) {
        st8(out + (size_t)m * ldc, (u32x2){pack2(o1[0], o1[1]), pack2(o1[2], o1[3])});
        st8(out + (size_t)m * ldc + 40, (u32x2){pack2(o2[0], o2[1]), pack2(o2[2], o2[3])});
      }
    }
  }
}

// swapped operands: acc[i][j][r] = C[mbase + i*16 + g*4 + r][nbase + j*16 + li]
template <int MT, int NT>
LCC_DEVICE void vit_v_epilogue(const f32x4 (&acc)[MT][NT], int mbase, int nbase, int li, int g, const bf16_t* __restrict__ bias, int M, int N,
                               const VitQkvEpi& vq) {
  int off[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) off[i] = vq.grp_off[min(mbase + i * 16 + g * 4, M - 4) >> 2];
  bf16_t braw[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) braw[j] = 0;
  if (bias != nullptr) {      // one branch around the four loads (a select per column made hipcc wait for each load separately)
#pragma unroll
    for (int j = 0; j < NT; ++j) braw[j] = bias[min(nbase + j * 16 + li, N - 1)];
  }
  float bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = bf2f(braw[j]);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = nbase + j * 16 + li;
    const int nc = min(n, N - 1);
    const int h = nc / 80, c = nc - h * 80;
    bf16_t* base = vq.vt + ((size_t)h * vq.total_blocks * 80 + c) * 32;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const u32x2 o = (u32x2){pack2(acc[i][j][0] + bv[j], acc[i][j][1] + bv[j]), pack2(acc[i][j][2] + bv[j], acc[i][j][3] + bv[j])};
      if (n < N && mbase + i * 16 + g * 4 < M) st8(base + off[i], o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// tiled GEMM
// ------------------------------------------------------------------------------------------------
template <int BM, int EPI>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
    const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int ldr,
    bf16_t* __restrict__ C, int ldc, int M, int N, int K, int tiles_m, int tiles_n, int w_packed,
    float* __restrict__ partial, int kt_per_split, const float* __restrict__ wscale) {
  constexpr int BN = 128, BK = 64;
  constexpr int WM = BM / 2;       // wave tile rows (of A)
  constexpr int MT = WM / 16;      // 16-row MFMA tiles per wave in M
  constexpr int NT = 4;            // 16-col MFMA tiles per wave in N (wave tile is WM x 64)
  constexpr int AI = BM * 8 / 256; // 16-byte chunks of the A tile per thread
  constexpr int BI = BN * 8 / 256;
  constexpr int STAGE = (BM + BN) * 8;  // 16-byte units per stage
  __shared__ u32x4 smem[2 * STAGE];

  // XCD-aware remap: hardware places block b on XCD b%8; give every XCD a contiguous range of tile ids
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = bid / tiles_m, tm = bid - tn * tiles_m;  // consecutive ids share the W panel
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, g = lane >> 4;

  // global -> register staging addresses
  const bf16_t* aptr[AI];
  const bf16_t* bptr[BI];
  int aoff[AI], boff[BI];  // LDS offsets (16B units) within a stage
  const int chunk = tid & 7;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int row = (tid >> 3) + i * 32;
    int gr = min(m0 + row, M - 1);
    aptr[i] = A + (size_t)gr * lda + chunk * 8;
    aoff[i] = row * 8 + (chunk ^ ((row >> 1) & 7));
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int row = (tid >> 3) + i * 32;
    int gr = min(n0 + row, N - 1);
    // row-major [N,K]: row*ldw + chunk*8, k-tile stride 64.  packed [N/16][Kp/32][4 g][16 rows][8]: the 1-KB fragment
    // sub-tile of (row tile, 32-k block), k-tile (= two 32-k blocks) stride 1024.
    bptr[i] = w_packed ? W + ((size_t)(gr >> 4) * ((K + 31) >> 5) + (chunk >> 2)) * 512 + (chunk & 3) * 128 + (gr & 15) * 8
                       : W + (size_t)gr * ldw + chunk * 8;
    boff[i] = BM * 8 + row * 8 + (chunk ^ ((row >> 1) & 7));
  }
  const int wstride = w_packed ? 1024 : BK;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[AI], rb[BI];
  const bf16_t* zp = reinterpret_cast<const bf16_t*>(lcc_zero_page);
  auto gload = [&](int kt) {
    const int k = kt * BK + chunk * 8;
    const bool ok = k < K;  // K % 8 == 0: a chunk is entirely inside or outside (K tail: load zeros, by address select)
#pragma unroll
    for (int i = 0; i < AI; ++i) ra[i] = ld16(ok ? aptr[i] + kt * BK : zp);
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = ld16(ok ? bptr[i] + (size_t)kt * wstride : zp);
  };
  auto sstore = [&](int buf) {
    u32x4* s = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i) s[aoff[i]] = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) s[boff[i]] = rb[i];
  };

  // split-K (EPI_PARTIAL): blockIdx.y owns k-tiles [kt0, nkt)
  const int kt0 = (EPI == EPI_PARTIAL) ? blockIdx.y * kt_per_split : 0;
  const int nkt = (EPI == EPI_PARTIAL) ? min((K + BK - 1) / BK, kt0 + kt_per_split) : (K + BK - 1) / BK;
  gload(kt0);
  sstore(0);
  __syncthreads();

  for (int kt = kt0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) gload(kt + 1);
    const u32x4* s = smem + ((kt - kt0) & 1) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int row = wm * WM + i * 16 + li;
        fa[i] = as_bf16x8(s[row * 8 + ((kk * 4 + g) ^ ((row >> 1) & 7))]);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        int row = wn * 64 + j * 16 + li;
        fb[j] = as_bf16x8(s[BM * 8 + row * 8 + ((kk * 4 + g) ^ ((row >> 1) & 7))]);
      }
      // swapped operands: D'[n][m] so that a lane owns 4 consecutive n of one row m (8-byte stores)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
    if (more) sstore((kt + 1 - kt0) & 1);
    __syncthreads();
  }

  tile_epilogue<EPI, MT, NT>(acc, m0 + wm * WM, n0 + wn * 64, (n0 + wn * 64) / 2, li, g, bias, residual, ldr, C, ldc, M, N, partial, wscale);
}

// ------------------------------------------------------------------------------------------------
// tiled GEMM v2: LDS-DMA (global_load_lds, 16 B/lane) into a 3-stage LDS ring, counted vmcnt, raw s_barrier
// ------------------------------------------------------------------------------------------------
// Both operands live in LDS in MFMA FRAGMENT ORDER: a 1-KB sub-tile = (16 rows, 32 k) as [4 g][16 rows][8 k], i.e. exactly
// the 64 lanes' 16-byte operands in lane order.  global_load_lds writes wave-uniform-base + lane*16, so one DMA instruction
// fills one sub-tile, and the fragment read is a linear, conflict-free ds_read_b128 (lane reads slot `lane`).  With packed
// weights the W sub-tile is also contiguous in HBM (one 1-KB burst per instruction); row-major operands (activations, or
// an unpacked W) are gathered as 16 rows x 64 B by the per-lane source addresses.
// Pipeline per k-tile:  wait(tile kt landed, leave tile kt+1 in flight) -> s_barrier -> issue DMA of tile kt+2 into the
// stage that was consumed in iteration kt-1 -> MFMAs of tile kt.   No VGPR staging, no ds_write.
template <int BM, int EPI>
__global__ __launch_bounds__(256) void gemm_glds_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
    const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int ldr,
    bf16_t* __restrict__ C, int ldc, int M, int N, int K, int tiles_m, int tiles_n, int w_packed,
    float* __restrict__ partial, int kt_per_split, const float* __restrict__ wscale) {
  constexpr int BN = 128, BK = 64, NSTAGE = 3;
  constexpr int WM = BM / 2, MT = WM / 16, NT = 4;
  constexpr int A_SUB = (BM / 16) * 2;          // 1-KB sub-tiles of the A tile (row tiles x 2 k-blocks)
  constexpr int B_SUB = (BN / 16) * 2;
  constexpr int STAGE = (A_SUB + B_SUB) * 64;   // 16-byte units per stage
  constexpr int A_PER_WAVE = A_SUB / 4, B_PER_WAVE = B_SUB / 4;
  constexpr int G = A_PER_WAVE + B_PER_WAVE;    // DMA instructions per wave (= per thread) per k-tile
  extern __shared__ __attribute__((aligned(16))) u32x4 dsmem[];

  const int nblk = tiles_m * tiles_n;
  for (int vb = blockIdx.x; vb < nblk; vb += gridDim.x) {     // one tile per block, or a persistent walk under the grid cap
  int bid = vb;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = bid / tiles_m, tm = bid - tn * tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, g = lane >> 4;
  const int K32 = (K + 31) >> 5;
  const bf16_t* zp = reinterpret_cast<const bf16_t*>(lcc_zero_page) + lane * 8;

  // per-lane source pointers of this wave's sub-tiles at k-block 0 (advanced by kb*kstep per k-block)
  const bf16_t* asrc[A_PER_WAVE];
  const bf16_t* bsrc[B_PER_WAVE];
  int akb[A_PER_WAVE], bkb[B_PER_WAVE];
#pragma unroll
  for (int q = 0; q < A_PER_WAVE; ++q) {
    const int st = wave * A_PER_WAVE + q, rt = st >> 1;
    akb[q] = st & 1;
    asrc[q] = A + (size_t)min(m0 + rt * 16 + li, M - 1) * lda + g * 8;
  }
#pragma unroll
  for (int q = 0; q < B_PER_WAVE; ++q) {
    const int st = wave * B_PER_WAVE + q, rt = st >> 1;
    bkb[q] = st & 1;
    const int row = min(n0 + rt * 16 + li, N - 1);
    bsrc[q] = w_packed ? W + (size_t)(row >> 4) * K32 * 512 + lane * 8 : W + (size_t)row * ldw + g * 8;
  }
  const int bstep = w_packed ? 512 : 32;   // elements per 32-k block

  auto issue = [&](int kt, int stage) {
    u32x4* sbase = dsmem + stage * STAGE;
#pragma unroll
    for (int q = 0; q < A_PER_WAVE; ++q) {
      const int kb = kt * 2 + akb[q];
      const bf16_t* src = (kb * 32 + g * 8 < K) ? asrc[q] + kb * 32 : zp;   // A: K % 8 == 0, chunk inside or outside
      glds16(src, lds_addr(sbase + (wave * A_PER_WAVE + q) * 64));
    }
#pragma unroll
    for (int q = 0; q < B_PER_WAVE; ++q) {
      const int kb = kt * 2 + bkb[q];
      const bool ok = w_packed ? (kb < K32) : (kb * 32 + g * 8 < K);
      const bf16_t* src = ok ? bsrc[q] + (size_t)kb * bstep : zp;
      glds16(src, lds_addr(sbase + (A_SUB + wave * B_PER_WAVE + q) * 64));
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int kt0 = (EPI == EPI_PARTIAL) ? blockIdx.y * kt_per_split : 0;
  const int nkt = (EPI == EPI_PARTIAL) ? min((K + BK - 1) / BK, kt0 + kt_per_split) : (K + BK - 1) / BK;
  issue(kt0, 0);
  if (kt0 + 1 < nkt) issue(kt0 + 1, 1);

  for (int kt = kt0; kt < nkt; ++kt) {
    // tile kt landed (this wave's pieces); the G DMAs of tile kt+1 may stay in flight across the barrier
    if (kt + 1 < nkt) {
      if (G == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // every wave's pieces of tile kt are in LDS; everyone finished reading tile kt-1
    if (kt + 2 < nkt) issue(kt + 2, (kt + 2 - kt0) % NSTAGE);
    const u32x4* s = dsmem + ((kt - kt0) % NSTAGE) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = as_bf16x8(s[((wm * MT + i) * 2 + kk) * 64 + lane]);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = as_bf16x8(s[(A_SUB + (wn * NT + j) * 2 + kk) * 64 + lane]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
  }
  tile_epilogue<EPI, MT, NT>(acc, m0 + wm * WM, n0 + wn * 64, (n0 + wn * 64) / 2, li, g, bias, residual, ldr, C, ldc, M, N, partial, wscale);
  if (vb + (int)gridDim.x < nblk) __syncthreads();     // the next tile's DMA reuses the stages
  }
}

// ------------------------------------------------------------------------------------------------
// tiled GEMM v3: 8 waves, BM x 256 x 64 block tile (BM = 256 or 128), LDS-DMA for both operands
// ------------------------------------------------------------------------------------------------
// The 64x128 / 128x128 tiles above move (1/BM + 1/BN) bytes per flop from L2 into LDS = 0.023 / 0.016 B/flop, i.e. 95 / 64
// B/clk/CU at the MFMA peak -- above what the L2->CU path delivers, so they top out at 20-29 % of the bf16 peak.  A 256x256
// tile needs 0.0078 B/flop (32 B/clk/CU).  8 waves = 2 (M) x 4 (N), wave tile (BM/2) x 64, 64 MFMAs per wave per k-tile.
//   * W (packed fragment order in HBM): one DMA instruction copies one 1-KB (16 rows, 32 k) fragment sub-tile; the fragment
//     read is the linear, conflict-free ds_read_b128 of slot `lane` (as in gemm_glds_kernel).
//   * A (row-major activations): staged in FULL 128-byte lines -- one DMA instruction = 8 rows x 128 B -- into a [BM][8 x 16 B]
//     image whose 16-byte chunks are XOR-swizzled with (row & 7).  LDS-DMA writes lane-linear, so the swizzle is applied to
//     the per-lane SOURCE chunk and again on the fragment read (the same involution on both sides).  Full-line reads keep the
//     texture-address path at one request per 128 B instead of the 16 rows x 64 B gather of a fragment-shaped load.
//   * pipeline: BM = 256: two 64-KB stages (128 KB), vmcnt(0) + s_barrier per k-tile, the DMA of tile kt+1 runs under the
//     MFMAs of tile kt.  BM = 128: three 48-KB stages, counted vmcnt so that one tile stays in flight across the barrier.
// Requires packed W, K % 64 == 0 (every shape of the 7B/2B/72B models except the 1176-wide patch embedding).
// W8 = true: W is fp8 e4m3 in the PACKED8 order (one 1-KB DMA = a 16-row x 64-k fragment = both k-steps of a k-tile); the
// fragment read is one ds_read_b128 per 16 rows whose halves are converted to bf16 in registers, and the activation chunk
// order follows PACKED8's k assignment (MFMA h of lane group g takes k = g*16 + h*8 ..).  Half the W bytes in L2 and LDS;
// the stage shrinks to 48 KB (BM 256) so that three stages fit and one tile stays in flight across the barrier.
// (the kernel proper is gemm_big_kernel below: it maps blockIdx to a tile (tm, tn) and runs this body -- once for every epilogue but
// EPI_VIT_QKV, whose column tiles take the q|k body or the V body)
template <int BM, int EPI, int SCHED, bool W8>
LCC_DEVICE void gemm_big_body(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int ldr,
    bf16_t* __restrict__ C, int ldc, int M, int N, int K, int tm, int tn,
    float* __restrict__ partial, int kt_per_split, const float* __restrict__ wscale, const VitQkvEpi& vq) {
  constexpr int BN = 256, BK = 64, NSTAGE = (BM >= 192 && !W8) ? 2 : 3;     // BM 192 (round 4): 56-KB stages, two of them
  constexpr int WM = BM / 2, MT = WM / 16, NT = 4;
  constexpr bool SWAP = EPI == EPI_VIT_V;         // activations as the MFMA A operand: a lane ends with 4 consecutive rows of one column
  constexpr int A_UNITS = BM * 8;                 // 16-byte units of the A image
  constexpr int B_SUB = (BN / 16) * (W8 ? 1 : 2); // 1-KB fragment sub-tiles of the W tile
  constexpr int STAGE = A_UNITS + B_SUB * 64;     // 16-byte units per stage
  constexpr int A_PER_WAVE = BM / 64;             // 8-row x 128-B pieces per wave (BM/8 pieces over 8 waves)
  constexpr int B_PER_WAVE = B_SUB / 8;
  constexpr int G = A_PER_WAVE + B_PER_WAVE;      // DMA instructions per wave per k-tile
  extern __shared__ __attribute__((aligned(16))) u32x4 dsmem[];

  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 15, g = lane >> 4;
  const int K32 = K >> 5, nfrag = N >> 4;
  const bool wave_has_rows = m0 + wm * WM < M;

  const bf16_t* asrc[A_PER_WAVE];
  const bf16_t* bsrc[B_PER_WAVE];
#pragma unroll
  for (int q = 0; q < A_PER_WAVE; ++q) {
    const int row = (wave * A_PER_WAVE + q) * 8 + (lane >> 3);
    asrc[q] = A + (size_t)min(m0 + row, M - 1) * lda + (((lane & 7) ^ (lane >> 3)) << 3);
  }
#pragma unroll
  for (int q = 0; q < B_PER_WAVE; ++q) {
    const int st = wave * B_PER_WAVE + q;
    if (W8) {                                      // sub-tile = n-fragment row st, all 64 k of the tile (1 KB of fp8)
      const int fr = min((n0 >> 4) + st, nfrag - 1);
      bsrc[q] = W + ((size_t)fr * (K >> 6)) * 512 + lane * 8;          // bf16_t units: 1024 B = 512 units per fragment
    } else {                                       // sub-tile: n-fragment row st >> 1, 32-k block st & 1
      const int fr = min((n0 >> 4) + (st >> 1), nfrag - 1);
      bsrc[q] = W + ((size_t)fr * K32 + (st & 1)) * 512 + lane * 8;
    }
  }

  auto issue = [&](int kt, int stage) {
    u32x4* sbase = dsmem + stage * STAGE;
    if (SCHED != 5) {      // (SCHED 4 / 5: timing diagnostics, only the activation / only the W half of the DMA ring)
#pragma unroll
    for (int q = 0; q < A_PER_WAVE; ++q)
      glds16((asrc[q] + kt * BK), lds_addr(sbase + (wave * A_PER_WAVE + q) * 64));
    }
    if (SCHED != 4)
#pragma unroll
    for (int q = 0; q < B_PER_WAVE; ++q)
      glds16((bsrc[q] + (size_t)kt * (W8 ? 512 : 1024)), lds_addr(sbase + A_UNITS + (wave * B_PER_WAVE + q) * 64));
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt_all = K / BK;
  const int kt0 = (EPI == EPI_PARTIAL) ? blockIdx.y * kt_per_split : 0;
  const int nkt = (EPI == EPI_PARTIAL) ? min(nkt_all, kt0 + kt_per_split) : nkt_all;
#pragma unroll
  for (int p = 0; p < NSTAGE - 1; ++p)
    if (kt0 + p < nkt) issue(kt0 + p, p);

  // fragment read offsets (16-byte units within a stage)
  int aoff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) aoff[kk] = (wm * WM + li) * 8 + ((W8 ? g * 2 + kk : kk * 4 + g) ^ (li & 7));
  const int boff = A_UNITS + (wn * NT * (W8 ? 1 : 2)) * 64 + lane;

  for (int kt = kt0; kt < nkt; ++kt) {
    // this wave's pieces of tile kt have landed; NSTAGE-2 later tiles may stay in flight across the barrier
    if (NSTAGE == 3 && kt + 1 < nkt) {
      if (G == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (G == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // tile kt complete in LDS; every wave is done reading tile kt-1
    // SCHED 6 issues the same pieces one at a time between the row tiles of MFMAs below (a burst of 8 DMA instructions right after the
    // barrier holds the wave's issue port for ~60-100 cycles each while BOTH waves of the SIMD are at the same point)
    if (SCHED != 2 && SCHED != 6 && kt + NSTAGE - 1 < nkt) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1 - kt0) % NSTAGE);   // SCHED 2: timing diagnostic, no DMA
    const u32x4* s = dsmem + ((kt - kt0) % NSTAGE) * STAGE;
    // ragged last row tile (e.g. M = 386 = 3 x 128 + 2): a wave whose WM rows are all past M skips the multiply and only keeps
    // feeding the DMA ring and the barriers (wave-uniform branch around the whole k-tile body; a finer per-16-row predicate
    // made hipcc if-convert the accumulators and spill)
    if (!wave_has_rows || (SCHED >= 3 && SCHED <= 5)) {      // SCHED 3-5: timing diagnostics, DMA ring (or half of it) + barriers only
      if (SCHED == 6 && kt + NSTAGE - 1 < nkt) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1 - kt0) % NSTAGE);   // no MFMA stream to spread them over
    } else if (SCHED == 0) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 fa[MT], fb[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (W8) { const u32x4 raw = s[boff + j * 64]; fb[j] = fp8x8_to_bf16x8(raw[2 * kk], raw[2 * kk + 1]); }
          else fb[j] = as_bf16x8(s[boff + (j * 2 + kk) * 64]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = as_bf16x8(s[aoff[kk] + i * 128]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = SWAP ? mfma16(fa[i], fb[j], acc[i][j]) : mfma16(fb[j], fa[i], acc[i][j]);
      }
    } else {
      // explicit software pipeline of the fragment reads: the 8 W fragments of the k-tile are read up front, the activation
      // fragments stream through a 3-register ring two steps (8 MFMAs = 128 cycles) ahead of their use; the issue order is
      // pinned with sched_group_barrier (DS read = 0x100, MFMA = 0x008) so that hipcc does not fall back to read-wait-use.
      bf16x8 fb[2][NT], fa[3];
      if (W8) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const u32x4 raw = s[boff + j * 64];
          fb[0][j] = fp8x8_to_bf16x8(raw[0], raw[1]);
          fb[1][j] = fp8x8_to_bf16x8(raw[2], raw[3]);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < NT; ++j) fb[kk][j] = as_bf16x8(s[boff + (j * 2 + kk) * 64]);
      }
      fa[0] = as_bf16x8(s[aoff[0]]);
      fa[1] = as_bf16x8(s[aoff[0] + 128]);
      // SCHED 6: piece t of the NEXT ring tile goes out after row-tile step t.  Unconditional (a branch would cut the scheduling
      // region): past the last tile the source is clamped and the copy lands in a stage nobody reads any more.
      const int kt_dma = min(kt + NSTAGE - 1, nkt - 1);
      u32x4* dma_base = dsmem + ((kt + NSTAGE - 1 - kt0) % NSTAGE) * STAGE;
#pragma unroll
      for (int t = 0; t < 2 * MT; ++t) {
        if (t + 2 < 2 * MT) fa[(t + 2) % 3] = as_bf16x8(s[aoff[(t + 2) / MT] + ((t + 2) % MT) * 128]);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[t % MT][j] = SWAP ? mfma16(fa[t % 3], fb[t / MT][j], acc[t % MT][j]) : mfma16(fb[t / MT][j], fa[t % 3], acc[t % MT][j]);
        if (SCHED == 6 && t < G) {
          if (t < A_PER_WAVE)
            glds16((asrc[t < A_PER_WAVE ? t : 0] + kt_dma * BK), lds_addr(dma_base + (wave * A_PER_WAVE + t) * 64));
          else
            glds16((bsrc[t >= A_PER_WAVE ? t - A_PER_WAVE : 0] + (size_t)kt_dma * (W8 ? 512 : 1024)), lds_addr(dma_base + A_UNITS + (wave * B_PER_WAVE + (t - A_PER_WAVE)) * 64));
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, (W8 ? NT : 2 * NT) + 2, 0);
#pragma unroll
      for (int t = 0; t < 2 * MT; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        if (SCHED == 6 && t < G) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);      // one VMEM (the LDS-DMA piece)
      }
    }
  }
  if (SCHED == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped copies of the last iterations
  if constexpr (EPI == EPI_VIT_QK) {
    // cos / sin rows of this block -> LDS (the fragment stages are dead): rows m0 .. m0 + BM - 1 of the [P, 40] fp32 tables are one
    // contiguous piece of BM * 10 16-byte units each; reads past the last row are clamped (those rows are never stored)
    __builtin_amdgcn_s_barrier();
    constexpr int UNITS = BM * 10;
    const long first = (long)m0 * 10, last = (long)M * 10 - 1;
    for (int t = wave; t * 64 < UNITS; t += 8) {
      const long u = min(first + t * 64 + lane, last);
      glds16((vq.cs + u * 4), lds_addr(dsmem + t * 64));
      glds16((vq.sn + u * 4), lds_addr(dsmem + UNITS + t * 64));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float* lcs = reinterpret_cast<const float*>(dsmem);
    vit_qk_epilogue<MT, NT>(acc, m0 + wm * WM, n0 + wn * 64, wm * WM, li, g, bias, C, ldc, M, N, vq.E, lcs, lcs + UNITS * 4);
  } else if constexpr (EPI == EPI_VIT_V) {
    vit_v_epilogue<MT, NT>(acc, m0 + wm * WM, n0 + wn * 64, li, g, bias, M, N, vq);
  } else tile_epilogue<EPI, MT, NT>(acc, m0 + wm * WM, n0 + wn * 64, (n0 + wn * 64) / 2, li, g, bias, residual, ldr, C, ldc, M, N, partial, wscale);
}

template <int BM, int EPI, int SCHED, bool W8>
__global__ __launch_bounds__(512) void gemm_big_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int ldr,
    bf16_t* __restrict__ C, int ldc, int M, int N, int K, int tiles_m, int tiles_n,
    float* __restrict__ partial, int kt_per_split, const float* __restrict__ wscale, VitQkvEpi vq, int raster) {
  const int nblk = tiles_m * tiles_n;
  for (int vb = blockIdx.x; vb < nblk; vb += gridDim.x) {     // one tile per block, or a persistent walk under the grid cap
  int bid = vb;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // tile id -> (tm, tn).  An XCD (one L2) runs a contiguous range of ids, ~32 of them at a time.
  //   raster 0: column-major -- the 32 share ~2.5 W panels and ALL row panels (M = 3088: 13 + 2.5 panels of 1.8 MB per 32 tiles);
  //   raster G (round 5): bands of G column tiles, row-major inside a band -- 32 consecutive ids are a (32 / G) x G super-tile
  //   (G = 8: 4 + 8 panels per 32 tiles: 0.77 x the bytes from beyond the L2, and the four blocks of a W panel run side by side).
  int tn, tm;
  if (raster > 0) {
    const int band = bid / (raster * tiles_m), rem = bid - band * raster * tiles_m;
    const int bw = min(raster, tiles_n - band * raster);       // the last band may be narrower
    tm = rem / bw;
    tn = band * raster + (rem - tm * bw);
  } else {
    tn = bid / tiles_m;
    tm = bid - tn * tiles_m;
  }
  if constexpr (EPI == EPI_VIT_QKV) {
    // one launch for q|k|v (N = 3E, 2E % 256 == 0): the column tiles of q and k run the rotation body, those of V the swapped-operand
    // body on rows 2E.. of the packed weight -- a block-uniform choice made once, outside the k-loop
    const int qk_tiles = (2 * vq.E) >> 8;
    if (tn >= qk_tiles)
      gemm_big_body<BM, EPI_VIT_V, SCHED, W8>(A, lda, W + (size_t)2 * vq.E * K, bias != nullptr ? bias + 2 * vq.E : nullptr, nullptr, 0, nullptr, 0, M,
                                                  vq.E, K, tm, tn - qk_tiles, nullptr, 0, nullptr, vq);
    else
      gemm_big_body<BM, EPI_VIT_QK, SCHED, W8>(A, lda, W, bias, nullptr, 0, C, ldc, M, 2 * vq.E, K, tm, tn, nullptr, 0, nullptr, vq);
  } else {
    gemm_big_body<BM, EPI, SCHED, W8>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, tm, tn, partial, kt_per_split, wscale, vq);
  }
  if (vb + (int)gridDim.x < nblk) __syncthreads();     // the next tile's DMA reuses the stages (and the rotation epilogue's tables)
  }
}

// ------------------------------------------------------------------------------------------------
// tiled GEMM v6 (round 5): the pipeline of gemm_big_kernel<256> over row tiles of VARIABLE height (16, 17 or 18 row fragments = 256 /
// 272 / 288 rows x 256 columns)
// ------------------------------------------------------------------------------------------------
// Why.  M is whatever the turn packs: 8 co-scheduled streaming chunks are 8 x 386 = 3088 rows = 12 x 256 + 16, so the 13th row tile of
// gemm_big_kernel<256> multiplies 16 live rows (and moves a whole W panel for them), and 13 x 148 = 1924 blocks are 7.5 rounds of the
// 256 CUs: the gate/up GEMM pays 8 rounds for 7.0 rounds of work (1.01 PF algorithmic against 1.18 PF inside full tiles).  Here the
// F = ceil(M / 16) row fragments are dealt out over floor(F / 16) row tiles, the first F % tiles of them one fragment taller: no ragged
// last tile wherever the leftover is at most two fragments per tile (3088 rows: 11 tiles of 256 + one of 272 -> 12 x 148 = 1776 blocks
// = 6.94 rounds; the first turn's 1131 rows: 4 tiles of 288 / 272 instead of 5; 9048 rows: 35 tiles instead of 36).
//   * waves: 2 (M) x 4 (N) as gemm_big_kernel; the M-wave pair splits the f fragments f/2 : f - f/2, i.e. a wave multiplies 8 or 9 row
//     fragments x 4 column fragments -- one copy of the k loop per count (the accumulator array must have a compile-time shape), same
//     barrier sequence in both.  Waves w and w + 4 share a SIMD, so a SIMD issues 4 f MFMAs per 32-k step: a 272-row tile costs 17/16 of
//     a 256-row tile, not two tiles.
//   * LDS image of a stage: [288 rows][8 x 16 B] activations (XOR-swizzled chunks, as gemm_big_kernel) + 32 W fragment sub-tiles = 69,632
//     B, two stages + a 1-KB scratch.  DMA pieces per wave and k-tile: 4 activation pieces (rows 0-255), ONE more for waves 0-3 when the
//     tile has the rows 256 + 8 w .. (else the slot copies one 16-byte piece into the scratch: uniform instruction stream), 4 W sub-tiles,
//     spread one per row-tile step behind the barrier (gemm_big_kernel's SCHED 6).
// Accumulation order per output element = gemm_big_kernel's (k-step 0, then 1, tile after tile): bit-identical outputs.
// Requires packed bf16 W, K % 64 == 0, F >= 16, F <= 18 * floor(F / 16).
// SMALL = 1 (round 5): the same kernel over row tiles of 8 or 9 fragments (128 / 144 rows) for the split-K projections of ONE streaming
// chunk (M = 386 = 25 fragments: 3 row tiles of 9 + 8 + 8 instead of gemm_big_kernel<128>'s 4, whose fourth holds 2 rows): an M-wave
// multiplies 4 or 5 row fragments, 2 activation pieces per wave + the extra slot (rows 128-143, waves 0-1), a ring of THREE 51,200-B stages
// (one more tile in flight: at 32-40 MFMAs per wave and k-tile a single stage of look-ahead left the DMA round trip exposed: 104 vs 72 us).
template <int EPI, int SMALL>
__global__ __launch_bounds__(512) void gemm_vh_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int ldr,
    bf16_t* __restrict__ C, int ldc, int M, int N, int K, int tiles_m, int tiles_n,
    float* __restrict__ partial, int kt_per_split, int raster) {
  constexpr int BN = 256, BK = 64, NT = 4;
  constexpr int MTX = SMALL ? 5 : 9, MTL = MTX - 1;          // row fragments of an M-wave: MTL or MTX
  constexpr int APW = SMALL ? 2 : 4;                         // 8-row activation pieces per wave below XBASE
  constexpr int XBASE = APW * 64;                            // 128 / 256: the rows above ride on the extra slot
  constexpr int AROWS = XBASE + (SMALL ? 16 : 32);           // 144 / 288
  constexpr int NSLOT = APW + 5;
  constexpr int A_UNITS = AROWS * 8;              // 16-byte units of the activation image
  constexpr int STAGE = A_UNITS + 32 * 64;        // + 32 W sub-tiles of 1 KB: 4352 units = 69,632 B (SMALL: 3200 units = 51,200 B)
  constexpr int NST = SMALL ? 3 : 2;              // ring depth: the small tile's k-step is too short to hide a DMA round trip behind one stage
  constexpr unsigned SCRATCH = (unsigned)NST * STAGE * 16u;  // byte offset of the 1-KB scratch behind the stages
  extern __shared__ __attribute__((aligned(16))) u32x4 dsmem[];

  const int nblk = tiles_m * tiles_n;
  for (int vb = blockIdx.x; vb < nblk; vb += gridDim.x) {     // one tile per block, or a persistent walk under the grid cap
  int bid = vb;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tn, tm;                                     // tile order: see gemm_big_kernel
  if (raster > 0) {
    const int band = bid / (raster * tiles_m), rem = bid - band * raster * tiles_m;
    const int bw = min(raster, tiles_n - band * raster);
    tm = rem / bw;
    tn = band * raster + (rem - tm * bw);
  } else {
    tn = bid / tiles_m;
    tm = bid - tn * tiles_m;
  }
  const int F = (M + 15) >> 4, fbase = F / tiles_m, frem = F - fbase * tiles_m;
  const int f = fbase + (tm < frem ? 1 : 0);                       // row fragments of this tile: 16..18 (SMALL: 8..9)
  const int m0 = (tm * fbase + min(tm, frem)) << 4, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 15, g = lane >> 4;
  const int K32 = K >> 5, nfrag = N >> 4;
  const int f0 = f >> 1;                                           // M-wave 0: f0 fragments, M-wave 1: the rest
  const int my_mt = wm ? f - f0 : f0, my_row0 = wm ? f0 * 16 : 0;
  const bool wave_has_rows = m0 + my_row0 < M;

  // DMA piece table of this wave: slots 0-3 activation rows (wave*4 + q)*8 .. +7, slot 4 rows 256 + wave*8 .. (waves 0-3, if the tile
  // has them), slots 5-8 W sub-tiles wave*4 + q (n-fragment row st >> 1, 32-k block st & 1)
  const bf16_t* psrc[NSLOT];
  unsigned pdst[NSLOT];
  const int swz = ((lane & 7) ^ (lane >> 3)) << 3;
#pragma unroll
  for (int q = 0; q < APW; ++q) {
    const int row = (wave * APW + q) * 8 + (lane >> 3);
    psrc[q] = A + (size_t)min(m0 + row, M - 1) * lda + swz;
    pdst[q] = (unsigned)(wave * APW + q) * 1024u;
  }
  {
    const bool have = wave < 4 && XBASE + wave * 8 < f * 16;       // wave-uniform
    const int row = XBASE + (wave & 3) * 8 + (lane >> 3);
    psrc[APW] = have ? A + (size_t)min(m0 + row, M - 1) * lda + swz : A;
    pdst[APW] = have ? (unsigned)(XBASE / 8 + wave) * 1024u : SCRATCH;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int st = wave * 4 + q, fr = min((n0 >> 4) + (st >> 1), nfrag - 1);
    psrc[APW + 1 + q] = W + ((size_t)fr * K32 + (st & 1)) * 512 + lane * 8;
    pdst[APW + 1 + q] = (unsigned)(A_UNITS + st * 64) * 16u;
  }
  const unsigned lds0 = lds_addr(dsmem);
  auto piece = [&](int q, int kt, unsigned stage_lds) {             // q <= APW: activations advance 64 elements per k-tile, W 1024
    glds16(psrc[q] + (size_t)kt * (q <= APW ? BK : 1024), pdst[q] == SCRATCH ? lds0 + SCRATCH : stage_lds + pdst[q]);
  };

  const int nkt_all = K / BK;
  const int kt0 = (EPI == EPI_PARTIAL) ? blockIdx.y * kt_per_split : 0;
  const int nkt = (EPI == EPI_PARTIAL) ? min(nkt_all, kt0 + kt_per_split) : nkt_all;
  if (kt0 < nkt) {      // tiles kt0 .. kt0 + NST - 2 (clamped) into stages 0 .. NST - 2
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
#pragma unroll
      for (int q = 0; q < NSLOT; ++q) piece(q, min(kt0 + p, nkt - 1), lds0 + (unsigned)(p * STAGE) * 16u);
  }
  int aoff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) aoff[kk] = (my_row0 + li) * 8 + ((kk * 4 + g) ^ (li & 7));
  const int boff = A_UNITS + (wn * NT * 2) * 64 + lane;

  auto run = [&](auto mt_c) {
    constexpr int MTW = decltype(mt_c)::value;
    f32x4 acc[MTW > 0 ? MTW : 1][NT];
#pragma unroll
    for (int i = 0; i < (MTW > 0 ? MTW : 1); ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int st_cur = 0;                   // stage of tile kt = (kt - kt0) % NST
    for (int kt = kt0; kt < nkt; ++kt) {
      // every wave issues exactly NSLOT pieces per tile, so with NST = 3 the newest tile's pieces may stay in flight across the barrier
      static_assert(NST == 2 || NSLOT == 7, "the counted wait below is written for 7 pieces per tile");
      if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // tile kt complete in LDS; every wave is done reading tile kt-1
      const u32x4* s = dsmem + st_cur * STAGE;
      // the pieces of tile kt+NST-1 go out unconditionally (a branch would cut the scheduling region): past the last tile the source is
      // clamped and the copy lands in a stage nobody reads any more (the stage of tile kt-1)
      const int kt_dma = min(kt + NST - 1, nkt - 1);
      const int st_dma = st_cur == 0 ? NST - 1 : st_cur - 1;
      const unsigned dma_lds = lds0 + (unsigned)(st_dma * STAGE) * 16u;
      st_cur = st_cur == NST - 1 ? 0 : st_cur + 1;
      if constexpr (MTW == 0) {
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) piece(q, kt_dma, dma_lds);
      } else {
        // pinned software pipeline of the fragment reads (gemm_big_kernel SCHED 6): the 8 W fragments of the k-tile up front, the
        // activation fragments through a 3-register ring two steps ahead, one DMA piece behind each of the first 9 row-tile steps
        bf16x8 fb[2][NT], fa[3];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < NT; ++j) fb[kk][j] = as_bf16x8(s[boff + (j * 2 + kk) * 64]);
        fa[0] = as_bf16x8(s[aoff[0]]);
        fa[1] = as_bf16x8(s[aoff[0] + 128]);
#pragma unroll
        for (int t = 0; t < 2 * MTW; ++t) {
          if (t + 2 < 2 * MTW) fa[(t + 2) % 3] = as_bf16x8(s[aoff[(t + 2) / MTW] + ((t + 2) % MTW) * 128]);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[t % MTW][j] = mfma16(fb[t / MTW][j], fa[t % 3], acc[t % MTW][j]);
          if (t < NSLOT) piece(t < NSLOT ? t : 0, kt_dma, dma_lds);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT + 2, 0);
#pragma unroll
        for (int t = 0; t < 2 * MTW; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped copies of the last iteration
    if constexpr (MTW > 0)
      tile_epilogue<EPI, MTW, NT>(acc, m0 + my_row0, n0 + wn * 64, (n0 + wn * 64) / 2, li, g, bias, residual, ldr, C, ldc, M, N, partial, nullptr);
  };
  using I0 = std::integral_constant<int, 0>;
  using IL = std::integral_constant<int, MTL>;
  using IX = std::integral_constant<int, MTX>;
  static_assert(2 * MTL >= NSLOT, "one DMA piece per row-tile step");
  if (!wave_has_rows) run(I0{});
  else if (my_mt == MTX) run(IX{});
  else run(IL{});
  if (vb + (int)gridDim.x < nblk) __syncthreads();     // the next tile's DMA reuses the stages
  }
}

// ------------------------------------------------------------------------------------------------
// tiled GEMM v4 ("tall"): ONE block row covers all of M (256 < M <= 448: a single-stream streaming chunk is 386 rows), 8 waves,
// 448 x 160 x 64 block tile, LDS-DMA for both operands
// ------------------------------------------------------------------------------------------------
// Why: M = 386 on the 128 x 256 tiles of gemm_big_kernel is 4 row tiles (the fourth holds 2 rows) x 148 = 592 blocks = 2.3 rounds
// of the 256 CUs at one block per CU: the gate/up GEMM of a chunk ran at 0.65 PF / 29.8 % MfmaUtil against 50.7 % at M = 3088
// (profiles/r02).  Split-K or stream-K on those tiles buys <= 13 % (1.75 instead of 2 rounds).  Here every block owns ALL rows and
// 160 columns, so N = 37888 is 237 blocks = ONE round on 92.5 % of the CUs, the W panel of a block is read once from L2, and the
// activation tile (448 x 64 per k-step, the same for every block) comes out of L2.
//   * waves: 4 (M) x 2 (N).  M-wave wm owns rows wm*112 .. +111 = 7 MFMA row tiles; N-wave 0 owns column tiles 0-5, N-wave 1
//     tiles 6-9 (gate/up pairs stay inside a wave: SwiGLU epilogue).  Waves w and w + 4 share a SIMD, so every SIMD issues
//     7 x (6 + 4) x 2 = 140 MFMAs per k-step.  A wave whose rows end early (M = 386: the last M-wave has 4 live row tiles)
//     takes the 4-row-tile code path; a wave without rows only feeds the DMA ring.
//   * LDS: A image 448 rows x 128 B (XOR-swizzled 16-byte chunks, as gemm_big_kernel) + W tile 20 fragment sub-tiles of 1 KB =
//     77,824 B per stage, two stages = 155,648 B; per k-step 76 KB of DMA (1216 clk at 64 B/clk) under 2240 clk of MFMA issue.
// Requires packed bf16 W, K % 64 == 0.
template <int EPI, int SCHED>
__global__ __launch_bounds__(512) void gemm_tall_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
    const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int ldr,
    bf16_t* __restrict__ C, int ldc, int M, int N, int K) {
  constexpr int BM = 448, BN = 160, BK = 64, WM = 112, MT = 7, NT0 = 6;
  constexpr int A_UNITS = BM * 8;                 // 16-byte units of the A image
  constexpr int B_SUB = (BN / 16) * 2;            // 20 fragment sub-tiles of 1 KB
  constexpr int STAGE = A_UNITS + B_SUB * 64;
  constexpr int A_PER_WAVE = BM / 64;             // 7 pieces of 8 rows x 128 B per wave
  extern __shared__ __attribute__((aligned(16))) u32x4 dsmem[];

  const int n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int li = lane & 15, g = lane >> 4;
  const int K32 = K >> 5, nfrag = N >> 4;
  const int rows_left = M - wm * WM;                                     // wave-uniform
  const int mt_live = rows_left <= 0 ? 0 : min(MT, (rows_left + 15) >> 4);

  int aoffs[A_PER_WAVE];                         // element offsets from A (the row is clamped into the matrix)
#pragma unroll
  for (int q = 0; q < A_PER_WAVE; ++q) {
    const int row = (wave * A_PER_WAVE + q) * 8 + (lane >> 3);
    aoffs[q] = min(row, M - 1) * lda + (((lane & 7) ^ (lane >> 3)) << 3);
  }
  // W sub-tiles: waves 0-3 copy three each (0..11), waves 4-7 two each (12..19); sub-tile st = fragment row st >> 1, 32-k block st & 1
  const int nb = wave < 4 ? 3 : 2;
  const int st0 = wave < 4 ? wave * 3 : 12 + (wave - 4) * 2;
  const bf16_t* bsrc[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int st = min(st0 + q, B_SUB - 1);
    const int fr = min((n0 >> 4) + (st >> 1), nfrag - 1);
    bsrc[q] = W + ((size_t)fr * K32 + (st & 1)) * 512 + lane * 8;
  }
  // Round 6: activation pieces (8 rows each) that lie entirely beyond the last 16-row MFMA tile are not fetched at all -- at M = 386 the rows
  // 400..447 of the 448-row image (6 of 56 pieces: 11 % of the activation DMA, which is 70 % of this kernel's L2 -> LDS traffic) were copies
  // of row 385 that no MFMA reads.  Wave-uniform; the ring's waits are vmcnt(0), so the piece count per wave need not be uniform.
  const int m_rows = (M + 15) & ~15;
  auto issue = [&](int kt, int stage) {
    u32x4* sbase = dsmem + stage * STAGE;
#pragma unroll
    for (int q = 0; q < A_PER_WAVE; ++q)
      if ((wave * A_PER_WAVE + q) * 8 < m_rows)
        glds16((A + aoffs[q] + kt * BK), lds_addr(sbase + (wave * A_PER_WAVE + q) * 64));
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (q < nb)
        glds16((bsrc[q] + (size_t)kt * 1024), lds_addr(sbase + A_UNITS + (st0 + q) * 64));
  };

  // SCHED 2: the 10 pieces of a wave (7 activation + 3 W; waves 4-7 repeat their second W sub-tile as the third) go out ONE at a time
  // between the row-tile steps of the MFMA stream instead of as a burst behind the barrier (every LDS-DMA instruction holds the wave's
  // issue port for ~60-100 cycles, and right after the barrier both waves of a SIMD are at the same point)
  auto issue_piece = [&](int kt, int stage, int q) {
    u32x4* sbase = dsmem + stage * STAGE;
    if (q < A_PER_WAVE) {
      if ((wave * A_PER_WAVE + q) * 8 < m_rows)
        glds16((A + aoffs[q < A_PER_WAVE ? q : 0] + kt * BK), lds_addr(sbase + (wave * A_PER_WAVE + q) * 64));
    } else {
      const int qq = min(q - A_PER_WAVE, nb - 1);
      glds16((bsrc[qq] + (size_t)kt * 1024), lds_addr(sbase + A_UNITS + (st0 + qq) * 64));
    }
  };
  constexpr int NPIECE = A_PER_WAVE + 3;

  f32x4 acc[MT][NT0];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT0; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = K / BK;
  issue(0, 0);
  int aoff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) aoff[kk] = (wm * WM + li) * 8 + ((kk * 4 + g) ^ (li & 7));
  const int boff = A_UNITS + (wn * NT0 * 2) * 64 + lane;

  // one k-tile of this wave: MTW live row tiles x NTW column tiles.  Per 32-k half: the NTW W fragments are read up front, the
  // activation fragments stream through a 3-register ring two row tiles ahead of their MFMAs.
  auto ktile = [&](const u32x4* s, auto mtw_c, auto ntw_c, int kt_dma, int st_dma) {
    constexpr int MTW = decltype(mtw_c)::value, NTW = decltype(ntw_c)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fb[NTW], fa[3];
#pragma unroll
      for (int j = 0; j < NTW; ++j) fb[j] = as_bf16x8(s[boff + (j * 2 + kk) * 64]);
      fa[0] = as_bf16x8(s[aoff[kk]]);
      if (MTW > 1) fa[1] = as_bf16x8(s[aoff[kk] + 128]);
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        if (i + 2 < MTW) fa[(i + 2) % 3] = as_bf16x8(s[aoff[kk] + (i + 2) * 128]);
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = mfma16(fb[j], fa[i % 3], acc[i][j]);
        if (SCHED == 2 && kk * MTW + i < NPIECE) issue_piece(kt_dma, st_dma, kk * MTW + i);
      }
      if (SCHED >= 1) {   // pin the issue order: the fragment reads up front, then one activation read per row tile of MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, NTW + (MTW > 1 ? 2 : 1), 0);
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          if (i + 2 < MTW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);
          if (SCHED == 2 && kk * MTW + i < NPIECE) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      }
    }
    if (SCHED == 2) {     // fewer row-tile steps than pieces (the 4-row-tile wave: 8 steps): the rest behind the last MFMAs
#pragma unroll
      for (int q = 2 * MTW; q < NPIECE; ++q) issue_piece(kt_dma, st_dma, q);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I4 = std::integral_constant<int, 4>;
  using I6 = std::integral_constant<int, 6>;
  using I7 = std::integral_constant<int, 7>;
  const int nbase = n0 + wn * NT0 * 16;
  // The live tile shape of a wave is loop-invariant, so every shape gets its OWN copy of the k loop (same barrier sequence in all
  // of them): one loop with a switch inside made hipcc keep the accumulators of all shapes alive across the paths and spill 190
  // registers; a single-shape loop needs 252.
  auto run = [&](auto mtw_c, auto ntw_c) {
    constexpr int MTW = decltype(mtw_c)::value, NTW = decltype(ntw_c)::value;
    for (int kt = 0; kt < nkt; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // tile kt complete in LDS; every wave is done reading tile kt-1
      // SCHED 2 (waves with rows): the pieces go out inside ktile, unconditionally -- past the last tile the source is clamped and the
      // copy lands in the stage nobody reads any more
      if ((SCHED != 2 || MTW == 0) && kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
      if constexpr (MTW > 0) ktile(dsmem + (kt & 1) * STAGE, mtw_c, ntw_c, min(kt + 1, nkt - 1), (kt + 1) & 1);
    }
    if (SCHED == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MTW > 0)
      tile_epilogue<EPI, (MTW > 0 ? MTW : 1), (MTW > 0 ? NTW : 2), MT, NT0>(acc, wm * WM, nbase, nbase / 2, li, g, bias, residual, ldr, C, ldc, M, N,
                                                                        nullptr, nullptr);
  };
  if (mt_live == 0) run(I0{}, I4{});                 // no rows: DMA + barriers only
  else if (wn == 0) { if (mt_live <= 4) run(I4{}, I6{}); else run(I7{}, I6{}); }
  else { if (mt_live <= 4) run(I4{}, I4{}); else run(I7{}, I4{}); }
}

// 0: register-staged 2-stage kernel; 1: LDS-DMA 3-stage kernel; 2 (default): measured best per tile shape --
// 64-row tiles (72 KB ring, 2 blocks/CU) take the LDS-DMA kernel (1.5-1.6x), 128-row tiles keep the register-staged
// kernel (64 KB, 2 blocks/CU; the 96 KB ring would leave 1 block/CU and measured 0.75x).
static int g_gemm_variant = 2;
static int g_gemm_sched = 1;   // fragment-read schedule of gemm_big_kernel: 0 compiler order, 1 pinned software pipeline
void set_gemm_variant(int v);

template <int BM, int EPI>
static void launch_tiled(const GemmArgs& a, hipStream_t st) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + 127) / 128;
  const int nkt = (a.K + 63) / 64, S = (EPI == EPI_PARTIAL) ? a.nsplit : 1;
  if (g_gemm_variant == 1 || ((g_gemm_variant == 2 || g_gemm_variant == 7) && BM == 64)) {   // 7: auto without the 8-wave kernel
    constexpr size_t lds = (size_t)3 * ((BM / 16) * 2 + 16) * 1024;   // 96 KB (BM 128) / 72 KB (BM 64)
    static DeviceOnce attr_set;   // per instantiation
    if (attr_set.first()) {
      (void)hipFuncSetAttribute((const void*)gemm_glds_kernel<BM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    gemm_glds_kernel<BM, EPI><<<dim3(capped_grid((long)tiles_m * tiles_n), S), dim3(256), lds, st>>>(
        a.A, a.lda, a.W, a.ldw, a.bias, a.residual, a.ldr, a.C, a.ldc, a.M, a.N, a.K, tiles_m, tiles_n, a.w_packed, a.partial,
        (nkt + S - 1) / S, a.wscale);
    return;
  }
  gemm_tiled_kernel<BM, EPI><<<dim3(tiles_m * tiles_n, S), dim3(256), 0, st>>>(
      a.A, a.lda, a.W, a.ldw, a.bias, a.residual, a.ldr, a.C, a.ldc, a.M, a.N, a.K, tiles_m, tiles_n, a.w_packed, a.partial,
      (nkt + S - 1) / S, a.wscale);
}

// LCC_GEMM_RASTER (A/B, read once): band width of the tile order of the 8-wave kernels (bands of G column tiles walked row-major: the ~32
// tiles an XCD runs at a time form a (32 / G) x G super-tile), 0 = column-major (rounds 2-4)
static int g_gemm_raster = [] { const char* v = getenv("LCC_GEMM_RASTER"); return v ? atoi(v) : -1; }();
// -1 (default): bands of 4 column tiles once there are more than 6 row tiles; with few row tiles column-major already puts every row
// tile of a W panel side by side (M = 1131: 335 us column-major vs 342-355 us in bands, profiles/r05/gemm_vh_raster_ab.jsonl)
static inline int raster_for(int tiles_m) { return g_gemm_raster >= 0 ? g_gemm_raster : (tiles_m > 6 ? 4 : 0); }
template <int BM, int EPI, int SCHED, bool W8>
static void launch_big_s(const GemmArgs& a, hipStream_t st) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + 255) / 256;
  const int nkt = a.K / 64, S = (EPI == EPI_PARTIAL) ? a.nsplit : 1;
  // bf16: 128 KB (BM 256, 2 stages) / 144 KB (BM 128, 3 stages); fp8 W: 144 KB (BM 256) / 96 KB (BM 128), 3 stages
  constexpr size_t lds = (size_t)((BM >= 192 && !W8) ? 2 : 3) * (BM * 8 + (W8 ? 1024 : 2048)) * 16;
  static DeviceOnce attr_set;   // per instantiation
  if (attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<BM, EPI, SCHED, W8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  gemm_big_kernel<BM, EPI, SCHED, W8><<<dim3(capped_grid((long)tiles_m * tiles_n), S), dim3(512), lds, st>>>(
      a.A, a.lda, a.W, a.bias, a.residual, a.ldr, a.C, a.ldc, a.M, a.N, a.K, tiles_m, tiles_n, a.partial, (nkt + S - 1) / S, a.wscale, a.vq,
      raster_for(tiles_m));
}
// variable-height row tiles (gemm_vh_kernel): tiles_m = floor(F / 16) row tiles for F = ceil(M / 16) row fragments (small class: F / 8)
static bool vh_legal(const GemmArgs& a, bool small = false) {
  const int F = (a.M + 15) >> 4, t = F >> (small ? 3 : 4);
  // wscale == nullptr: gemm_w8's dequantisation fallback re-enters with w_fp8 = 0 and a LIVE row scale for the epilogue, which
  // this kernel does not carry
  return a.w_packed && !a.w_fp8 && a.wscale == nullptr && (a.K % 64) == 0 && t >= 1 && F <= (small ? 9 : 18) * t && (a.N & 15) == 0;
}
template <int EPI, int SMALL = 0>
static void launch_vh(const GemmArgs& a, hipStream_t st) {
  const int F = (a.M + 15) >> 4, tiles_m = F >> (SMALL ? 3 : 4), tiles_n = (a.N + 255) / 256;
  const int nkt = a.K / 64, S = (EPI == EPI_PARTIAL) ? a.nsplit : 1;
  constexpr size_t lds = (size_t)(SMALL ? 3 : 2) * ((SMALL ? 144 : 288) * 8 + 2048) * 16 + 1024;      // 2 x 69,632 B / 3 x 51,200 B + the scratch
  static DeviceOnce attr_set;   // per instantiation
  if (attr_set.first()) (void)hipFuncSetAttribute((const void*)gemm_vh_kernel<EPI, SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  g_launch_counts[LC_GEMM_VH]++;
  gemm_vh_kernel<EPI, SMALL><<<dim3(capped_grid((long)tiles_m * tiles_n), S), dim3(512), lds, st>>>(
      a.A, a.lda, a.W, a.bias, a.residual, a.ldr, a.C, a.ldc, a.M, a.N, a.K, tiles_m, tiles_n, a.partial, (nkt + S - 1) / S, raster_for(tiles_m));
}
template <int BM, int EPI>
static void launch_big(const GemmArgs& a, hipStream_t st) {
  if constexpr (BM == 256 && EPI == EPI_SWIGLU) {
    // LCC_GEMM_DIAG (tools/bench_gemm_diag.py; results are WRONG by construction): 2 = no DMA after the prologue, 3 = no MFMAs
    static const int diag = [] { const char* v = getenv("LCC_GEMM_DIAG"); return v ? atoi(v) : 0; }();
    if (diag == 2 && !a.w_fp8) return launch_big_s<BM, EPI, 2, false>(a, st);
    if (diag == 3 && !a.w_fp8) return launch_big_s<BM, EPI, 3, false>(a, st);
    if (diag == 4 && !a.w_fp8) return launch_big_s<BM, EPI, 4, false>(a, st);
    if (diag == 5 && !a.w_fp8) return launch_big_s<BM, EPI, 5, false>(a, st);
  }
  // pinned fragment-read pipeline with the LDS-DMA pieces spread between the row-tile steps (SCHED 6, default since round 3:
  // bit-identical to the burst-behind-the-barrier order SCHED 1 -- tools/gemm_checksum.py -- and 1.6-3.3 % faster on the prefill
  // GEMMs, +0.6 / +0.8 % tokens/s at 1 / 8 streams, profiles/r03/gemm_sched_spread_dma.txt); LCC_GEMM_SCHED=1 restores SCHED 1
  static const int spread = [] { const char* v = getenv("LCC_GEMM_SCHED"); return (v && atoi(v) == 1) ? 0 : 1; }();
  if (a.w_fp8) launch_big_s<(BM == 192 ? 256 : BM), EPI, 1, true>(a, st);     // the 192-row tile is a bf16-weight shape (big_tile_rows never picks it for fp8)
  else if (g_gemm_sched && spread) launch_big_s<BM, EPI, 6, false>(a, st);
  else if (g_gemm_sched) launch_big_s<BM, EPI, 1, false>(a, st);
  else launch_big_s<BM, EPI, 0, false>(a, st);
}
// 13: the 192-row tile wherever eligible (bf16 weights); 14: variable-height row tiles (gemm_vh_kernel) wherever legal, else as 3;
// 15: the small variable-height class (8 / 9 fragments) for split-K slabs wherever legal, else as 4
void set_gemm_variant(int v) {
  g_gemm_variant = v;
  g_gemm_sched = (v == 5 || v == 6) ? 0 : 1;
}
static bool big_eligible(const GemmArgs& a) { return a.w_packed && (a.K % 64) == 0 && a.M > 16; }
static_assert(GEMM_EPI_VIT_QK == (int)EPI_VIT_QK && GEMM_EPI_VIT_V == (int)EPI_VIT_V && GEMM_EPI_VIT_QKV == (int)EPI_VIT_QKV, "kernels.h / common.h disagree");
// M > 64: up to 64 rows the plain projection runs on the weight-streaming kernels (another fp32 summation order, and the better shape for so
// few rows) -- the fused form takes over where the plain one is an MFMA tile kernel, so both forms give the same bits
bool gemm_vit_qkv_eligible(int M, int E, int K) { return K > 0 && (K % 64) == 0 && M > 64 && (M & 3) == 0 && E > 0 && (E & 31) == 0; }

// 0 = not the 8-wave kernel, else its BM.  Variants 3 / 4 (5 / 6) force BM 256 / 128 wherever the kernel is eligible.
// Auto (2): estimated relative throughput = row utilisation x wave quantisation x measured tile efficiency.  The 8-wave
// kernels run one block per CU (256 slots), the 4-wave 64-row kernel two (512 slots).  Measured on MI355X (7B shapes):
// 256x256 tiles ~1.05-1.1 PF, 128x256 ~0.9 PF, 64x128 ~0.55 PF when the grid fills the chip; with few blocks (qkv at
// M = 386: 72 blocks of 128x256) the small tile wins.
// small variable-height class: on unless LCC_GEMM_VH_SMALL=0; shapes = one block-row class of a streaming chunk (fragment count not a
// multiple of 8, at most 448 rows) -- the engine asks gemm_tiled_num_splits with the same predicate, so split count and tile agree
static bool vh_small_on() { static const int on = [] { const char* v = getenv("LCC_GEMM_VH_SMALL"); return v ? atoi(v) : 1; }(); return on != 0; }
static bool vh_small_shape(int M, int K) {
  const int F = (M + 15) >> 4, t = F >> 3;
  return M > 64 && M <= 448 && (K % 64) == 0 && t >= 1 && F <= 9 * t && (F & 7) != 0;
}
static float tile_score(int M, int N, int S, int BM, int BN, int slots, float eff, bool ragged_skip) {
  const long tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN, blocks = tm * tn * S;
  const int half = BM / 2, mh = (M + half - 1) / half * half;   // the 8-wave kernel skips an all-padding half tile (~25 % cost)
  const float rows = ragged_skip ? (float)mh + 0.25f * (float)(tm * BM - mh) : (float)(tm * BM);
  const float util = (float)M / rows;
  const float quant = (float)blocks / (float)((blocks + slots - 1) / slots * slots);
  return util * quant * eff;
}
// BM 192 (round 4, bf16 weights): a 256-row tile leaves the chip partly idle whenever (M / 256) x (N / 256) falls just short of a round --
// the o / down projections of 8 co-scheduled chunks are M = 3088 x N = 3584 = 13 x 14 = 182 blocks on 256 CUs (29 % idle); 17 x 14 = 238
// blocks of 192 rows fill 93 % of one round at 3/4 of the tile time.  The tile moves 17 % more L2 -> LDS bytes per flop (efficiency 0.93).
// Variable-height tiles (round 5; returned as 272): floor(F / 16) row tiles of 16-18 fragments -- scored with every tile as tall as the
// tallest (a 17-fragment tile costs 17/16 of a 256-row tile on every SIMD), no ragged tile, the 256-row tile's efficiency.
// LCC_GEMM_VH=0: rounds 2-4 tile choice.
static int big_tile_rows(const GemmArgs& a, int S) {
  if (!big_eligible(a)) return 0;
  if (g_gemm_variant == 14) return vh_legal(a) ? 272 : 256;
  if (g_gemm_variant == 3 || g_gemm_variant == 5) return 256;
  if (g_gemm_variant == 4 || g_gemm_variant == 6) return 128;
  if (g_gemm_variant == 13) return a.w_fp8 ? 256 : 192;
  if (g_gemm_variant == 15) return (a.partial != nullptr && vh_small_shape(a.M, a.K) && vh_legal(a, true)) ? 144 : 128;
  if (g_gemm_variant != 2) return 0;
  const float s256 = tile_score(a.M, a.N, S, 256, 256, 256, 1.0f, true);
  static const int allow192 = [] { const char* v = getenv("LCC_GEMM_192"); return v ? atoi(v) : 1; }();     // A/B: 0 = round-3 tile choice
  const float s192 = (a.w_fp8 || !allow192) ? 0.f : tile_score(a.M, a.N, S, 192, 256, 256, 0.93f, true);
  const float s128 = tile_score(a.M, a.N, S, 128, 256, 256, 0.85f, true);
  const float s64 = tile_score(a.M, a.N, S, 64, 128, 512, 0.55f, false);
  static const int allow_vh = [] { const char* v = getenv("LCC_GEMM_VH"); return v ? atoi(v) : 1; }();
  if (allow_vh && vh_legal(a) && (((a.M + 15) >> 4) & 15) != 0) {      // F % 16 == 0: the 256-row tiles are the same thing
    const int F = (a.M + 15) >> 4, t = F >> 4, maxf = (F + t - 1) / t;
    const long blocks = (long)t * ((a.N + 255) / 256) * S;
    const float svh = (float)a.M / (float)(t * maxf * 16) * (float)blocks / (float)((blocks + 255) / 256 * 256);
    if (svh > s256 && svh > s192 && svh > s128 && svh > s64) return 272;
  }
  // small class (returned as 144): split-K slabs of one streaming chunk's projections, where the 128-row tiles leave a nearly empty last row tile
  if (vh_small_on() && a.partial != nullptr && vh_small_shape(a.M, a.K) && vh_legal(a, true)) {
    const int F = (a.M + 15) >> 4, t = F >> 3, maxf = (F + t - 1) / t;
    const long blocks = (long)t * ((a.N + 255) / 256) * S;
    const float svs = 0.85f * (float)a.M / (float)(t * maxf * 16) * (float)blocks / (float)((blocks + 255) / 256 * 256);
    if (svs > s256 && svs > s192 && svs > s128 && svs > s64) return 144;
  }
  if (s256 >= s192 && s256 >= s128 && s256 >= s64) return 256;
  if (s192 >= s128 && s192 >= s64) return 192;
  if (s128 >= s64) return 128;
  return 0;
}

// The tall kernel serves one-block-row problems whose column tiles fill most of ONE round of the chip (7B gate/up at a streaming
// chunk: 237 blocks).  Variant 8 forces it wherever it is legal (tests).
static bool tall_legal(const GemmArgs& a) { return big_eligible(a) && !a.w_fp8 && a.M <= 448 && (a.N & 15) == 0; }
static bool tall_wanted(const GemmArgs& a) {
  if (!tall_legal(a)) return false;
  if (g_gemm_variant == 8) return true;
  if (g_gemm_variant != 2) return false;
  const int blocks = (a.N + 159) / 160;
  return a.M > 256 && blocks >= 192 && blocks <= 256;
}
// (a 4-stage ring of 32-k half tiles in the same LDS was measured equal-to-slower -- 122.5 vs 125.2 us at M = 386, profiles/r03/gemm_tall_ring.txt:
// the L2 -> LDS DMA is throughput-bound, a deeper ring has nothing to hide -- and was retired in round 5)
template <int EPI>
static void launch_tall(const GemmArgs& a, hipStream_t st) {
  g_launch_counts[LC_GEMM_TALL]++;
  constexpr size_t lds = (size_t)2 * (448 * 8 + 20 * 64) * 16;   // 155,648 B
  static DeviceOnce attr_set;   // per instantiation
  if (attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)gemm_tall_kernel<EPI, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_tall_kernel<EPI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_tall_kernel<EPI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  // 0 compiler order, 1 pinned (default since round 6: 127.6 vs 131.6-132.6 us at M = 386 on N(0,1) activations, A/B/A/B, bit-identical --
  // profiles/r06/m386_gemms_tall_sched_ab.jsonl; before the LDS-DMA went into inline asm the compiler's order had been the faster one), 2 pinned + spread DMA
  static const int sched = [] { const char* v = getenv("LCC_TALL_SCHED"); return v ? atoi(v) : 1; }();
  if (sched == 2)
    gemm_tall_kernel<EPI, 2><<<dim3((a.N + 159) / 160), dim3(512), lds, st>>>(a.A, a.lda, a.W, a.bias, a.residual, a.ldr, a.C, a.ldc, a.M, a.N, a.K);
  else if (sched == 1)
    gemm_tall_kernel<EPI, 1><<<dim3((a.N + 159) / 160), dim3(512), lds, st>>>(a.A, a.lda, a.W, a.bias, a.residual, a.ldr, a.C, a.ldc, a.M, a.N, a.K);
  else
    gemm_tall_kernel<EPI, 0><<<dim3((a.N + 159) / 160), dim3(512), lds, st>>>(a.A, a.lda, a.W, a.bias, a.residual, a.ldr, a.C, a.ldc, a.M, a.N, a.K);
}

template <int EPI>
static void launch_tiled_bm(const GemmArgs& a, hipStream_t st) {
  if constexpr (EPI != EPI_PARTIAL) {       // split-K slabs stay on the 128/256-row tiles (the tall kernel has no K split)
    if (tall_wanted(a)) return launch_tall<EPI>(a, st);
  }
  const int big = big_tile_rows(a, 1);
  if (big == 272) return launch_vh<EPI>(a, st);
  if (big == 256) return launch_big<256, EPI>(a, st);
  if (big == 192) return launch_big<192, EPI>(a, st);
  if (big == 128) return launch_big<128, EPI>(a, st);
  // 128-row tiles only for large M with a grid that fills the chip; otherwise 64-row tiles (less waste on a ragged M such as
  // 386 rows, more blocks, and the faster LDS-DMA kernel)
  const long blocks128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (a.M >= 1024 && blocks128 >= 512) launch_tiled<128, EPI>(a, st);
  else launch_tiled<64, EPI>(a, st);
}

// Which kernel family serves a packed-weight GEMM of this shape under the current variant (host logic only: no launch, no GPU needed):
// 16 = weight-streaming GEMV, 448 = tall, 272 / 144 = variable-height tiles (big / small class), 256 / 192 / 128 = the 8-wave tile of
// that height, 64 = a 4-wave tile kernel.  `nsplit` > 0: the split-K slab form with that many splits (as the engine asks: gemm_tiled_num_splits).
int gemm_plan(int M, int N, int K, int epilogue, int nsplit, bool w_fp8) {
  GemmArgs a; a.M = M; a.N = N; a.K = K; a.w_packed = 1; a.w_fp8 = w_fp8 ? 1 : 0; a.epilogue = epilogue; a.nsplit = nsplit;
  static float dummy;
  if (nsplit > 0) a.partial = &dummy;      // never dereferenced here: only "is this the slab form?"
  const bool skinny = gemm_routes_skinny(M, K, w_fp8) && epilogue != EPI_QUICK_GELU && epilogue != EPI_GELU_ERF && epilogue != EPI_RESIDUAL;
  if (skinny) return 16;
  if (nsplit <= 0 && !w_fp8 && tall_wanted(a)) return 448;
  const int big = big_tile_rows(a, nsplit > 0 ? nsplit : 1);
  return big != 0 ? big : 64;
}

// ------------------------------------------------------------------------------------------------
// skinny GEMV-like kernel (M <= 16; 17..64 rows with MG = 2..4 activation fragments per weight fragment): HBM-bound weight streaming
// ------------------------------------------------------------------------------------------------
// Block = 4 waves.  The block owns NTILE*16 consecutive W rows and the 64-element K chunks [c_begin, c_end) of split
// blockIdx.y; wave w takes chunks c_begin + w, + 4, ...  W fragments go HBM -> VGPR -> MFMA (no LDS: each byte is used
// once); per chunk and tile a lane issues two 16-byte loads plus the matching x fragment (L2/L1 resident).  The four
// partial accumulators are reduced through LDS and wave 0 runs the epilogue.  D'[n][m]: lane (m = l&15, g) ends with 4
// consecutive n for activation row m.
//
// W layouts:  row-major [N,K] (a wave load = 16 rows x 64 B), or PACKED [N/16][K/32][4 g][16 rows][8 k]: the 1-KB
// A-operand fragment of (row tile, 32-k block) is contiguous in lane order, so every wave load is one contiguous KB and a
// block streams one contiguous 16*K*2-byte region per tile -- linear DRAM streams instead of 64-byte row segments.
//
// PIPE 0: UNR chunks are loaded, then multiplied (latency hidden by ~20 resident waves per CU);
// PIPE 1: two-stage software pipeline (the next stage's loads are in flight while this one is multiplied).
__device__ unsigned int lcc_zero_page_g[64];  // 256 zero bytes: x operand of absent K chunks (address select, no branch)

// MG = 16-row groups of activations (M <= 16 * MG): the weight fragment of a k-block is loaded ONCE and multiplied with MG activation
// fragments (rows mg*16 + li), so 17..64 decode streams still read every weight byte once per step (round 4: before, they went through the
// 64-row GEMM tiles: 8.5-11.9 ms per 32-stream step where the weight stream alone costs ~2.6 ms).
template <int NTILE, int MODE, bool PACKED, int UNR, int PIPE, int MG = 1>
__global__ __launch_bounds__(256) void gemv_skinny_kernel(
    const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ W, int ldw,
    const bf16_t* __restrict__ bias, void* __restrict__ out, int ldo, int M, int N, int K, int chunks_per_split, GemvTail tail) {
  constexpr int NW = 4;
  static_assert(MODE != 3 || MG == 1, "the fused tail serves M <= 2");
  __shared__ f32x4 red[NW - 1][NTILE * MG][64];
  // MG > 1: the activation rows of a 64-k chunk are fetched in FULL 128-byte lines (one load instruction = 8 rows x 128 B; the
  // fragment-shaped load -- 16 rows x 64 B per instruction -- was 2.3x slower at M = 32: 99 vs 43 us for the gate/up GEMV,
  // profiles/r04/bench_32streams_mg_gemv_first_version.json) and turned into MFMA fragments through a wave-private LDS scratch with the
  // (chunk ^ row & 7) swizzle of gemm_big_kernel's activation image: ds_write_b128 of the raw lines, ds_read_b128 of the fragments.
  __shared__ u32x4 xs_all[MG > 1 ? NW * MG * 128 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (NTILE * 16);
  const int split = blockIdx.y;
  const int nchunk = (K + 63) >> 6, K32 = (K + 31) >> 5;
  const int cb = split * chunks_per_split, ce = min(nchunk, cb + chunks_per_split);

  // MG == 1: xp[0] = this lane's fragment slice (row li, 8 k of lane group g).  MG > 1: xp[i], i < 2 MG = line piece of instruction i:
  // row i*8 + (lane >> 3), 16-byte piece lane & 7 of the row's 128-byte chunk line (slots [h][mg] of the stage arrays hold i = h*MG + mg)
  const bf16_t* xp[MG > 1 ? 2 * MG : 1];
  if (MG == 1) {
    xp[0] = X + (size_t)min(li, M - 1) * ldx + g * 8;
  } else {
#pragma unroll
    for (int i = 0; i < 2 * MG; ++i) xp[i] = X + (size_t)min(i * 8 + (lane >> 3), M - 1) * ldx + (lane & 7) * 8;
  }
  const bf16_t* zp = reinterpret_cast<const bf16_t*>(lcc_zero_page_g) + (MG == 1 ? g * 8 : (lane & 7) * 8);
  u32x4* xs = xs_all + (MG > 1 ? wave * MG * 128 : 0);
  const bf16_t* wp[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
    wp[t] = PACKED ? W + (size_t)((n0 >> 4) + t) * K32 * 512 + lane * 8
                   : W + (size_t)min(n0 + t * 16 + li, N - 1) * ldw + g * 8;

  f32x4 acc[NTILE][MG];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) acc[t][mg] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // loads are unconditional (32-k block index clamped into range) so that the compiler emits straight-line loads with
  // counted waits; an absent block is cancelled by pointing the (shared) x fragment at a zero page.
  auto load_stage = [&](int c0, u32x4 (&wv)[UNR][2][NTILE], u32x4 (&xv)[UNR][2][MG]) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int cc = c0 + u * NW;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kb = 2 * cc + h;                        // 32-k block
        const bool ok = cc < ce && kb < K32;              // wave-uniform
        const int kbc = min(kb, K32 - 1);
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
          wv[u][h][t] = __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)kbc * (PACKED ? 512 : 32)));
        if (MG == 1) {
          xv[u][h][0] = ld16(ok ? xp[0] + kbc * 32 : zp);
        } else {       // full lines of the 64-k chunk (both k-blocks): instruction i = h*MG + mg; a chunk whose second k-block is past K reads the
                       // zero page for the whole chunk only when the chunk itself is absent (K % 64 == 32 never occurs with MG > 1: checked by the launcher)
#pragma unroll
          for (int mg = 0; mg < MG; ++mg) xv[u][h][mg] = ld16((cc < ce) ? xp[h * MG + mg] + min(cc, nchunk - 1) * 64 : zp);
        }
      }
    }
  };
  auto mma_stage = [&](const u32x4 (&wv)[UNR][2][NTILE], const u32x4 (&xv)[UNR][2][MG]) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      u32x4 xf[2][MG];
      if (MG == 1) {
        xf[0][0] = xv[u][0][0];
        xf[1][0] = xv[u][1][0];
      } else {
        // raw lines -> wave-private LDS image [row][8 x 16 B], chunk c of row r at r*8 + (c ^ (r & 7)) -> fragments (row mg*16 + li,
        // k-block h, lane group g).  One wave and in-order LDS: no s_barrier is needed, but the cross-LANE dependence (lane a's store, lane
        // b's load of the same address) is invisible in per-thread alias analysis -- a wavefront-scope fence + wave barrier (no
        // instructions on gfx950) keeps the compiler from reordering the two groups, here and in front of the next stage's stores.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2 * MG; ++i) {
          const int row = i * 8 + (lane >> 3);
          xs[row * 8 + ((lane & 7) ^ (row & 7))] = xv[u][i / MG][i % MG];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int mg = 0; mg < MG; ++mg) xf[h][mg] = xs[(mg * 16 + li) * 8 + ((h * 4 + g) ^ (li & 7))];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
#pragma unroll
          for (int mg = 0; mg < MG; ++mg) acc[t][mg] = mfma16(as_bf16x8(wv[u][h][t]), as_bf16x8(xf[h][mg]), acc[t][mg]);
    }
  };
  constexpr int STEP = UNR * NW;  // chunk stride of one stage
  if (PIPE == 0) {
    for (int c = cb + wave; c < ce; c += STEP) {
      u32x4 wa[UNR][2][NTILE], xa[UNR][2][MG];
      load_stage(c, wa, xa);
      mma_stage(wa, xa);
    }
  } else {
    u32x4 wa[UNR][2][NTILE], xa[UNR][2][MG], wb[UNR][2][NTILE], xb[UNR][2][MG];
    int c = cb + wave;
    load_stage(c, wa, xa);
    for (; c < ce; c += 2 * STEP) {
      // MG > 1: the issue order is PINNED (all loads of the next stage, then the whole current stage).  Left to itself hipcc sinks the next
      // stage's activation loads behind this stage's MFMAs and waits for them a few instructions later (vmcnt retires in order: the weight
      // loads issued before them are waited for too) -- the two-stage pipeline collapsed and the M = 32 gate/up GEMV ran at 2.8 TB/s
      // (98.8 us, profiles/r04/bench_32streams_mg_gemv_*.json)
      load_stage(c + STEP, wb, xb);
      if (MG > 1) __builtin_amdgcn_sched_barrier(0);
      mma_stage(wa, xa);
      if (MG > 1) __builtin_amdgcn_sched_barrier(0);
      load_stage(c + 2 * STEP, wa, xa);
      if (MG > 1) __builtin_amdgcn_sched_barrier(0);
      mma_stage(wb, xb);
      if (MG > 1) __builtin_amdgcn_sched_barrier(0);
    }
  }

  // cross-wave reduction
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) red[wave - 1][t * MG + mg][lane] = acc[t][mg];
  }
  __syncthreads();
  if (MODE == 3) {
    // split-K slabs + FUSED TAIL.  Publish/consume protocol (cdna_hip_programming.md G16 R1, counter form): write-through (sc1)
    // slab stores -> vmcnt(0) -> __syncthreads -> one lane: relaxed agent fetch_add (ticket).  The block that draws the last
    // ticket: one lane agent-scope acquire -> __syncthreads -> plain loads of every slab.  Placement-independent.
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) acc[t][0] += red[w][t * MG][lane];
      if (li < M) {
        float* o = (float*)out + ((size_t)split * M + li) * ldo;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          const int n = n0 + t * 16 + g * 4;
          if (n < N) {
            // write-through (sc1) slab stores: relaxed agent-scope 8-byte atomic stores lower to `global_store_dwordx2 sc0 sc1`,
            // so NO per-block release fence (buffer_wbl2) is needed -- measured: a release fence in each of the ~900 blocks
            // made the whole kernel ~17 us slower
            unsigned long long* p8 = reinterpret_cast<unsigned long long*>(o + n);
            const unsigned long long lo = (unsigned long long)__float_as_uint(acc[t][0][0]) | ((unsigned long long)__float_as_uint(acc[t][0][1]) << 32);
            const unsigned long long hi = (unsigned long long)__float_as_uint(acc[t][0][2]) | ((unsigned long long)__float_as_uint(acc[t][0][3]) << 32);
            __hip_atomic_store(p8, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p8 + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    }
    __syncthreads();
    float* flag = reinterpret_cast<float*>(&red[0][0][0]);   // all LDS in ONE array
    if (threadIdx.x == 0) {
      const int ticket = __hip_atomic_fetch_add(tail.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = (ticket == (int)(gridDim.x * gridDim.y) - 1) ? 1.f : 0.f;
    }
    __syncthreads();
    if (flag[0] == 0.f) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    const int nsplit = gridDim.y;
    if (tail.kind == 1) {
      for (int m = 0; m < M; ++m)
        tail_add_rmsnorm_row(tail.h + (size_t)m * N, (const float*)out, nsplit, (size_t)M * ldo, (size_t)m * ldo, tail.norm_w,
                             tail.y + (size_t)m * N, N, tail.eps, flag + 8);
    } else {
      RopeTailArgs ra{tail.bias, tail.cs, tail.sn, tail.tok_stream, tail.tok_pos, tail.kv_len, tail.kv_base, tail.lay, tail.layer,
                      tail.q_out, tail.n_q_heads};
      tail_rope_kv_append((const float*)out, nsplit, M, ra);
    }
    if (threadIdx.x == 0) __hip_atomic_store(tail.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    return;
  }
  if (wave > 0) return;
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
      for (int w = 0; w < NW - 1; ++w) acc[t][mg] += red[w][t * MG + mg][lane];

#pragma unroll
  for (int mg = 0; mg < MG; ++mg) {
    const int m = mg * 16 + li;
    if (m >= M) continue;
    if (MODE == 0) {
      float* o = (float*)out + ((size_t)split * M + m) * ldo;
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const int n = n0 + t * 16 + g * 4;
        if (n < N) *reinterpret_cast<f32x4*>(o + n) = acc[t][mg];
      }
    } else if (MODE == 1) {
      bf16_t* o = (bf16_t*)out + (size_t)m * ldo;
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const int n = n0 + t * 16 + g * 4;
        if (n >= N) continue;
        float v[4] = {acc[t][mg][0], acc[t][mg][1], acc[t][mg][2], acc[t][mg][3]};
        if (bias != nullptr) {
          u32x2 b = ld8(bias + n);
          v[0] += lo2f(b.x); v[1] += hi2f(b.x); v[2] += lo2f(b.y); v[3] += hi2f(b.y);
        }
        st8(o + n, (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])});
      }
    } else {  // swiglu: tile 0 = 16 gate rows, tile 1 = the 16 matching up rows
      static_assert(MODE != 2 || NTILE == 2, "swiglu needs the gate and up tile in one block");
      bf16_t* o = (bf16_t*)out + (size_t)m * ldo;
      const int oc = n0 / 2 + g * 4;
      if (n0 < N) {
        float r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = silu_bf16(rbf(acc[0][mg][q])) * rbf(acc[NTILE - 1][mg][q]);
        st8(o + oc, (u32x2){pack2(r[0], r[1]), pack2(r[2], r[3])});
      }
    }
  }
}

static int g_skinny_rows = 64;
// THE predicate of the weight-streaming route (the engine used to decide it from its own copy of the row limit, so a forced tile
// variant or a K that is not a multiple of 64 could leave it handing GEMV-derived split counts to the tiled kernels): M rows against a
// [N, K] weight go through gemv_skinny_kernel / gemv_w8_kernel iff this holds (and the epilogue is one the GEMV kernels have).
bool gemm_routes_skinny(int M, int K, bool w_fp8) {
  if (w_fp8) return M <= 16;
  return M <= (g_gemm_variant == 2 ? g_skinny_rows : 16) && (K % 32 == 0) && (M <= 16 || K % 64 == 0);
}
int set_skinny_rows(int rows) { const int old = g_skinny_rows; g_skinny_rows = rows < 16 ? 16 : (rows > 64 ? 64 : rows); return old; }
static int g_gemv_variant = 1;  // 0: UNR2 single stage; 1: UNR1 two-stage pipeline (default: best measured); 2: UNR2 two-stage
void set_gemv_variant(int v) { g_gemv_variant = v; }

template <int NTILE, int MODE, bool PACKED, int MG>
static void launch_gemv_mg(dim3 grid, const GemmArgs& a, void* out, int ldo, int cps, hipStream_t st) {
  if (MG > 1) {     // 17..64 rows: the two-stage pipeline with one chunk per stage (register budget: NTILE x MG accumulators + MG x fragments)
    gemv_skinny_kernel<NTILE, MODE, PACKED, 1, 1, MG><<<grid, dim3(256), 0, st>>>(a.A, a.lda, a.W, a.ldw, a.bias, out, ldo, a.M, a.N, a.K, cps, a.tail);
    return;
  }
  switch (g_gemv_variant) {
    case 1:
      gemv_skinny_kernel<NTILE, MODE, PACKED, 1, 1><<<grid, dim3(256), 0, st>>>(a.A, a.lda, a.W, a.ldw, a.bias, out, ldo, a.M, a.N, a.K, cps, a.tail);
      break;
    case 2:
      gemv_skinny_kernel<NTILE, MODE, PACKED, 2, 1><<<grid, dim3(256), 0, st>>>(a.A, a.lda, a.W, a.ldw, a.bias, out, ldo, a.M, a.N, a.K, cps, a.tail);
      break;
    default:
      gemv_skinny_kernel<NTILE, MODE, PACKED, 2, 0><<<grid, dim3(256), 0, st>>>(a.A, a.lda, a.W, a.ldw, a.bias, out, ldo, a.M, a.N, a.K, cps, a.tail);
  }
}
template <int NTILE, int MODE, bool PACKED>
static void launch_gemv(dim3 grid, const GemmArgs& a, void* out, int ldo, int cps, hipStream_t st) {
  const int mg = (a.M + 15) / 16;
  if (mg <= 1) launch_gemv_mg<NTILE, MODE, PACKED, 1>(grid, a, out, ldo, cps, st);
  else if (mg == 2) launch_gemv_mg<NTILE, MODE, PACKED, 2>(grid, a, out, ldo, cps, st);
  else if (mg == 3) launch_gemv_mg<NTILE, MODE, PACKED, 3>(grid, a, out, ldo, cps, st);
  else launch_gemv_mg<NTILE, MODE, PACKED, 4>(grid, a, out, ldo, cps, st);
}
template <int NTILE, int MODE>
static void launch_gemv_l(dim3 grid, const GemmArgs& a, void* out, int ldo, int cps, hipStream_t st) {
  if (a.w_packed) launch_gemv<NTILE, MODE, true>(grid, a, out, ldo, cps, st);
  else launch_gemv<NTILE, MODE, false>(grid, a, out, ldo, cps, st);
}

// ------------------------------------------------------------------------------------------------
// fp8 weights (OCP e4m3, per-output-row fp32 scale): decode GEMV + exact dequantisation for the tiled GEMMs
// ------------------------------------------------------------------------------------------------
// PACKED8 layout [N/16][K/64][4 g][16 rows][16 k] bytes: lane (g, row) of a 16-row x 64-k fragment owns the 16 consecutive
// k = kb*64 + g*16 .. +15 of its row, i.e. ONE 16-byte load per lane per fragment and 1 KB contiguous per wave load -- the same
// linear HBM stream as the bf16 packed layout at half the bytes.  The 16 bytes feed TWO 16x16x32 MFMAs (bytes 0-7, 8-15): the
// k -> MFMA-slot assignment is arbitrary as long as the activation fragment uses the same one (x[kb*64 + g*16 + 0..7] and
// + 8..15).  e4m3 -> bf16 is exact (v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32), products accumulate in fp32, and the row scale is
// applied once to the fp32 sum -- so the result is the bf16 GEMV of the exactly dequantised integers times the scale.

template <int NTILE, int MODE>   // MODE 0 fp32 split-K slabs, 1 bf16 + bias, 2 SwiGLU (tile 0 = 16 gate rows, tile 1 = their up rows)
__global__ __launch_bounds__(256) void gemv_w8_kernel(
    const bf16_t* __restrict__ X, int ldx, const uint8_t* __restrict__ W, const float* __restrict__ wscale,
    const bf16_t* __restrict__ bias, void* __restrict__ out, int ldo, int M, int N, int K, int chunks_per_split) {
  constexpr int NW = 4, UNR = 2;
  __shared__ f32x4 red[NW - 1][NTILE][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (NTILE * 16);
  const int split = blockIdx.y;
  const int nchunk = K >> 6;
  const int cb = split * chunks_per_split, ce = min(nchunk, cb + chunks_per_split);
  const int nfrag = N >> 4;

  const bf16_t* xp = X + (size_t)min(li, M - 1) * ldx + g * 16;
  const bf16_t* zp = reinterpret_cast<const bf16_t*>(lcc_zero_page_g) + g * 16;
  const uint8_t* wp[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) wp[t] = W + (size_t)min((n0 >> 4) + t, nfrag - 1) * nchunk * 1024 + lane * 16;

  f32x4 acc[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto load_stage = [&](int c0, u32x4 (&wv)[UNR][NTILE], u32x4 (&xv)[UNR][2]) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int cc = c0 + u * NW;
      const bool ok = cc < ce;                          // wave-uniform
      const int ccl = min(cc, nchunk - 1);
#pragma unroll
      for (int t = 0; t < NTILE; ++t) wv[u][t] = __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)ccl * 1024));
      const bf16_t* xs = ok ? xp + ccl * 64 : zp;       // an absent chunk is cancelled by a zero activation fragment
      xv[u][0] = ld16(xs);
      xv[u][1] = ld16(xs + 8);
    }
  };
  auto mma_stage = [&](const u32x4 (&wv)[UNR][NTILE], const u32x4 (&xv)[UNR][2]) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        acc[t] = mfma16(fp8x8_to_bf16x8(wv[u][t][0], wv[u][t][1]), as_bf16x8(xv[u][0]), acc[t]);
        acc[t] = mfma16(fp8x8_to_bf16x8(wv[u][t][2], wv[u][t][3]), as_bf16x8(xv[u][1]), acc[t]);
      }
  };
  constexpr int STEP = UNR * NW;
  {
    u32x4 wa[UNR][NTILE], xa[UNR][2], wb[UNR][NTILE], xb[UNR][2];
    int c = cb + wave;
    load_stage(c, wa, xa);
    for (; c < ce; c += 2 * STEP) {
      load_stage(c + STEP, wb, xb);
      mma_stage(wa, xa);
      load_stage(c + 2 * STEP, wa, xa);
      mma_stage(wb, xb);
    }
  }
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
    acc[t] *= *reinterpret_cast<const f32x4*>(wscale + min(n0 + t * 16 + g * 4, N - 4));
  }
  if (li >= M) return;
  if (MODE == 0) {
    float* o = (float*)out + ((size_t)split * M + li) * ldo;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      const int n = n0 + t * 16 + g * 4;
      if (n < N) *reinterpret_cast<f32x4*>(o + n) = acc[t];
    }
  } else if (MODE == 1) {
    bf16_t* o = (bf16_t*)out + (size_t)li * ldo;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      const int n = n0 + t * 16 + g * 4;
      if (n >= N) continue;
      float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
      if (bias != nullptr) {
        u32x2 b = ld8(bias + n);
        v[0] += lo2f(b.x); v[1] += hi2f(b.x); v[2] += lo2f(b.y); v[3] += hi2f(b.y);
      }
      st8(o + n, (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])});
    }
  } else {
    static_assert(MODE != 2 || NTILE == 2, "swiglu needs the gate and up tile in one block");
    bf16_t* o = (bf16_t*)out + (size_t)li * ldo;
    const int oc = n0 / 2 + g * 4;
    if (n0 < N) {
      float r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = silu_bf16(rbf(acc[0][q])) * rbf(acc[NTILE - 1][q]);
      st8(o + oc, (u32x2){pack2(r[0], r[1]), pack2(r[2], r[3])});
    }
  }
}

// PACKED8 fp8 -> bf16 packed fragments (exact; no scale): thread = one 16-byte piece (row, 16 k) of a 16 x 64 fragment ->
// the two 16-byte pieces (g' = 2*(g&1), 2*(g&1)+1) of the bf16 fragment of 32-k block 2*kb + (g >> 1)
__global__ __launch_bounds__(256) void dequant_w8_kernel(const uint8_t* __restrict__ W8, bf16_t* __restrict__ out, size_t n_pieces) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pieces) return;
  const size_t frag = i >> 6;                    // (row tile, 64-k block)
  const int lane = (int)(i & 63), g = lane >> 4, row = lane & 15;
  const u32x4 q = ld16(W8 + i * 16);
  const bf16x8 lo = fp8x8_to_bf16x8(q[0], q[1]), hi = fp8x8_to_bf16x8(q[2], q[3]);
  // bf16 packed: [row tile][32-k block][4 g'][16 rows][8 k]; 32-k block index = 2*kb64 + (g >> 1); frag already = rt*nchunk + kb64
  bf16_t* dst = out + (frag * 2 + (size_t)(g >> 1)) * 512 + ((g & 1) * 2 * 16 + row) * 8;
  st16(dst, as_u32x4(lo));
  st16(dst + 16 * 8, as_u32x4(hi));
}

// layout probe (tests/test_gpu_ops.py::test_mfma_layout_probe): D[16x16] = A[16x32] * B[32x16] with the fragment maps of
// common.h, one wave.  Verifies the operand/result lane maps every kernel in this library relies on.
__global__ __launch_bounds__(64) void mfma_probe_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                        float* __restrict__ D) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  bf16_t av[8], bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    av[e] = A[i * 32 + g * 8 + e];        // A[i][k = g*8+e]
    bv[e] = B[(g * 8 + e) * 16 + i];      // B[k = g*8+e][j = i]
  }
  u32x4 a = (u32x4){(unsigned)av[0] | ((unsigned)av[1] << 16), (unsigned)av[2] | ((unsigned)av[3] << 16),
                    (unsigned)av[4] | ((unsigned)av[5] << 16), (unsigned)av[6] | ((unsigned)av[7] << 16)};
  u32x4 b = (u32x4){(unsigned)bv[0] | ((unsigned)bv[1] << 16), (unsigned)bv[2] | ((unsigned)bv[3] << 16),
                    (unsigned)bv[4] | ((unsigned)bv[5] << 16), (unsigned)bv[6] | ((unsigned)bv[7] << 16)};
  f32x4 c = mfma16(as_bf16x8(a), as_bf16x8(b), (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
  for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + i] = c[r];  // D[row = g*4+r][col = i]
}
int mfma_probe(const bf16_t* A, const bf16_t* B, float* D, hipStream_t st) {
  mfma_probe_kernel<<<dim3(1), dim3(64), 0, st>>>(A, B, D);
  return 0;
}

// number of inter-block K splits of a skinny GEMV: enough blocks (x4 waves) to keep >= ~16 waves per CU in flight,
// every wave keeps >= ~3 64-element chunks, and the fp32 slab traffic stays small (S <= 8)
int gemv_num_splits(int N, int K) {
  // tuning knobs (debug): LCC_GEMV_TARGET = blocks aimed at (default 1024), LCC_GEMV_MINCHUNK = 64-k chunks per split at least (12)
  static const int target = [] { const char* e = getenv("LCC_GEMV_TARGET"); return e ? std::max(64, atoi(e)) : 1024; }();
  static const int minchunk = [] { const char* e = getenv("LCC_GEMV_MINCHUNK"); return e ? std::max(1, atoi(e)) : 12; }();
  const int tiles = (N + 15) / 16, nchunk = (K + 63) / 64;
  int s = (target + tiles - 1) / tiles;
  s = std::min(s, std::max(1, nchunk / minchunk));
  return std::max(1, std::min(8, s));
}

// split-K factor for a tiled GEMM whose output grid alone cannot fill 256 CUs (e.g. M = 386, N = 3584: 196 tiles)
int gemm_tiled_num_splits(int M, int N, int K, bool packed_bf16) {
  const long tiles = (long)((M + 63) / 64) * ((N + 127) / 128);
  if (tiles >= 400 || M > 4096) return 1;
  int s = (int)((640 + tiles - 1) / tiles);
  s = std::min(s, std::max(1, ((K + 63) / 64) / 8));
  s = std::max(1, std::min(4, s));
  // small variable-height tiles (bf16 weights): floor(F / 8) row tiles x 256-column tiles -> as many splits as fill one round of the chip
  // (7B chunk, M = 386: o / down 3 x 14 tiles x 6 = 252 blocks, q|k|v 3 x 18 x 4 = 216), never fewer than the 128-row rule gives
  if (packed_bf16 && vh_small_on() && vh_small_shape(M, K) && (N & 15) == 0) {
    const long vt = (long)(((M + 15) >> 4) >> 3) * ((N + 255) / 256);
    int sv = (int)std::min<long>(6, 256 / std::max<long>(1, vt));
    sv = std::min(sv, std::max(1, ((K + 63) / 64) / 8));
    s = std::max(s, sv);
  }
  return s;
}

static int gemm_w8(const GemmArgs& a, hipStream_t st) {
  if ((a.K & 63) || (a.N & 15) || (a.lda & 7) || a.wscale == nullptr) return LCC_ERR_SHAPE;
  if ((((uintptr_t)a.A | (uintptr_t)a.W) & 15) != 0 || ((uintptr_t)a.wscale & 15) != 0) return LCC_ERR_ALIGN;
  if (a.epilogue == EPI_SWIGLU && (a.N & 31)) return LCC_ERR_SHAPE;
  const uint8_t* W8 = reinterpret_cast<const uint8_t*>(a.W);
  const bool skinny = a.M <= 16 && a.epilogue != EPI_QUICK_GELU && a.epilogue != EPI_GELU_ERF && a.epilogue != EPI_RESIDUAL;
  if (skinny) {
    const int nchunk = a.K / 64;
    if (a.epilogue == EPI_SWIGLU) {
      gemv_w8_kernel<2, 2><<<dim3((a.N + 31) / 32, 1), dim3(256), 0, st>>>(a.A, a.lda, W8, a.wscale, nullptr, a.C, a.ldc, a.M, a.N, a.K, nchunk);
    } else if (a.partial != nullptr) {
      const int S = a.nsplit > 0 ? a.nsplit : 1;
      if (S > nchunk || a.tail.kind != 0) return LCC_ERR_ARG;
      gemv_w8_kernel<1, 0><<<dim3((a.N + 15) / 16, S), dim3(256), 0, st>>>(a.A, a.lda, W8, a.wscale, nullptr, a.partial, a.N, a.M, a.N, a.K,
                                                                             (nchunk + S - 1) / S);
    } else {
      gemv_w8_kernel<1, 1><<<dim3((a.N + 15) / 16, 1), dim3(256), 0, st>>>(a.A, a.lda, W8, a.wscale, a.bias, a.C, a.ldc, a.M, a.N, a.K, nchunk);
    }
    return 0;
  }
  // M > 16: the 8-wave kernel consumes the fp8 fragments directly (e4m3 -> bf16 after the LDS read) ...
  {
    GemmArgs c = a; c.w_packed = 1;
    const bool part = a.partial != nullptr;
    if (part && (a.epilogue != EPI_NONE || a.nsplit < 1 || a.nsplit > 8 || a.nsplit > a.K / 64)) return LCC_ERR_ARG;
    if (a.epilogue == EPI_RESIDUAL && a.residual == nullptr) return LCC_ERR_ARG;
    const int big = (g_gemm_variant == 7) ? 0 : big_tile_rows(c, part ? a.nsplit : 1);
    if (big != 0) {
#define LCC_BIG8(EPI) (big == 256 ? launch_big<256, EPI>(c, st) : launch_big<128, EPI>(c, st))
      if (part) { LCC_BIG8(EPI_PARTIAL); return 0; }
      switch (a.epilogue) {
        case EPI_NONE: LCC_BIG8(EPI_NONE); return 0;
        case EPI_QUICK_GELU: LCC_BIG8(EPI_QUICK_GELU); return 0;
        case EPI_GELU_ERF: LCC_BIG8(EPI_GELU_ERF); return 0;
        case EPI_RESIDUAL: LCC_BIG8(EPI_RESIDUAL); return 0;
        case EPI_SWIGLU: LCC_BIG8(EPI_SWIGLU); return 0;
        default: return LCC_ERR_ARG;
      }
#undef LCC_BIG8
    }
  }
  // ... shapes with too few 256-column tiles: exact dequantisation into the bf16 packed order, then the 4-wave bf16 GEMM with
  // the row scale in its epilogue
  if (a.dq_scratch == nullptr) return LCC_ERR_ARG;
  const size_t pieces = (size_t)a.N * a.K / 16;
  dequant_w8_kernel<<<dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st>>>(W8, a.dq_scratch, pieces);
  GemmArgs b = a;
  b.w_fp8 = 0; b.W = a.dq_scratch; b.w_packed = 1; b.ldw = a.K;
  return gemm_bf16(b, st);
}

int gemm_bf16(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return 0;
  if (a.w_fp8) return gemm_w8(a, st);
  if ((a.K & 7) || (a.N & 15) || (a.lda & 7) || (!a.w_packed && (a.ldw & 7)) || (a.ldc & 3)) return LCC_ERR_SHAPE;
  if ((((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.C) & 15) != 0) return LCC_ERR_ALIGN;
  if (a.epilogue == EPI_SWIGLU && (a.N & 31)) return LCC_ERR_SHAPE;
  if (a.epilogue == EPI_RESIDUAL && a.residual == nullptr) return LCC_ERR_ARG;
  if (a.epilogue == EPI_VIT_QKV || a.epilogue == EPI_VIT_QK || a.epilogue == EPI_VIT_V) {
    // vision-tower q|k|v projection with RoPE / the V transpose in the epilogue: all three in one launch (E % 128 == 0), or q|k and V apart
    const VitQkvEpi& q = a.vq;
    const bool all = a.epilogue == EPI_VIT_QKV, qk = all || a.epilogue == EPI_VIT_QK, v = all || a.epilogue == EPI_VIT_V;
    if (!a.w_packed || a.partial != nullptr || !gemm_vit_qkv_eligible(a.M, q.E, a.K) || a.N != (all ? 3 : qk ? 2 : 1) * q.E || (all && (q.E & 127))) return LCC_ERR_ARG;
    if (qk && (q.cs == nullptr || q.sn == nullptr || a.C == nullptr)) return LCC_ERR_ARG;
    if (v && (q.grp_off == nullptr || q.vt == nullptr || q.total_blocks <= 0)) return LCC_ERR_ARG;
    if (qk && (((uintptr_t)q.cs | (uintptr_t)q.sn) & 15) != 0) return LCC_ERR_ALIGN;
    if ((v && ((uintptr_t)q.vt & 7) != 0) || (a.bias != nullptr && ((uintptr_t)a.bias & 7) != 0)) return LCC_ERR_ALIGN;
    g_launch_counts[LC_GEMM_VIT_QKV]++;
    int big = big_tile_rows(a, 1);
    if (big == 272) big = 256;      // the rotation / transposed-V epilogues live on the fixed-height tiles
    if (a.epilogue == EPI_VIT_QKV) {
      if (big == 256) launch_big_s<256, EPI_VIT_QKV, 6, false>(a, st);
      else if (big == 192) launch_big_s<192, EPI_VIT_QKV, 6, false>(a, st);
      else launch_big_s<128, EPI_VIT_QKV, 6, false>(a, st);
    } else if (qk) {
      if (big == 256) launch_big_s<256, EPI_VIT_QK, 6, false>(a, st);
      else if (big == 192) launch_big_s<192, EPI_VIT_QK, 6, false>(a, st);
      else launch_big_s<128, EPI_VIT_QK, 6, false>(a, st);
    } else {
      if (big == 256) launch_big_s<256, EPI_VIT_V, 6, false>(a, st);
      else if (big == 192) launch_big_s<192, EPI_VIT_V, 6, false>(a, st);
      else launch_big_s<128, EPI_VIT_V, 6, false>(a, st);
    }
    return 0;
  }
  // weight-streaming path: up to g_skinny_rows rows (64: decode batches of 17-64 streams multiply every weight fragment with 2-4 activation
  // fragments; lcc_debug_set_skinny_rows(16) restores the round-3 routing of 17-64 rows through the 64-row GEMM tiles)
  // (a forced GEMM tile variant -- tests, A/B runs -- keeps 17-64 rows on the tiles it asks for)
  const bool skinny = gemm_routes_skinny(a.M, a.K, false) && a.epilogue != EPI_QUICK_GELU && a.epilogue != EPI_GELU_ERF && a.epilogue != EPI_RESIDUAL;
  if (skinny) {
    const int nchunk = (a.K + 63) / 64;
    if (a.epilogue == EPI_SWIGLU) {
      launch_gemv_l<2, 2>(dim3((a.N + 31) / 32, 1), a, a.C, a.ldc, nchunk, st);
    } else if (a.partial != nullptr) {
      const int S = a.nsplit > 0 ? a.nsplit : 1;
      if (S > nchunk) return LCC_ERR_SHAPE;
      if (a.tail.kind != 0) {   // fused tail: last-arriving block reduces the slabs and runs the consumer (M <= 2, packed W)
        if (a.M > 2 || !a.w_packed || a.tail.counter == nullptr || S > 8 || (a.tail.kind == 1 && (a.N > 8192 || (a.N & 7)))) return LCC_ERR_ARG;
        g_launch_counts[LC_GEMV_FUSED_TAIL]++;
        gemv_skinny_kernel<1, 3, true, 1, 1><<<dim3((a.N + 15) / 16, S), dim3(256), 0, st>>>(
            a.A, a.lda, a.W, a.ldw, nullptr, a.partial, a.N, a.M, a.N, a.K, (nchunk + S - 1) / S, a.tail);
        return 0;
      }
      launch_gemv_l<1, 0>(dim3((a.N + 15) / 16, S), a, a.partial, a.N, (nchunk + S - 1) / S, st);
    } else {
      launch_gemv_l<1, 1>(dim3((a.N + 15) / 16, 1), a, a.C, a.ldc, nchunk, st);
    }
    return 0;
  }
  if (a.partial != nullptr) {   // split-K slabs on the tiled path (prefill GEMMs with few output tiles)
    if (a.epilogue != EPI_NONE || a.nsplit < 1 || a.nsplit > 8 || a.nsplit > (a.K + 63) / 64) return LCC_ERR_ARG;
    const int big = big_tile_rows(a, a.nsplit);
    if (big == 272) launch_vh<EPI_PARTIAL>(a, st);
    else if (big == 144) launch_vh<EPI_PARTIAL, 1>(a, st);
    else if (big == 256) launch_big<256, EPI_PARTIAL>(a, st);
    else if (big == 192) launch_big<192, EPI_PARTIAL>(a, st);
    else if (big == 128) launch_big<128, EPI_PARTIAL>(a, st);
    else launch_tiled<64, EPI_PARTIAL>(a, st);
    return 0;
  }
  switch (a.epilogue) {
    case EPI_NONE: launch_tiled_bm<EPI_NONE>(a, st); break;
    case EPI_QUICK_GELU: launch_tiled_bm<EPI_QUICK_GELU>(a, st); break;
    case EPI_GELU_ERF: launch_tiled_bm<EPI_GELU_ERF>(a, st); break;
    case EPI_RESIDUAL: launch_tiled_bm<EPI_RESIDUAL>(a, st); break;
    case EPI_SWIGLU: launch_tiled_bm<EPI_SWIGLU>(a, st); break;
    default: return LCC_ERR_ARG;
  }
  return 0;
}

}  // namespace lcc
