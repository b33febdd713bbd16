// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the LiveCC hot path.
// wave = 64 lanes; MFMA fragments follow the gfx950 16x16x32 bf16 layout:
//   A operand: lane l holds A[i = l&15][k = (l>>4)*8 + e], e = 0..7   (8 bf16 = 16 bytes)
//   B operand: lane l holds B[k = (l>>4)*8 + e][j = l&15]
//   C/D      : lane l holds D[row = (l>>4)*4 + r][col = l&15], r = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcc {

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LCC_DEVICE __device__ __forceinline__

LCC_DEVICE float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// round-to-nearest-even fp32 -> bf16 (same rounding as torch's c10::BFloat16): gfx950 has it in hardware
// (v_cvt_pk_bf16_f32), which the __bf16 conversions below lower to -- one VALU op per PAIR instead of ~10 per element.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
LCC_DEVICE bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// value of x after rounding to bf16 (used to reproduce HF's per-op bf16 rounding points)
LCC_DEVICE float rbf(float x) { return (float)(__bf16)x; }

LCC_DEVICE unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
LCC_DEVICE float lo2f(unsigned v) { return __uint_as_float(v << 16); }
LCC_DEVICE float hi2f(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

LCC_DEVICE bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
LCC_DEVICE u32x4 as_u32x4(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }

LCC_DEVICE u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
LCC_DEVICE void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
LCC_DEVICE u32x2 ld8(const void* p) { return *reinterpret_cast<const u32x2*>(p); }
LCC_DEVICE void st8(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }

#ifndef LCC_GLDS_POLICY      // cache-policy suffix of the LDS-DMA instruction (build-time experiment knob: " nt", " sc0", ...)
#define LCC_GLDS_POLICY ""
#endif
// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4): lane l's 16 bytes at `gsrc` land at LDS byte address `lds_dst` + l * 16
// (`lds_dst` wave-uniform).  INLINE ASM ON PURPOSE (round 5): the builtin is a FLAT-encoded instruction with an LDS memory operand, so
// hipcc's waitcnt pass books it as "may access LDS through flat" -- from then on EVERY s_waitcnt it inserts in front of a ds_read
// consumer is lgkmcnt(0) (and vmcnt(0) in front of the use of any plain load), i.e. the counted fragment-read pipelines of the MFMA
// kernels collapsed into read-wait-use (ISA of gemm_big_kernel: an lgkmcnt(0) right behind the ds_read issued for two steps later, every
// third row-tile step).  Hidden in an asm statement the DMA is absent from the compiler's bookkeeping: its completion is counted by
// the kernels' own `s_waitcnt vmcnt(N)` + barrier (they did that already), and the ds_read waits become lgkmcnt(N) ladders.
// M0 is saved and restored around the statement (cdna_hip_programming.md section 5.7); the s_nop covers the M0 write -> LDS-DMA hazard.
LCC_DEVICE void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" LCC_GLDS_POLICY "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");   // readfirstlane: free when provably uniform
}
// Make the compiler wait HERE for a plain global load it is tracking (it must insert its own s_waitcnt in front of a statement that reads
// the registers).  Needed wherever plain loads are followed by glds16() pieces: a vmcnt(N) the compiler places later -- at the first use
// of the value, possibly inside the ring loop -- would also drain the LDS-DMAs it cannot see (vmcnt retires in order).
LCC_DEVICE void settle_load(u32x4& v) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); }
// LDS byte address of a __shared__ object (wave-uniform when the pointer is)
LCC_DEVICE unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }

LCC_DEVICE f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 8 OCP e4m3 bytes (two dwords) -> 8 bf16 (exact): v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32
LCC_DEVICE bf16x8 fp8x8_to_bf16x8(unsigned a, unsigned b) {
  const f32x2_t a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)a, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)a, true);
  const f32x2_t b0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, false), b1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, true);
  return as_bf16x8((u32x4){pack2(a0[0], a0[1]), pack2(a1[0], a1[1]), pack2(b0[0], b0[1]), pack2(b1[0], b1[1])});
}

// wave-wide reductions (64 lanes)
LCC_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
LCC_DEVICE float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blocks of NW waves; `red` is a shared float[NW] scratch
template <int NW>
LCC_DEVICE float block_sum(float v, float* red) {
  v = wave_sum(v);
  if (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

// activations, with HF's bf16 rounding points (inputs are already-rounded bf16 values held in fp32).
// exp and the reciprocal use the hardware units (v_exp_f32, v_rcp_f32) with the error terms folded back in:
//   exp(t): t*log2(e) split into hi + lo (FMA residual), exp2(hi) * (1 + lo*ln2)      -> ~1 ulp of fp32
//   1/d   : v_rcp_f32 + one Newton step                                              -> ~0.5 ulp of fp32
// i.e. 10 VALU instructions instead of ~30 for libm expf + IEEE division; the fp32 result is then rounded to bf16 (2^-9
// relative), so it differs from the correctly rounded activation in ~5e-5 of the elements by one bf16 ulp -- two orders of
// magnitude below the effect of the fp32 summation order of the GEMM that feeds it.  (The epilogue was 9 % of the SwiGLU
// gate/up GEMM and ~25 % of the ViT fc1 GEMM.)
LCC_DEVICE float exp_fast(float t) {
  const float L2E_HI = 1.4426950216293335f, L2E_LO = 1.9259629911266175e-8f;   // log2(e) = hi + lo
  const float hi = t * L2E_HI;
  const float lo = fmaf(t, L2E_HI, -hi) + t * L2E_LO;
  return __builtin_amdgcn_exp2f(hi) * fmaf(lo, 0.6931471805599453f, 1.0f);
}
LCC_DEVICE float rcp_fast(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
// quick_gelu: HF activations.py QuickGELUActivation: input * sigmoid(1.702 * input) on bf16 tensors
LCC_DEVICE float quick_gelu_bf16(float x) {
  float t = rbf(1.702f * x);
  float s = rbf(rcp_fast(1.0f + exp_fast(-t)));
  return rbf(x * s);
}
// exact (erf) GELU: torch gelu on bf16 computes in fp32 and rounds once
LCC_DEVICE float gelu_erf_bf16(float x) { return rbf(0.5f * x * (1.0f + erff(x * 0.70710678118654752440f))); }
// silu on bf16: x / (1 + exp(-x)) in fp32, rounded once
LCC_DEVICE float silu_bf16(float x) { return rbf(x * rcp_fast(1.0f + exp_fast(-x))); }

enum Epilogue : int {
  EPI_NONE = 0,        // C = bf16(acc + bias)
  EPI_QUICK_GELU = 1,  // C = quick_gelu(bf16(acc + bias))
  EPI_GELU_ERF = 2,    // C = gelu(bf16(acc + bias))
  EPI_RESIDUAL = 3,    // C = bf16(residual + bf16(acc + bias))
  EPI_SWIGLU = 4,      // W rows interleaved [16 gate | 16 up]; C[:, n/2] = bf16(silu(bf16 g) * bf16 u)
  EPI_PARTIAL = 5,     // internal: raw fp32 split-K slab [split][M][N] (consumer reduces, adds bias, rounds)
  EPI_VIT_QK = 6,      // internal: vision-tower q|k projection, W rows in the `qkv_w_rope` order: 2-D RoPE in the epilogue, stored at the natural columns
  EPI_VIT_V = 7,       // internal: vision-tower V projection, MFMA operands swapped: stored blocked-transposed for the attention kernels
  EPI_VIT_QKV = 8,     // internal: both in one launch (N = 3E): column tiles below 2E run the q|k body, the others the V body
};

// ViT 2-D RoPE on one rotation pair (x1, x2) = (channel c, channel c + 40) of a head: HF apply_rotary_pos_emb_vision
// (modeling_qwen2_vl.py:225-248) computes q*cos + rotate_half(q)*sin in fp32 -- two separately rounded products and one add, no FMA.
// Contraction is switched off here so that every kernel applying the rotation (vit_rope_vt_kernel, the q|k|v GEMM epilogue) produces the
// same bits, and those are HF's.
LCC_DEVICE void vit_rope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
#pragma clang fp contract(off)
  o1 = x1 * c - x2 * s;
  o2 = x2 * c + x1 * s;
}

}  // namespace lcc
