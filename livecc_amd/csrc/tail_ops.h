// Block-level "tail" operations shared by the stand-alone elementwise kernels and by the fused GEMV tails (gemm.hip):
// the LAST-ARRIVING block of a split-K weight-streaming GEMV reduces the fp32 slabs and runs the consumer op
// (residual add + RMSNorm, or bias + M-RoPE + KV append) itself, which removes one kernel launch + boundary per site on the
// batch-1 decode path.  All functions are called by every thread of a 256-thread block.
#pragma once
#include "common.h"
#include "kernels.h"

namespace lcc {

// sum of up to 8 split-K slabs (slab order), runtime count, straight-line: the slab index is clamped and absent slabs are
// multiplied by 0 (x*1 and +0 are exact, so the result is bit-identical to summing exactly `nsplit` slabs in order)
LCC_DEVICE void sum_slabs8(const float* __restrict__ part, int nsplit, size_t slab_stride, size_t off, float (&d)[8]) {
  f32x4 pa[8], pb[8];
#pragma unroll
  for (int sp = 0; sp < 8; ++sp) {
    const float* pp = part + (size_t)min(sp, nsplit - 1) * slab_stride + off;
    pa[sp] = *reinterpret_cast<const f32x4*>(pp);
    pb[sp] = *reinterpret_cast<const f32x4*>(pp + 4);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) d[e] = 0.f;
#pragma unroll
  for (int sp = 0; sp < 8; ++sp) {
    const float k = sp < nsplit ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] += pa[sp][e] * k; d[4 + e] += pb[sp][e] * k; }
  }
}

// one row: h += bf16(sum of slabs); y = rmsnorm(h) * w   (HF rounding points, see add_rmsnorm_kernel).  256 threads.
LCC_DEVICE void tail_add_rmsnorm_row(bf16_t* __restrict__ hr, const float* __restrict__ part, int nsplit, size_t slab_stride,
                                     size_t row_off, const bf16_t* __restrict__ w, bf16_t* __restrict__ yr, int dim, float eps,
                                     float* red) {
  constexpr int MAXC = 4;
  const int tid = threadIdx.x, nchunk = dim / 8;
  float v[MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = tid + c * 256;
    if (ch < nchunk) {
      const u32x4 q = ld16(hr + ch * 8);
      float d[8];
      sum_slabs8(part, nsplit, slab_stride, row_off + ch * 8, d);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[c][2 * e] = lo2f(q[e]); v[c][2 * e + 1] = hi2f(q[e]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = rbf(v[c][e] + rbf(d[e]));
      st16(hr + ch * 8, (u32x4){pack2(v[c][0], v[c][1]), pack2(v[c][2], v[c][3]), pack2(v[c][4], v[c][5]), pack2(v[c][6], v[c][7])});
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[c][e] * v[c][e];
    }
  }
  if (w == nullptr) return;
  const float var = block_sum<4>(s, red) / (float)dim;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = tid + c * 256;
    if (ch < nchunk) {
      const u32x4 wq = ld16(w + ch * 8);
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = lo2f(wq[e]) * rbf(v[c][2 * e] * rstd);
        o[2 * e + 1] = hi2f(wq[e]) * rbf(v[c][2 * e + 1] * rstd);
      }
      st16(yr + ch * 8, (u32x4){pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])});
    }
  }
}

struct RopeTailArgs {
  const bf16_t* bias; const bf16_t* cs; const bf16_t* sn;
  const int32_t* tok_stream; const int32_t* tok_pos; const int32_t* kv_len;
  bf16_t* const* kv_base; KvLayout lay; int layer; bf16_t* q_out; int n_q_heads;
};

// 8 values of the q/k/v Linear output of token s at column col = bf16(sum of slabs + bias)
LCC_DEVICE void tail_qkv8(const float* __restrict__ part, int nsplit, int S, const bf16_t* __restrict__ bias, int s, int ld, int col,
                          float (&v)[8]) {
  sum_slabs8(part, nsplit, (size_t)S * ld, (size_t)s * ld + col, v);
  const u32x4 bq = ld16(bias + col);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = rbf(v[2 * e] + lo2f(bq[e])); v[2 * e + 1] = rbf(v[2 * e + 1] + hi2f(bq[e])); }
}

// all S tokens: M-RoPE on q,k + KV append (the body of rope_kv_append_kernel<1,*>), looped over by one 256-thread block
LCC_DEVICE void tail_rope_kv_append(const float* __restrict__ part, int nsplit, int S, const RopeTailArgs& a) {
  constexpr int D = 128;
  const int hkv = a.lay.n_kv_heads, nq = a.n_q_heads;
  const int per_tok = (nq + hkv) * 8 + hkv * 16;
  const int ld = (nq + 2 * hkv) * D;
  for (int idx = threadIdx.x; idx < S * per_tok; idx += 256) {
    const int s = idx / per_tok;
    int it = idx - s * per_tok;
    const int strm = a.tok_stream[s];
    const int slot = a.tok_pos != nullptr ? a.tok_pos[s] : a.kv_len[strm];
    bf16_t* base = a.kv_base[strm] + (size_t)a.layer * a.lay.layer_stride();
    if (it < (nq + hkv) * 8) {
      const int head = it >> 3, c0 = (it & 7) * 8;
      float x1[8], x2[8], o1[8], o2[8];
      tail_qkv8(part, nsplit, S, a.bias, s, ld, head * D + c0, x1);
      tail_qkv8(part, nsplit, S, a.bias, s, ld, head * D + c0 + 64, x2);
      const u32x4 cq = ld16(a.cs + (size_t)s * 64 + c0), sq = ld16(a.sn + (size_t)s * 64 + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float c = (e & 1) ? hi2f(cq[e >> 1]) : lo2f(cq[e >> 1]);
        const float sv = (e & 1) ? hi2f(sq[e >> 1]) : lo2f(sq[e >> 1]);
        o1[e] = rbf(x1[e] * c) + rbf(-x2[e] * sv);
        o2[e] = rbf(x2[e] * c) + rbf(x1[e] * sv);
      }
      bf16_t* dst = head < nq ? a.q_out + (size_t)s * nq * D + head * D
                              : base + (size_t)(head - nq) * a.lay.head_stride() + (size_t)slot * D;
      st16(dst + c0, (u32x4){pack2(o1[0], o1[1]), pack2(o1[2], o1[3]), pack2(o1[4], o1[5]), pack2(o1[6], o1[7])});
      st16(dst + c0 + 64, (u32x4){pack2(o2[0], o2[1]), pack2(o2[2], o2[3]), pack2(o2[4], o2[5]), pack2(o2[6], o2[7])});
    } else {
      it -= (nq + hkv) * 8;
      const int hv = it >> 4, c0 = (it & 15) * 8;
      float x[8];
      tail_qkv8(part, nsplit, S, a.bias, s, ld, (nq + hkv + hv) * D + c0, x);
      bf16_t* dst = base + a.lay.kv_stride() + (size_t)hv * a.lay.head_stride() + ((size_t)(slot >> 5) * D + c0) * 32 + (slot & 31);
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e * 32] = f2bf(x[e]);
    }
  }
}

}  // namespace lcc
