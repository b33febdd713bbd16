// PyTorch-ROCm custom ops over the C-ABI (round 6; SURVEY 8b "Native (C-ABI / torch custom-op) layer", BASELINE north_star: "host
// orchestration stays Python calling into PyTorch-ROCm custom ops").
//
// The C-ABI of include/livecc_amd.h stays the source of truth: every op below validates its tensors with TORCH_CHECK (dtype, device,
// contiguity, shapes -- never UB on a bad shape), allocates its outputs as torch tensors, and forwards to the SAME `lcc_*` symbol the
// ctypes binding (livecc_amd/ops.py) calls, on `c10::hip::getCurrentHIPStream()`.  No arithmetic here.  What the registration buys over
// ctypes-on-data_ptr(): the ops are visible to the dispatcher -- `torch.ops.livecc_amd.*`, profiler ranges (`record_function` shows
// `livecc_amd::rmsnorm`), schema-checked arguments, mutation annotations for the in-place KV append, usable from TorchScript / an
// exported graph.  They are inference ops: no autograd formulas (CompositeExplicitAutograd is deliberately NOT claimed; a tensor that
// requires grad is refused).
//
// Built by livecc_amd/build.py (g++, host code only) into livecc_amd/_C/liblivecc_torch_ops.so, linked against liblivecc_amd.so with
// rpath $ORIGIN; loaded by livecc_amd/torch_ops.py (torch.ops.load_library).  The HF operator plugins (plugin.py) call through
// torch.ops when the library is present.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <array>
#include <vector>

#include "../../include/livecc_amd.h"

namespace {

using at::Tensor;
using c10::optional;

void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void chk(const Tensor& t, at::ScalarType dt, const char* name) {
  TORCH_CHECK(t.defined(), name, ": undefined tensor");
  TORCH_CHECK(t.is_cuda(), name, ": expected a GPU tensor (livecc_amd has no CPU path)");
  TORCH_CHECK(t.scalar_type() == dt, name, ": expected ", dt, ", got ", t.scalar_type());
  TORCH_CHECK(t.is_contiguous(), name, ": tensor must be contiguous");
  // inference ops: no autograd formula.  A parameter of an HF module has requires_grad = true even under torch.no_grad() / inference_mode(),
  // so the refusal applies only where a graph would actually be recorded
  TORCH_CHECK(!(at::GradMode::is_enabled() && t.requires_grad()), name, ": livecc_amd ops are inference ops (no autograd formula): call under torch.no_grad()");
  // the kernels read bf16 / fp32 operands with 16-byte loads (index tables and uint8 frames have no such requirement)
  if (dt == at::kBFloat16 || dt == at::kFloat) TORCH_CHECK(((uintptr_t)t.data_ptr() & 15) == 0, name, ": data pointer must be 16-byte aligned");
}
const void* opt_ptr(const optional<Tensor>& t, at::ScalarType dt, const char* name) {
  if (!t.has_value() || !t->defined()) return nullptr;
  chk(*t, dt, name);
  return t->data_ptr();
}
void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", lcc_last_error()); }
int64_t rows_of(const Tensor& x) { return x.numel() / x.size(-1); }
lcc_kv_layout layout_of(int64_t n_layers, int64_t n_kv_heads, int64_t lmax) {
  lcc_kv_layout l; l.n_layers = (int)n_layers; l.n_kv_heads = (int)n_kv_heads; l.lmax = (int)lmax; l.head_dim = 128;
  return l;
}
// the KV arena travels as (kv_ptrs int64 [n_slots] = device pointers of the per-slot arenas, kv_buf = the tensor that owns the memory, so
// that the dispatcher sees what an appending op mutates)
void* const* kv_base_of(const Tensor& kv_ptrs, const Tensor& kv_buf) {
  chk(kv_ptrs, at::kLong, "kv_ptrs");
  TORCH_CHECK(kv_buf.is_cuda() && kv_buf.scalar_type() == at::kBFloat16, "kv_buf: bf16 GPU tensor expected");
  return (void* const*)kv_ptrs.data_ptr();
}

// ---- Qwen2VLRMSNorm (Q2VL:96-110) / nn.LayerNorm (Q2VL:428-429) / SwiGLU product (Q2VL:465) ----
Tensor rmsnorm(const Tensor& x, const Tensor& w, double eps) {
  chk(x, at::kBFloat16, "x"); chk(w, at::kBFloat16, "w");
  TORCH_CHECK(x.dim() >= 1 && w.numel() == x.size(-1), "rmsnorm: weight has ", w.numel(), " elements, rows have ", x.size(-1));
  Tensor y = at::empty_like(x);
  ok(lcc_rmsnorm_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), (int)rows_of(x), (int)x.size(-1), (float)eps, cur_stream(x)), "lcc_rmsnorm_bf16");
  return y;
}
Tensor layernorm(const Tensor& x, const Tensor& w, const Tensor& b, double eps) {
  chk(x, at::kBFloat16, "x"); chk(w, at::kBFloat16, "w"); chk(b, at::kBFloat16, "b");
  TORCH_CHECK(x.dim() >= 1 && w.numel() == x.size(-1) && b.numel() == x.size(-1), "layernorm: weight / bias size mismatch");
  Tensor y = at::empty_like(x);
  ok(lcc_layernorm_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), (int)rows_of(x), (int)x.size(-1), (float)eps, cur_stream(x)),
     "lcc_layernorm_bf16");
  return y;
}
Tensor swiglu(const Tensor& gate, const Tensor& up) {
  chk(gate, at::kBFloat16, "gate"); chk(up, at::kBFloat16, "up");
  TORCH_CHECK(gate.sizes() == up.sizes(), "swiglu: gate and up must have the same shape");
  Tensor out = at::empty_like(gate);
  ok(lcc_swiglu_bf16(gate.data_ptr(), up.data_ptr(), out.data_ptr(), gate.numel(), cur_stream(gate)), "lcc_swiglu_bf16");
  return out;
}

// ---- nn.Linear with a fused epilogue (every Linear / Conv3d of the path; epilogue codes LCC_EPI_*).  packed_n / packed_k > 0: `w` is a
//      weight in the MFMA-fragment order of livecc_amd.ops.pack_weight for an [packed_n, packed_k] matrix ----
Tensor linear(const Tensor& x, const Tensor& w, const optional<Tensor>& bias, int64_t epilogue, const optional<Tensor>& residual,
              int64_t packed_n, int64_t packed_k) {
  chk(x, at::kBFloat16, "x"); chk(w, at::kBFloat16, "w");
  TORCH_CHECK(x.dim() == 2, "linear: x must be [M, K]");
  const bool packed = packed_n > 0;
  TORCH_CHECK(packed || w.dim() == 2, "linear: w must be [N, K] (or packed with packed_n / packed_k)");
  const int64_t M = x.size(0), K = x.size(1), N = packed ? packed_n : w.size(0);
  TORCH_CHECK((packed ? packed_k : w.size(1)) == K, "linear: K mismatch");
  TORCH_CHECK(!packed || w.numel() >= N * ((K + 31) / 32 * 32), "linear: packed weight too small");
  TORCH_CHECK(epilogue >= 0 && epilogue <= 4, "linear: epilogue must be one of LCC_EPI_* (0..4)");
  TORCH_CHECK(epilogue != LCC_EPI_RESIDUAL || (residual.has_value() && residual->defined()), "linear: the residual epilogue needs `residual`");
  if (residual.has_value() && residual->defined()) TORCH_CHECK(residual->dim() == 2 && residual->size(0) == M && residual->size(1) == N, "linear: residual must be [M, N]");
  if (bias.has_value() && bias->defined()) TORCH_CHECK(bias->numel() == N, "linear: bias must have N elements");
  const int64_t No = epilogue == LCC_EPI_SWIGLU ? N / 2 : N;
  Tensor out = at::empty({M, No}, x.options());
  ok(lcc_gemm_bf16(x.data_ptr(), (int)K, w.data_ptr(), (int)K, packed ? 1 : 0, opt_ptr(bias, at::kBFloat16, "bias"),
                   opt_ptr(residual, at::kBFloat16, "residual"), (int)N, out.data_ptr(), (int)No, (int)M, (int)N, (int)K, (int)epilogue, nullptr, 0,
                   cur_stream(x)), "lcc_gemm_bf16");
  return out;
}

// ---- bias + M-RoPE (Q2VL:180-222) + in-place Cache.update (cache_utils.py:127-146); returns the rotated q [S, Hq*128] ----
Tensor rope_kv_append(const optional<Tensor>& qkv, const optional<Tensor>& partial, const optional<Tensor>& bias, const Tensor& cos, const Tensor& sin,
                      const Tensor& tok_stream, const optional<Tensor>& tok_pos, const optional<Tensor>& kv_len, const Tensor& kv_ptrs, Tensor& kv_buf,
                      int64_t n_layers, int64_t n_kv_heads, int64_t lmax, int64_t layer, int64_t n_q_heads) {
  chk(cos, at::kBFloat16, "cos"); chk(sin, at::kBFloat16, "sin"); chk(tok_stream, at::kInt, "tok_stream");
  TORCH_CHECK(cos.dim() == 2 && cos.size(1) == 64 && sin.sizes() == cos.sizes(), "rope_kv_append: cos / sin must be [S, 64]");
  const int64_t S = cos.size(0);
  TORCH_CHECK(tok_stream.numel() == S, "rope_kv_append: tok_stream must have S entries");
  const bool have_qkv = qkv.has_value() && qkv->defined(), have_part = partial.has_value() && partial->defined();
  TORCH_CHECK(have_qkv != have_part, "rope_kv_append: exactly one of qkv (bf16 [S, (Hq+2Hkv)*128]) and partial (fp32 [NS, S, ...]) is needed");
  TORCH_CHECK(layer >= 0 && layer < n_layers && (lmax % 32) == 0 && n_q_heads >= 0, "rope_kv_append: bad layer / layout");
  const int64_t width = (n_q_heads + 2 * n_kv_heads) * 128;
  if (have_qkv) TORCH_CHECK(qkv->dim() == 2 && qkv->size(0) == S && qkv->size(1) == width, "rope_kv_append: qkv must be [S, (Hq+2Hkv)*128]");
  if (have_part) TORCH_CHECK(partial->dim() == 3 && partial->size(1) == S && partial->size(2) == width, "rope_kv_append: partial must be [NS, S, (Hq+2Hkv)*128]");
  Tensor q = at::empty({S, std::max<int64_t>(n_q_heads, 1) * 128}, cos.options());
  ok(lcc_rope_kv_append_bf16(opt_ptr(qkv, at::kBFloat16, "qkv"), (const float*)opt_ptr(partial, at::kFloat, "partial"), have_part ? (int)partial->size(0) : 0,
                             opt_ptr(bias, at::kBFloat16, "bias"), cos.data_ptr(), sin.data_ptr(), (const int32_t*)tok_stream.data_ptr(),
                             (const int32_t*)opt_ptr(tok_pos, at::kInt, "tok_pos"), (const int32_t*)opt_ptr(kv_len, at::kInt, "kv_len"),
                             kv_base_of(kv_ptrs, kv_buf), layout_of(n_layers, n_kv_heads, lmax), (int)layer, q.data_ptr(), (int)S, (int)n_q_heads,
                             cur_stream(cos)), "lcc_rope_kv_append_bf16");
  return n_q_heads > 0 ? q : q.narrow(1, 0, 0);
}

// ---- Qwen2VLAttention core (Q2VL:537-556): causal GQA over the in-place cache ----
Tensor attn_prefill(const Tensor& q, const Tensor& kv_ptrs, const Tensor& kv_buf, int64_t n_layers, int64_t n_kv_heads, int64_t lmax, int64_t layer,
                    const Tensor& tile_stream, const Tensor& tile_q0, const Tensor& tile_nq, const Tensor& tile_pos0, int64_t n_q_heads,
                    int64_t tile_rows, int64_t nsplit) {
  chk(q, at::kBFloat16, "q");
  chk(tile_stream, at::kInt, "tile_stream"); chk(tile_q0, at::kInt, "tile_q0"); chk(tile_nq, at::kInt, "tile_nq"); chk(tile_pos0, at::kInt, "tile_pos0");
  TORCH_CHECK(q.dim() == 2 && q.size(1) == n_q_heads * 128, "attn_prefill: q must be [S, Hq*128]");
  const int64_t nt = tile_stream.numel();
  TORCH_CHECK(tile_q0.numel() == nt && tile_nq.numel() == nt && tile_pos0.numel() == nt, "attn_prefill: the four tile tables must have one entry per tile");
  TORCH_CHECK(tile_rows >= 16 && tile_rows <= 64 && nsplit >= 1 && nsplit <= 8 && layer >= 0 && layer < n_layers, "attn_prefill: bad tile_rows / nsplit / layer");
  Tensor out = at::empty_like(q), ws_o, ws_ml;
  if (nsplit > 1) {
    ws_o = at::empty({q.size(0) * n_q_heads * nsplit * 128}, q.options().dtype(at::kFloat));
    ws_ml = at::empty({q.size(0) * n_q_heads * nsplit * 2}, q.options().dtype(at::kFloat));
  }
  ok(lcc_attn_prefill_bf16(q.data_ptr(), out.data_ptr(), (const int32_t*)tile_stream.data_ptr(), (const int32_t*)tile_q0.data_ptr(),
                           (const int32_t*)tile_nq.data_ptr(), (const int32_t*)tile_pos0.data_ptr(), kv_base_of(kv_ptrs, kv_buf),
                           layout_of(n_layers, n_kv_heads, lmax), (int)layer, (int)nt, (int)n_q_heads, (int)tile_rows, (int)nsplit, (int)q.size(0),
                           nsplit > 1 ? (float*)ws_o.data_ptr() : nullptr, nsplit > 1 ? (float*)ws_ml.data_ptr() : nullptr, cur_stream(q)),
     "lcc_attn_prefill_bf16");
  return out;
}
Tensor attn_decode(const Tensor& q, const Tensor& kv_ptrs, const Tensor& kv_buf, int64_t n_layers, int64_t n_kv_heads, int64_t lmax, int64_t layer,
                   const Tensor& slots, const Tensor& kv_len, int64_t n_q_heads, int64_t nsplit) {
  chk(q, at::kBFloat16, "q"); chk(slots, at::kInt, "slots"); chk(kv_len, at::kInt, "kv_len");
  TORCH_CHECK(q.dim() == 2 && q.size(1) == n_q_heads * 128 && slots.numel() == q.size(0), "attn_decode: q must be [B, Hq*128] with one slot per row");
  TORCH_CHECK(nsplit >= 1 && nsplit <= 128 && layer >= 0 && layer < n_layers, "attn_decode: bad nsplit / layer");
  const int64_t B = q.size(0);
  Tensor out = at::empty_like(q);
  Tensor ws_o = at::empty({B * n_kv_heads * nsplit * 16 * 128}, q.options().dtype(at::kFloat));
  Tensor ws_ml = at::empty({B * n_kv_heads * nsplit * 16 * 2}, q.options().dtype(at::kFloat));
  ok(lcc_attn_decode_bf16(q.data_ptr(), out.data_ptr(), (const int32_t*)slots.data_ptr(), (const int32_t*)kv_len.data_ptr(), kv_base_of(kv_ptrs, kv_buf),
                          layout_of(n_layers, n_kv_heads, lmax), (int)layer, (int)B, (int)n_q_heads, (int)nsplit, (float*)ws_o.data_ptr(),
                          (float*)ws_ml.data_ptr(), cur_stream(q)), "lcc_attn_decode_bf16");
  return out;
}

// ---- HF video processor: rescale + normalise + patchify (video_processing_qwen2_vl.py:236-274); layout 0 = [T,H,W,3], 1 = [T,3,H,W] ----
Tensor patchify_norm(const Tensor& frames, int64_t layout, c10::ArrayRef<double> mean255, c10::ArrayRef<double> std255) {
  chk(frames, at::kByte, "frames");
  TORCH_CHECK(frames.dim() == 4 && (layout == 0 || layout == 1) && mean255.size() == 3 && std255.size() == 3, "patchify_norm: uint8 [T,H,W,3] / [T,3,H,W], 3 means, 3 stds");
  const int64_t T = frames.size(0), H = layout == 0 ? frames.size(1) : frames.size(2), W = layout == 0 ? frames.size(2) : frames.size(3);
  TORCH_CHECK((layout == 0 ? frames.size(3) : frames.size(1)) == 3 && H % 28 == 0 && W % 28 == 0, "patchify_norm: 3 channels, H and W multiples of 28");
  const float m[3] = {(float)mean255[0], (float)mean255[1], (float)mean255[2]}, s[3] = {(float)std255[0], (float)std255[1], (float)std255[2]};
  Tensor out = at::empty({((T + 1) / 2) * (H / 14) * (W / 14), 1176}, frames.options().dtype(at::kBFloat16));
  ok(lcc_patchify_norm_u8((const uint8_t*)frames.data_ptr(), (int)layout, (int)T, (int)H, (int)W, m, s, out.data_ptr(), 1176, cur_stream(frames)),
     "lcc_patchify_norm_u8");
  return out;
}

// ---- torchvision resize(uint8, BICUBIC, antialias=True) (ref video_process_patch.py:150-155); tap tables from livecc_amd.resize ----
Tensor resize_bicubic_aa(const Tensor& frames, int64_t layout, int64_t Hout, int64_t Wout, const Tensor& xmin, const Tensor& xsize, const Tensor& wx,
                         int64_t kx, const Tensor& ymin, const Tensor& ysize, const Tensor& wy, int64_t ky) {
  chk(frames, at::kByte, "frames"); chk(xmin, at::kInt, "xmin"); chk(xsize, at::kInt, "xsize"); chk(wx, at::kFloat, "wx");
  chk(ymin, at::kInt, "ymin"); chk(ysize, at::kInt, "ysize"); chk(wy, at::kFloat, "wy");
  TORCH_CHECK(frames.dim() == 4 && (layout == 0 || layout == 1), "resize_bicubic_aa: uint8 [T,H,W,3] / [T,3,H,W]");
  const int64_t T = frames.size(0), Hin = layout == 0 ? frames.size(1) : frames.size(2), Win = layout == 0 ? frames.size(2) : frames.size(3);
  TORCH_CHECK((layout == 0 ? frames.size(3) : frames.size(1)) == 3, "resize_bicubic_aa: 3 channels expected");
  TORCH_CHECK(xmin.numel() == Wout && xsize.numel() == Wout && wx.numel() == kx * Wout && ymin.numel() == Hout && ysize.numel() == Hout && wy.numel() == ky * Hout,
              "resize_bicubic_aa: tap tables do not match the output size");
  Tensor out = at::empty({T, 3, Hout, Wout}, frames.options());
  Tensor tmp = at::empty({T * 3 * Hin * Wout}, frames.options().dtype(at::kFloat));
  ok(lcc_resize_bicubic_aa_u8((const uint8_t*)frames.data_ptr(), (int)layout, (int)T, (int)Hin, (int)Win, (uint8_t*)out.data_ptr(), (int)Hout, (int)Wout,
                              (const int32_t*)xmin.data_ptr(), (const int32_t*)xsize.data_ptr(), (const float*)wx.data_ptr(), (int)kx,
                              (const int32_t*)ymin.data_ptr(), (const int32_t*)ysize.data_ptr(), (const float*)wy.data_ptr(), (int)ky,
                              (float*)tmp.data_ptr(), cur_stream(frames)), "lcc_resize_bicubic_aa_u8");
  return out;
}

// ---- RepetitionPenalty -> [MinNewTokens EOS mask] -> ThresholdLogitsProcessor -> argmax (ref demo/infer.py:10-23); returns
//      (tokens int32 [n_slots], processed scores fp32 [B, V] or an empty tensor) ----
std::tuple<Tensor, Tensor> sample_greedy(const Tensor& logits, const Tensor& seen, const Tensor& slots, double repetition_penalty, int64_t thr_token,
                                         bool use_thr, double thr_value, int64_t eos_token, int64_t eos_token2, bool suppress_eos, bool want_scores) {
  chk(logits, at::kBFloat16, "logits"); chk(seen, at::kInt, "seen"); chk(slots, at::kInt, "slots");
  TORCH_CHECK(logits.dim() == 2 && seen.dim() == 2 && slots.numel() == logits.size(0), "sample_greedy: logits [B, V], seen [n_slots, V/32], one slot per row");
  const int64_t B = logits.size(0), V = logits.size(1);
  TORCH_CHECK(seen.size(1) * 32 >= V && (V % 32) == 0, "sample_greedy: V % 32 == 0 and a seen bitmap of V/32 words per slot");
  Tensor out = at::zeros({seen.size(0)}, logits.options().dtype(at::kInt));
  Tensor scores = want_scores ? at::empty({B, V}, logits.options().dtype(at::kFloat)) : at::empty({0}, logits.options().dtype(at::kFloat));
  ok(lcc_sample_greedy(logits.data_ptr(), (int)V, (int)B, (int)V, (uint32_t*)seen.data_ptr(), (int)seen.size(1), (const int32_t*)slots.data_ptr(),
                       (float)repetition_penalty, (int)thr_token, use_thr ? 1 : 0, (float)thr_value, (int)eos_token, (int)eos_token2, suppress_eos ? 1 : 0,
                       nullptr, (int32_t*)out.data_ptr(), nullptr, 0, nullptr, want_scores ? (float*)scores.data_ptr() : nullptr, nullptr,
                       cur_stream(logits)), "lcc_sample_greedy");
  return {out, scores};
}

}  // namespace

TORCH_LIBRARY(livecc_amd, m) {
  m.def("rmsnorm(Tensor x, Tensor w, float eps) -> Tensor");
  m.def("layernorm(Tensor x, Tensor w, Tensor b, float eps) -> Tensor");
  m.def("swiglu(Tensor gate, Tensor up) -> Tensor");
  m.def("linear(Tensor x, Tensor w, Tensor? bias, int epilogue, Tensor? residual, int packed_n, int packed_k) -> Tensor");
  m.def("rope_kv_append(Tensor? qkv, Tensor? partial, Tensor? bias, Tensor cos, Tensor sin, Tensor tok_stream, Tensor? tok_pos, Tensor? kv_len, "
        "Tensor kv_ptrs, Tensor(a!) kv_buf, int n_layers, int n_kv_heads, int lmax, int layer, int n_q_heads) -> Tensor");
  m.def("attn_prefill(Tensor q, Tensor kv_ptrs, Tensor kv_buf, int n_layers, int n_kv_heads, int lmax, int layer, Tensor tile_stream, Tensor tile_q0, "
        "Tensor tile_nq, Tensor tile_pos0, int n_q_heads, int tile_rows, int nsplit) -> Tensor");
  m.def("attn_decode(Tensor q, Tensor kv_ptrs, Tensor kv_buf, int n_layers, int n_kv_heads, int lmax, int layer, Tensor slots, Tensor kv_len, "
        "int n_q_heads, int nsplit) -> Tensor");
  m.def("patchify_norm(Tensor frames, int layout, float[] mean255, float[] std255) -> Tensor");
  m.def("resize_bicubic_aa(Tensor frames, int layout, int Hout, int Wout, Tensor xmin, Tensor xsize, Tensor wx, int kx, Tensor ymin, Tensor ysize, "
        "Tensor wy, int ky) -> Tensor");
  m.def("sample_greedy(Tensor logits, Tensor(a!) seen, Tensor slots, float repetition_penalty, int thr_token, bool use_thr, float thr_value, "
        "int eos_token, int eos_token2, bool suppress_eos, bool want_scores) -> (Tensor, Tensor)");
}

// device key "CUDA" is the HIP device key of PyTorch-ROCm (c10::DispatchKey::CUDA == HIP builds' GPU key)
TORCH_LIBRARY_IMPL(livecc_amd, CUDA, m) {
  m.impl("rmsnorm", &rmsnorm);
  m.impl("layernorm", &layernorm);
  m.impl("swiglu", &swiglu);
  m.impl("linear", &linear);
  m.impl("rope_kv_append", &rope_kv_append);
  m.impl("attn_prefill", &attn_prefill);
  m.impl("attn_decode", &attn_decode);
  m.impl("patchify_norm", &patchify_norm);
  m.impl("resize_bicubic_aa", &resize_bicubic_aa);
  m.impl("sample_greedy", &sample_greedy);
}
