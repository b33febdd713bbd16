// Model-level engine: the forward/generate arithmetic of `Qwen2VLForConditionalGeneration` (HF
// modeling_qwen2_vl.py: visual.forward 700-729, Qwen2VLModel.forward 1144-1204, Qwen2VLTextModel.forward 762-844,
// lm_head 1320-1323, GenerationMixin._sample utils.py:2783-2960) as one C++ launch sequence per call over
// the HIP kernels, with every per-stream state (KV, lengths, rope position, seen-id bitmap, generated ids)
// resident in HBM.  Host code only enqueues; the decode loop runs n steps without a host round trip.
//
// Memory comes from the caller (PyTorch allocations): weights, one KV arena per stream slot, an activation
// workspace, a small device state block, and a pinned-host + device "meta" ring through which the per-call
// integer tables (ids, positions, tile tables) travel in ONE async copy per call.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/livecc_amd.h"
#include "kernels.h"
#include "grid_sync.h"

using namespace lcc;

// 1: decode pipeline v2 launches down_proj(l) + q/k/v(l+1) as ONE chained launch where both grids fit the chip at once (decode_v2.hip).
// Measured on MI355X (LiveCC-7B, one stream, no ViT prefetch; profiles/r03/decode_chain_ab.jsonl): bit-identical, but SLOWER --
// 3105-3229 us per decode step against 2989 us for the two launches (the chained kernel 43.6 us vs 25.5 + 10.7 us).  The consumer's
// 33 MB of weights are served at the START of the launch (total HBM bytes are the same), and what the hand-off then exposes after the
// producer is the consumer's whole serial tail (flag -> statistics -> normalise -> LDS -> 14 MFMAs -> reduce -> RoPE epilogue, ~5 us)
// that a stand-alone launch hides under its own weight stream -- as much as the removed kernel boundary was worth.  Kept as a tested
// variant (lcc_debug_set_decode_chain(1)); default off.
static int g_decode_chain = 0;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(x)                                                                          \
  do {                                                                                      \
    hipError_t e__ = (x);                                                                   \
    if (e__ != hipSuccess) return fail(LCC_ERR_HIP, "%s: %s", #x, hipGetErrorString(e__)); \
  } while (0)
#define LCC_TRY(x)                                                            \
  do {                                                                        \
    int r__ = (x);                                                            \
    if (r__ != 0) {                                                           \
      if (g_err[0] == 0 || r__ != LCC_ERR_HIP) fail(r__, "%s failed (%d)", #x, r__); \
      return r__;                                                             \
    }                                                                         \
  } while (0)
static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LCC_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

extern "C" const char* lcc_last_error(void) { return g_err; }
extern "C" const char* lcc_version(void) { return "livecc_amd 0.1.0 (gfx950)"; }
extern "C" int lcc_device_info(int* cu_count, size_t* hbm_bytes, char* arch, int arch_len) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, dev));
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", p.gcnArchName);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
namespace {

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct VitLayerW { const bf16_t *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b; };
struct LlmLayerW {
  const bf16_t *in_norm, *qkv_w, *qkv_b, *o_w, *post_norm, *gate_up_w, *down_w;
  const bf16_t* qkv_w_dec;   // optional row-permuted decode copy of qkv_w (decode pipeline v2), nullptr when absent
  const float *qkv_s, *o_s, *gate_up_s, *down_s;   // fp8 weights: per-output-row scales (nullptr for bf16 weights)
  const float* qkv_s_dec;                          // scales of the row-permuted decode copy (fp8 arenas with decode copies)
};

struct Carver {  // bump allocator over a caller-provided region
  char* base = nullptr;
  size_t off = 0;
  template <class T>
  T* take(size_t n) {
    T* p = reinterpret_cast<T*>(base + off);
    off = align_up(off + n * sizeof(T));
    return p;
  }
};

constexpr int META_RING = 4;
constexpr int MAX_SPLIT = 8;

}  // namespace

struct lcc_engine {
  lcc_model_config c;
  lcc_engine_limits lim;
  int qd, kvd, qkvd, words, E, vit_hd;
  int cu_count = 256;          // compute units of the device current at lcc_engine_create
  KvLayout lay;

  // weights
  std::map<std::string, const void*> w;
  std::vector<VitLayerW> vit;
  std::vector<LlmLayerW> llm;
  const bf16_t *patch_embed = nullptr, *mg_ln_w = nullptr, *mg_ln_b = nullptr, *mg_fc1_w = nullptr, *mg_fc1_b = nullptr,
               *mg_fc2_w = nullptr, *mg_fc2_b = nullptr, *embed = nullptr, *final_norm = nullptr, *lm_head = nullptr;
  const float* inv_freq = nullptr;
  const float* lm_head_s = nullptr;
  bool weights_resolved = false;

  // buffers
  char* ws = nullptr; size_t ws_bytes = 0;
  char* state = nullptr; size_t state_bytes = 0;
  char *meta_dev = nullptr, *meta_host = nullptr; size_t meta_bytes = 0, meta_slot_bytes = 0;
  int meta_next = 0;
  hipEvent_t meta_ev[META_RING] = {};
  bool meta_ev_used[META_RING] = {};
  // optional private workspace + meta ring of the ViT, so that lcc_vit_encode may run on a SECOND stream concurrently with the LLM
  // (the next turn's frames are encoded under the current turn's decode steps); slot events are recorded after the LAST ViT kernel
  char* ws_vit = nullptr; size_t ws_vit_bytes = 0;
  char *vmeta_dev = nullptr, *vmeta_host = nullptr; size_t vmeta_slot_bytes = 0;
  int vmeta_next = 0;
  hipEvent_t vmeta_ev[2] = {};
  bool vmeta_ev_used[2] = {};

  // device state (inside `state`)
  int32_t *d_kv_len = nullptr, *d_pos = nullptr, *d_hist_col = nullptr, *d_cur_tok = nullptr, *d_done = nullptr, *d_history = nullptr;
  int32_t* d_counter = nullptr;   // arrival counter of the fused GEMV tails (zero between launches)
  int32_t* d_attn_cnt = nullptr;  // [16 streams x Hkv] arrival counters of the fused decode attention (zero between launches)
  uint32_t* d_seen = nullptr;
  uint32_t* d_rng_ctr = nullptr;  // per-slot Philox draw counter of the sampling kernel (zero for a fresh stream)
  unsigned* d_chain = nullptr;    // [128] monotonic hand-off counters of the chained decode launches (one per layer) + [128] = error word
  std::vector<unsigned> chain_epoch;   // host mirror: launches issued per counter (the consumer's target = epoch * producer blocks)
  bf16_t** d_kv_base = nullptr;
  // optional live timing of the dominant kernel (decode gate/up GEMV): hipEvent pairs on the launch stream
  std::vector<hipEvent_t> prof_ev;   // 2 * capacity
  int prof_n = 0; bool prof_on = false;
  std::vector<hipEvent_t> step_ev;   // whole decode steps (layers + lm_head + sampler), 2 * capacity
  std::vector<int> step_rel;         // index of each sampled step inside its lcc_llm_decode call (0 = right after the prefill)
  int step_n = 0;
  // parity instrumentation (lcc_debug_set_llm_taps / lcc_debug_set_vit_taps): residual-stream taps and per-layer input overrides
  bf16_t* llm_taps = nullptr; const bf16_t* llm_over = nullptr; int llm_tap_rows = 0;
  bf16_t* vit_taps = nullptr; const bf16_t* vit_over = nullptr; int vit_tap_rows = 0;
  const int32_t* forced = nullptr; int forced_steps = 0, forced_B = 0;   // teacher forcing (lcc_debug_set_forced_tokens)
  // host mirrors
  std::vector<int> h_kv_len, h_pos;
  std::vector<void*> h_kv_base;

  size_t llm_ws_bytes() const;
  size_t vit_ws_bytes() const;
};

size_t lcc_engine::llm_ws_bytes() const {
  const size_t S = lim.max_new_rows, B = lim.max_slots;
  const size_t H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size;
  size_t t = 0;
  t += align_up(S * H * 2) * 2;              // h, xn
  t += align_up(S * qkvd * 2);               // qkv
  t += align_up(S * qd * 2) * 2;             // q, attn
  t += align_up(S * I * 2);                  // act
  t += align_up(S * 64 * 2) * 2;             // cos, sin
  t += align_up(std::max((size_t)MAX_SPLIT * 16 * std::max<size_t>(qkvd, H), (size_t)4 * std::min<size_t>(S, 4096) * H) * 4);  // split-K slabs
  t += align_up(B * H * 2) * 2;              // last_h, last_xn
  t += align_up(16 * (H / 16 + 4) * 4);      // decode v2: per-tile sums of squares of the residual rows
  t += align_up(B * V * 2);                  // logits
  t += align_up(std::max<size_t>(B * c.n_kv_heads * 128 * 16, std::min<size_t>(S, 1024) * c.n_q_heads * 8) * 128 * 4) * 2;  // attention split partials (o, ml)
  if (c.llm_fp8) t += align_up(std::max<size_t>((size_t)qkvd * H, 2 * I * H) * 2);   // bf16 dequantisation scratch of the largest LLM weight
  return t + 4096;
}
size_t lcc_engine::vit_ws_bytes() const {
  const size_t P = lim.max_patches, Ev = c.vit_embed;
  const size_t blocks = P / 32 + 2 * (P / 64 + 1) + 64;  // generous: every segment rounds up to a 32-key block
  size_t t = 0;
  t += align_up(P * (size_t)c.patch_dim * 2);  // patches
  t += align_up(P * Ev * 2) * 3;               // x, xn, attn
  t += align_up(P * 3 * Ev * 2);               // qkv
  t += align_up(P * (size_t)c.vit_mlp * 2);    // mlp
  t += align_up((size_t)c.vit_heads * blocks * 80 * 32 * 2);  // vt
  t += align_up(P / 4 * 4 * Ev * 2 + 256);     // merger hidden
  return t + 4096;
}

extern "C" lcc_engine* lcc_engine_create(const lcc_model_config* cfg, const lcc_engine_limits* lim) {
  if (!cfg || !lim) { fail(LCC_ERR_ARG, "null config"); return nullptr; }
  if (cfg->head_dim != 128 || cfg->vit_embed / cfg->vit_heads != 80 || cfg->vit_embed % cfg->vit_heads) {
    fail(LCC_ERR_SHAPE, "head_dim must be 128 (LLM) and 80 (ViT)"); return nullptr;
  }
  if (cfg->n_q_heads % cfg->n_kv_heads || cfg->n_q_heads / cfg->n_kv_heads > 16) { fail(LCC_ERR_SHAPE, "GQA group must be <= 16"); return nullptr; }
  if ((lim->max_kv_len & 31) || lim->max_slots <= 0 || lim->max_slots > 16 * 1024) { fail(LCC_ERR_SHAPE, "max_kv_len %% 32, max_slots"); return nullptr; }
  if ((cfg->vocab_size & 31) || (cfg->hidden_size & 15) || (cfg->intermediate_size & 15) || cfg->patch_dim != 1176 || cfg->merge != 2) {
    fail(LCC_ERR_SHAPE, "vocab %% 32, hidden/intermediate %% 16, patch_dim 1176, merge 2"); return nullptr;
  }
  if (cfg->mrope_sec_t + cfg->mrope_sec_h + cfg->mrope_sec_w != 64) { fail(LCC_ERR_SHAPE, "mrope sections must sum to 64"); return nullptr; }
  lcc_engine* e = new lcc_engine();
  e->c = *cfg; e->lim = *lim;
  e->qd = cfg->n_q_heads * 128; e->kvd = cfg->n_kv_heads * 128; e->qkvd = e->qd + 2 * e->kvd;
  e->words = cfg->vocab_size / 32; e->E = cfg->vit_embed; e->vit_hd = 80;
  e->lay = KvLayout{cfg->n_layers, cfg->n_kv_heads, lim->max_kv_len, 128};
  e->vit.resize(cfg->vit_depth); e->llm.resize(cfg->n_layers);
  e->chain_epoch.assign(128, 0u);
  { int dev = 0; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&e->cu_count, hipDeviceAttributeMultiprocessorCount, dev); }
  if (e->cu_count <= 0) e->cu_count = 256;
  e->h_kv_len.assign(lim->max_slots, 0); e->h_pos.assign(lim->max_slots, 0); e->h_kv_base.assign(lim->max_slots, nullptr);
  return e;
}
extern "C" void lcc_engine_destroy(lcc_engine* e) {
  if (!e) return;
  for (int i = 0; i < META_RING; ++i) if (e->meta_ev[i]) (void)hipEventDestroy(e->meta_ev[i]);
  for (int i = 0; i < 2; ++i) if (e->vmeta_ev[i]) (void)hipEventDestroy(e->vmeta_ev[i]);
  for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->step_ev) (void)hipEventDestroy(ev);
  delete e;
}
extern "C" int lcc_engine_profile(lcc_engine* e, int enable, int max_samples) {
  if (!e) return fail(LCC_ERR_ARG, "null engine");
  if (enable) {
    while ((int)e->prof_ev.size() < 2 * max_samples) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      e->prof_ev.push_back(ev);
    }
    while ((int)e->step_ev.size() < 2 * max_samples) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      e->step_ev.push_back(ev);
    }
    e->prof_n = 0; e->step_n = 0;
    e->step_rel.assign(max_samples, 0);
  }
  e->prof_on = enable != 0;
  return 0;
}
extern "C" int lcc_engine_profile_read(lcc_engine* e, float* ms_out, int max_n, int* n_out) {
  if (!e || !ms_out || !n_out) return fail(LCC_ERR_ARG, "null argument");
  const int n = std::min(e->prof_n, max_n);
  for (int i = 0; i < n; ++i) {
    HIP_TRY(hipEventSynchronize(e->prof_ev[2 * i + 1]));
    HIP_TRY(hipEventElapsedTime(&ms_out[i], e->prof_ev[2 * i], e->prof_ev[2 * i + 1]));
  }
  *n_out = n;
  return 0;
}
extern "C" int lcc_engine_profile_read_steps(lcc_engine* e, float* ms_out, int max_n, int* n_out) {
  if (!e || !ms_out || !n_out) return fail(LCC_ERR_ARG, "null argument");
  const int n = std::min(e->step_n, max_n);
  for (int i = 0; i < n; ++i) {
    HIP_TRY(hipEventSynchronize(e->step_ev[2 * i + 1]));
    HIP_TRY(hipEventElapsedTime(&ms_out[i], e->step_ev[2 * i], e->step_ev[2 * i + 1]));
  }
  *n_out = n;
  return 0;
}
extern "C" int lcc_engine_profile_read_step_index(lcc_engine* e, int32_t* idx_out, int max_n, int* n_out) {
  if (!e || !idx_out || !n_out) return fail(LCC_ERR_ARG, "null argument");
  const int n = std::min(std::min(e->step_n, max_n), (int)e->step_rel.size());
  for (int i = 0; i < n; ++i) idx_out[i] = e->step_rel[i];
  *n_out = n;
  return 0;
}
extern "C" size_t lcc_engine_workspace_bytes(const lcc_engine* e) { return std::max(e->llm_ws_bytes(), e->vit_ws_bytes()); }
extern "C" size_t lcc_engine_state_bytes(const lcc_engine* e) {
  const size_t B = e->lim.max_slots;
  return align_up(B * 4) * 6 + 256 + align_up(B * 8) + align_up(B * (size_t)e->lim.max_history * 4) + align_up(B * (size_t)e->words * 4) + 4096 + 1024 + 1024;
}
extern "C" size_t lcc_engine_kv_bytes_per_slot(const lcc_engine* e) { return e->lay.total() * 2; }
extern "C" size_t lcc_engine_meta_bytes(const lcc_engine* e) {
  const size_t S = e->lim.max_new_rows, P = e->lim.max_patches, B = e->lim.max_slots;
  const size_t llm = (7 * S + 4 * (S / 16 + B + 1) + 4 * B + 64) * 4;
  const size_t vit = (P + 8 * (P / 16 + 64) + 64) * 4;
  return align_up(std::max(llm, vit) + 4096, 4096) * META_RING;
}

extern "C" int lcc_engine_bind_buffers(lcc_engine* e, void* workspace_dev, size_t ws_bytes, void* state_dev, size_t state_bytes,
                                       void* meta_dev, void* meta_host_pinned, size_t meta_bytes) {
  if (!e || !workspace_dev || !state_dev || !meta_dev || !meta_host_pinned) return fail(LCC_ERR_ARG, "null buffer");
  if (ws_bytes < lcc_engine_workspace_bytes(e) || state_bytes < lcc_engine_state_bytes(e) || meta_bytes < lcc_engine_meta_bytes(e))
    return fail(LCC_ERR_STATE, "buffer too small: ws %zu/%zu state %zu/%zu meta %zu/%zu", ws_bytes, lcc_engine_workspace_bytes(e),
                state_bytes, lcc_engine_state_bytes(e), meta_bytes, lcc_engine_meta_bytes(e));
  if (((uintptr_t)workspace_dev | (uintptr_t)state_dev | (uintptr_t)meta_dev) & 255) return fail(LCC_ERR_ALIGN, "buffers must be 256-byte aligned");
  e->ws = (char*)workspace_dev; e->ws_bytes = ws_bytes;
  e->state = (char*)state_dev; e->state_bytes = state_bytes;
  e->meta_dev = (char*)meta_dev; e->meta_host = (char*)meta_host_pinned; e->meta_bytes = meta_bytes;
  e->meta_slot_bytes = lcc_engine_meta_bytes(e) / META_RING;
  const size_t B = e->lim.max_slots;
  Carver cv; cv.base = e->state;
  e->d_kv_len = cv.take<int32_t>(B); e->d_pos = cv.take<int32_t>(B); e->d_hist_col = cv.take<int32_t>(B);
  e->d_cur_tok = cv.take<int32_t>(B); e->d_done = cv.take<int32_t>(B); e->d_rng_ctr = cv.take<uint32_t>(B); e->d_counter = cv.take<int32_t>(16); e->d_attn_cnt = cv.take<int32_t>(256); e->d_chain = cv.take<unsigned>(160); e->d_kv_base = cv.take<bf16_t*>(B);
  e->d_history = cv.take<int32_t>(B * (size_t)e->lim.max_history);
  e->d_seen = cv.take<uint32_t>(B * (size_t)e->words);
  HIP_TRY(hipMemset(e->state, 0, state_bytes));
  std::fill(e->chain_epoch.begin(), e->chain_epoch.end(), 0u);   // the device hand-off counters were just zeroed: host mirror follows
  for (int i = 0; i < META_RING; ++i) if (!e->meta_ev[i]) HIP_TRY(hipEventCreateWithFlags(&e->meta_ev[i], hipEventDisableTiming));
  return 0;
}

extern "C" size_t lcc_engine_vit_workspace_bytes(const lcc_engine* e) { return e->vit_ws_bytes(); }
extern "C" size_t lcc_engine_vit_meta_bytes(const lcc_engine* e) {
  const size_t P = e->lim.max_patches;
  return align_up((P + 8 * (P / 16 + 64) + 64) * 4 + 4096, 4096) * 2;
}
extern "C" int lcc_engine_bind_vit_buffers(lcc_engine* e, void* workspace_dev, size_t ws_bytes, void* meta_dev, void* meta_host_pinned,
                                           size_t meta_bytes) {
  if (!e || !workspace_dev || !meta_dev || !meta_host_pinned) return fail(LCC_ERR_ARG, "null buffer");
  if (ws_bytes < e->vit_ws_bytes() || meta_bytes < lcc_engine_vit_meta_bytes(e))
    return fail(LCC_ERR_STATE, "ViT buffer too small: ws %zu/%zu meta %zu/%zu", ws_bytes, e->vit_ws_bytes(), meta_bytes, lcc_engine_vit_meta_bytes(e));
  if (((uintptr_t)workspace_dev | (uintptr_t)meta_dev) & 255) return fail(LCC_ERR_ALIGN, "buffers must be 256-byte aligned");
  e->ws_vit = (char*)workspace_dev; e->ws_vit_bytes = ws_bytes;
  e->vmeta_dev = (char*)meta_dev; e->vmeta_host = (char*)meta_host_pinned; e->vmeta_slot_bytes = lcc_engine_vit_meta_bytes(e) / 2;
  for (int i = 0; i < 2; ++i) if (!e->vmeta_ev[i]) HIP_TRY(hipEventCreateWithFlags(&e->vmeta_ev[i], hipEventDisableTiming));
  return 0;
}

extern "C" int lcc_engine_bind_kv(lcc_engine* e, int slot, void* kv_dev, size_t bytes) {
  if (!e || slot < 0 || slot >= e->lim.max_slots || !kv_dev) return fail(LCC_ERR_ARG, "bad slot/pointer");
  if (!e->state) return fail(LCC_ERR_STATE, "bind_buffers first");
  if (bytes < lcc_engine_kv_bytes_per_slot(e)) return fail(LCC_ERR_STATE, "kv arena too small: %zu < %zu", bytes, lcc_engine_kv_bytes_per_slot(e));
  if ((uintptr_t)kv_dev & 255) return fail(LCC_ERR_ALIGN, "kv arena must be 256-byte aligned");
  e->h_kv_base[slot] = kv_dev;
  HIP_TRY(hipMemcpy(e->d_kv_base + slot, &kv_dev, sizeof(void*), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int lcc_engine_set_weight(lcc_engine* e, const char* name, const void* dev, int64_t numel) {
  if (!e || !name || !dev) return fail(LCC_ERR_ARG, "null");
  if ((uintptr_t)dev & 15) return fail(LCC_ERR_ALIGN, "weight %s not 16-byte aligned", name);
  (void)numel;
  e->w[name] = dev;
  e->weights_resolved = false;
  return 0;
}

static int resolve_weights(lcc_engine* e, std::string* missing) {
  auto get = [&](const std::string& n) -> const bf16_t* {
    auto it = e->w.find(n);
    if (it == e->w.end()) { if (missing) { *missing += n; *missing += ' '; } return nullptr; }
    return (const bf16_t*)it->second;
  };
  e->patch_embed = get("vit.patch_embed");
  for (int i = 0; i < e->c.vit_depth; ++i) {
    const std::string p = "vit." + std::to_string(i) + ".";
    VitLayerW& L = e->vit[i];
    L.ln1_w = get(p + "ln1_w"); L.ln1_b = get(p + "ln1_b"); L.qkv_w = get(p + "qkv_w"); L.qkv_b = get(p + "qkv_b");
    L.proj_w = get(p + "proj_w"); L.proj_b = get(p + "proj_b"); L.ln2_w = get(p + "ln2_w"); L.ln2_b = get(p + "ln2_b");
    L.fc1_w = get(p + "fc1_w"); L.fc1_b = get(p + "fc1_b"); L.fc2_w = get(p + "fc2_w"); L.fc2_b = get(p + "fc2_b");
  }
  e->mg_ln_w = get("merger.ln_w"); e->mg_ln_b = get("merger.ln_b"); e->mg_fc1_w = get("merger.fc1_w");
  e->mg_fc1_b = get("merger.fc1_b"); e->mg_fc2_w = get("merger.fc2_w"); e->mg_fc2_b = get("merger.fc2_b");
  e->embed = get("embed");
  for (int i = 0; i < e->c.n_layers; ++i) {
    const std::string p = "llm." + std::to_string(i) + ".";
    LlmLayerW& L = e->llm[i];
    L.in_norm = get(p + "in_norm"); L.qkv_w = get(p + "qkv_w"); L.qkv_b = get(p + "qkv_b"); L.o_w = get(p + "o_w");
    L.post_norm = get(p + "post_norm"); L.gate_up_w = get(p + "gate_up_w"); L.down_w = get(p + "down_w");
    { auto it = e->w.find(p + "qkv_w_dec"); L.qkv_w_dec = it == e->w.end() ? nullptr : (const bf16_t*)it->second; }
    L.qkv_s = L.o_s = L.gate_up_s = L.down_s = L.qkv_s_dec = nullptr;
    if (e->c.llm_fp8) {
      { auto it = e->w.find(p + "qkv_w_dec.scale"); L.qkv_s_dec = it == e->w.end() ? nullptr : (const float*)it->second; }
      L.qkv_s = (const float*)get(p + "qkv_w.scale"); L.o_s = (const float*)get(p + "o_w.scale");
      L.gate_up_s = (const float*)get(p + "gate_up_w.scale"); L.down_s = (const float*)get(p + "down_w.scale");
    }
  }
  e->final_norm = get("final_norm"); e->lm_head = get("lm_head");
  e->lm_head_s = e->c.llm_fp8 ? (const float*)get("lm_head.scale") : nullptr;
  e->inv_freq = (const float*)get("inv_freq");
  if (missing && !missing->empty()) return LCC_ERR_STATE;
  e->weights_resolved = true;
  return 0;
}
extern "C" int lcc_engine_weights_ready(const lcc_engine* e, char* missing, int missing_len) {
  std::string m;
  int r = resolve_weights(const_cast<lcc_engine*>(e), &m);
  if (missing && missing_len > 0) snprintf(missing, missing_len, "%s", m.c_str());
  return r == 0 ? 1 : 0;
}
static int ensure_ready(lcc_engine* e) {
  if (!e) return fail(LCC_ERR_ARG, "null engine");
  if (!e->ws) return fail(LCC_ERR_STATE, "buffers not bound");
  if (!e->weights_resolved) {
    std::string m;
    if (resolve_weights(e, &m) != 0) return fail(LCC_ERR_STATE, "missing weights: %.400s", m.c_str());
  }
  return 0;
}

// meta ring: fill host slot, one async H2D copy, return device pointers with the same offsets
struct MetaWriter {
  lcc_engine* e; int slot; char* host; char* dev; size_t off = 0, cap;
  template <class T>
  T* put(const T* src, size_t n, T** dev_out) {
    T* h = reinterpret_cast<T*>(host + off);
    if (off + n * sizeof(T) > cap) return nullptr;
    if (src) memcpy(h, src, n * sizeof(T));
    *dev_out = reinterpret_cast<T*>(dev + off);
    off = align_up(off + n * sizeof(T), 16);
    return h;
  }
};
static int meta_begin(lcc_engine* e, MetaWriter* mw) {
  const int s = e->meta_next;
  e->meta_next = (s + 1) % META_RING;
  if (e->meta_ev_used[s]) HIP_TRY(hipEventSynchronize(e->meta_ev[s]));
  mw->e = e; mw->slot = s; mw->host = e->meta_host + (size_t)s * e->meta_slot_bytes; mw->dev = e->meta_dev + (size_t)s * e->meta_slot_bytes;
  mw->off = 0; mw->cap = e->meta_slot_bytes;
  return 0;
}
static int meta_commit(MetaWriter* mw, hipStream_t st) {
  if (mw->off == 0) return 0;
  HIP_TRY(hipMemcpyAsync(mw->dev, mw->host, mw->off, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(mw->e->meta_ev[mw->slot], st));
  mw->e->meta_ev_used[mw->slot] = true;
  return 0;
}

extern "C" int lcc_slot_reset(lcc_engine* e, int slot, void* stream) {
  if (!e || slot < 0 || slot >= e->lim.max_slots) return fail(LCC_ERR_ARG, "bad slot");
  if (!e->state) return fail(LCC_ERR_STATE, "buffers not bound");
  hipStream_t st = (hipStream_t)stream;
  e->h_kv_len[slot] = 0; e->h_pos[slot] = 0;
  HIP_TRY(hipMemsetAsync(e->d_kv_len + slot, 0, 4, st));
  HIP_TRY(hipMemsetAsync(e->d_pos + slot, 0, 4, st));
  HIP_TRY(hipMemsetAsync(e->d_hist_col + slot, 0, 4, st));
  HIP_TRY(hipMemsetAsync(e->d_done + slot, 0, 4, st));
  HIP_TRY(hipMemsetAsync(e->d_rng_ctr + slot, 0, 4, st));
  HIP_TRY(hipMemsetAsync(e->d_seen + (size_t)slot * e->words, 0, (size_t)e->words * 4, st));
  return 0;
}
extern "C" int lcc_slot_set_length(lcc_engine* e, int slot, int kv_len, int next_pos, void* stream) {
  if (!e || slot < 0 || slot >= e->lim.max_slots) return fail(LCC_ERR_ARG, "bad slot");
  if (kv_len < 0 || kv_len > e->lim.max_kv_len) return fail(LCC_ERR_STATE, "kv_len %d out of range", kv_len);
  hipStream_t st = (hipStream_t)stream;
  MetaWriter mw; LCC_TRY(meta_begin(e, &mw));
  int32_t v[2] = {kv_len, next_pos}; int32_t* d = nullptr;
  mw.put<int32_t>(v, 2, &d);
  LCC_TRY(meta_commit(&mw, st));
  HIP_TRY(hipMemcpyAsync(e->d_kv_len + slot, d, 4, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(e->d_pos + slot, d + 1, 4, hipMemcpyDeviceToDevice, st));
  e->h_kv_len[slot] = kv_len; e->h_pos[slot] = next_pos;
  return 0;
}
extern "C" int lcc_slot_get_length(const lcc_engine* e, int slot, int* kv_len, int* next_pos) {
  if (!e || slot < 0 || slot >= e->lim.max_slots) return fail(LCC_ERR_ARG, "bad slot");
  if (kv_len) *kv_len = e->h_kv_len[slot];
  if (next_pos) *next_pos = e->h_pos[slot];
  return 0;
}
extern "C" int lcc_slot_read_tokens(lcc_engine* e, int slot, int32_t* out, int max_n, int* n_generated, void* stream) {
  if (!e || slot < 0 || slot >= e->lim.max_slots || !out || max_n < 0) return fail(LCC_ERR_ARG, "bad args");
  hipStream_t st = (hipStream_t)stream;
  int32_t v[3];
  unsigned chain_err = 0;
  HIP_TRY(hipMemcpyAsync(&chain_err, e->d_chain + 128, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(&v[0], e->d_kv_len + slot, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(&v[1], e->d_pos + slot, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(&v[2], e->d_hist_col + slot, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (chain_err != 0) {   // a consumer block of a chained decode launch gave up waiting: the step's results are not valid
    g_decode_chain = 0;   // (never observed; the separate launches are the safe path from here on)
    HIP_TRY(hipMemset(e->d_chain, 0, 160 * sizeof(unsigned)));
    std::fill(e->chain_epoch.begin(), e->chain_epoch.end(), 0u);
    return fail(LCC_ERR_STATE, "decode: a chained launch hand-off timed out (results invalid); chained launches are now disabled");
  }
  e->h_kv_len[slot] = v[0]; e->h_pos[slot] = v[1];   // device counters are authoritative (EOS freezes them)
  const int n = std::min(std::min(v[2], max_n), e->lim.max_history);
  if (n > 0) HIP_TRY(hipMemcpy(out, e->d_history + (size_t)slot * e->lim.max_history, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (n_generated) *n_generated = v[2];
  return 0;
}

// ------------------------------------------------------------------------------------------------
// ViT
// ------------------------------------------------------------------------------------------------
extern "C" int lcc_vit_encode(lcc_engine* e, int n_clips, const lcc_clip* clips, const float mean255[3], const float std255[3],
                              const float* rope_cos, const float* rope_sin, void* out_embeds, void* stream) {
  LCC_TRY(ensure_ready(e));
  if (n_clips <= 0 || !clips || !rope_cos || !rope_sin || !out_embeds) return fail(LCC_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  const int E = e->E, heads = e->c.vit_heads, MLP = e->c.vit_mlp, H = e->c.hidden_size, PD = e->c.patch_dim;
  // segment tables
  std::vector<int32_t> seg_start, seg_len, seg_blk, seg_of_patch, tile_seg, tile_q0, grp_seg, grp_q0, g8_seg, g8_q0;
  int P = 0, blocks = 0;
  for (int ci = 0; ci < n_clips; ++ci) {
    const lcc_clip& c = clips[ci];
    if (c.T <= 0 || c.H % 28 || c.W % 28 || c.H <= 0 || c.W <= 0) return fail(LCC_ERR_SHAPE, "clip %d: T=%d H=%d W=%d (H,W must be multiples of 28)", ci, c.T, c.H, c.W);
    if (!c.frames && !c.pixel_values) return fail(LCC_ERR_ARG, "clip %d has neither frames nor pixel_values", ci);
    const int gt = (c.T + 1) / 2, n = (c.H / 14) * (c.W / 14);
    for (int t = 0; t < gt; ++t) {
      const int sg = (int)seg_start.size();
      seg_start.push_back(P); seg_len.push_back(n); seg_blk.push_back(blocks);
      for (int q = 0; q < n; q += 32) { tile_seg.push_back(sg); tile_q0.push_back(q); }
      for (int q = 0; q < n; q += 128) { grp_seg.push_back(sg); grp_q0.push_back(q); }
      for (int q = 0; q < n; q += 256) { g8_seg.push_back(sg); g8_q0.push_back(q); }     // attention variant 3: 8 waves x 32 rows
      seg_of_patch.insert(seg_of_patch.end(), n, sg);
      P += n; blocks += (n + 31) / 32;
    }
  }
  if (P > e->lim.max_patches) return fail(LCC_ERR_STATE, "%d patches > max_patches %d", P, e->lim.max_patches);
  const int n_tiles = (int)tile_seg.size(), n_seg = (int)seg_start.size(), n_groups = (int)grp_seg.size();

  const bool own = e->ws_vit != nullptr;     // private buffers: this call may overlap LLM work on another stream
  Carver cv; cv.base = own ? e->ws_vit : e->ws;
  bf16_t* patches = cv.take<bf16_t>((size_t)P * PD);
  bf16_t* x = cv.take<bf16_t>((size_t)P * E);
  bf16_t* xn = cv.take<bf16_t>((size_t)P * E);
  bf16_t* attn = cv.take<bf16_t>((size_t)P * E);
  bf16_t* qkv = cv.take<bf16_t>((size_t)P * 3 * E);
  bf16_t* mlp = cv.take<bf16_t>((size_t)P * MLP);
  bf16_t* vt = cv.take<bf16_t>((size_t)heads * blocks * 80 * 32);
  bf16_t* mg = cv.take<bf16_t>((size_t)(P / 4) * 4 * E);
  if (cv.off > (own ? e->ws_vit_bytes : e->ws_bytes)) return fail(LCC_ERR_STATE, "workspace too small for %d patches", P);

  MetaWriter mw;
  int vslot = -1;
  if (own) {   // private 2-slot ring; a slot is reused only after the ViT call that used it has completely finished
    vslot = e->vmeta_next; e->vmeta_next ^= 1;
    if (e->vmeta_ev_used[vslot]) HIP_TRY(hipEventSynchronize(e->vmeta_ev[vslot]));
    mw.e = e; mw.slot = vslot; mw.host = e->vmeta_host + (size_t)vslot * e->vmeta_slot_bytes; mw.dev = e->vmeta_dev + (size_t)vslot * e->vmeta_slot_bytes;
    mw.off = 0; mw.cap = e->vmeta_slot_bytes;
  } else {
    LCC_TRY(meta_begin(e, &mw));
  }
  int32_t *d_seg_start, *d_seg_len, *d_seg_blk, *d_seg_of_patch, *d_tile_seg, *d_tile_q0, *d_grp_seg, *d_grp_q0, *d_g8_seg, *d_g8_q0;
  const int n_groups8 = (int)g8_seg.size();
  if (!mw.put(seg_start.data(), n_seg, &d_seg_start) || !mw.put(seg_len.data(), n_seg, &d_seg_len) ||
      !mw.put(seg_blk.data(), n_seg, &d_seg_blk) || !mw.put(seg_of_patch.data(), P, &d_seg_of_patch) ||
      !mw.put(tile_seg.data(), n_tiles, &d_tile_seg) || !mw.put(tile_q0.data(), n_tiles, &d_tile_q0) ||
      !mw.put(grp_seg.data(), n_groups, &d_grp_seg) || !mw.put(grp_q0.data(), n_groups, &d_grp_q0) ||
      !mw.put(g8_seg.data(), n_groups8, &d_g8_seg) || !mw.put(g8_q0.data(), n_groups8, &d_g8_q0))
    return fail(LCC_ERR_STATE, "meta ring slot too small");
  if (own) HIP_TRY(hipMemcpyAsync(mw.dev, mw.host, mw.off, hipMemcpyHostToDevice, st));
  else LCC_TRY(meta_commit(&mw, st));

  // K1: patches
  {
    size_t row = 0;
    for (int ci = 0; ci < n_clips; ++ci) {
      const lcc_clip& c = clips[ci];
      const size_t np = (size_t)((c.T + 1) / 2) * (c.H / 14) * (c.W / 14);
      if (c.frames) LCC_TRY(patchify_norm_u8(c.frames, c.layout, c.T, c.H, c.W, mean255, std255, patches + row * PD, PD, st));
      else LCC_TRY(cast_f32_bf16(c.pixel_values, patches + row * PD, (int64_t)np * PD, st));
      row += np;
    }
  }
  HIP_TRY(hipMemsetAsync(vt, 0, (size_t)heads * blocks * 80 * 32 * 2, st));
  GemmArgs g;
  // K2: patch embed (Conv3d k=s=(2,14,14) == GEMM, no bias; K = 1176 is not a multiple of 32: row-major weight)
  g = GemmArgs(); g.w_packed = 0; g.A = patches; g.lda = PD; g.W = e->patch_embed; g.ldw = PD; g.C = x; g.ldc = E; g.M = P; g.N = E; g.K = PD;
  LCC_TRY(gemm_bf16(g, st));
  if ((e->vit_taps || e->vit_over) && P > e->vit_tap_rows) return fail(LCC_ERR_STATE, "ViT taps bound for %d rows, call has %d patches", e->vit_tap_rows, P);
  const size_t tap_stride = (size_t)e->vit_tap_rows * E;
  if (e->vit_taps) HIP_TRY(hipMemcpyAsync(e->vit_taps, x, (size_t)P * E * 2, hipMemcpyDeviceToDevice, st));   // tap 0 = PatchEmbed output
  for (int l = 0; l < e->c.vit_depth; ++l) {
    const VitLayerW& L = e->vit[l];
    if (e->vit_over) HIP_TRY(hipMemcpyAsync(x, e->vit_over + (size_t)l * tap_stride, (size_t)P * E * 2, hipMemcpyDeviceToDevice, st));
    LCC_TRY(layernorm_bf16(x, L.ln1_w, L.ln1_b, xn, P, E, 1e-6f, st));
    g = GemmArgs(); g.w_packed = 1; g.A = xn; g.lda = E; g.W = L.qkv_w; g.ldw = E; g.bias = L.qkv_b; g.C = qkv; g.ldc = 3 * E; g.M = P; g.N = 3 * E; g.K = E;
    LCC_TRY(gemm_bf16(g, st));
    LCC_TRY(vit_rope_vt_bf16(qkv, rope_cos, rope_sin, d_seg_of_patch, d_seg_start, d_seg_blk, vt, P, heads, blocks, st));
    // 32x32x16 kernel (8 waves x 32 rows per block) once its grid fills the chip: 8 streams' chunks = 768 blocks, 170 vs 359 us per
    // block of the tower; ONE 2-frame chunk is only 6 groups x 16 heads = 96 blocks (49 us) -- there the 16-row-per-wave LDS-shared
    // kernel with twice the blocks stays (41 us)
    // (the 4-wave form of the 32x32x16 kernel -- 128-row groups, 192 blocks for one chunk -- measured the same as the 16-row kernel:
    // 261.6 vs 262.1 tokens/s without prefetch, profiles/r03/knob_sweeps_call11_13.txt; off unless LCC_VIT32_MIN_BLOCKS4 says otherwise)
    static const int vit32_min4 = [] { const char* v = getenv("LCC_VIT32_MIN_BLOCKS4"); return v ? atoi(v) : (1 << 30); }();
    if (get_attn_variant() == 3 && e->vit_hd == 80 && (long)n_groups8 * heads >= 224)
      LCC_TRY(attn_vit32_launch(qkv, vt, attn, d_g8_seg, d_g8_q0, d_seg_start, d_seg_len, d_seg_blk, n_groups8, heads, blocks,
                                1.4426950408889634f / sqrtf(80.f), st, 256));
    else if (get_attn_variant() == 3 && e->vit_hd == 80 && (long)n_groups * heads >= vit32_min4)   // 128-row groups: 4 waves, one per SIMD
      LCC_TRY(attn_vit32_launch(qkv, vt, attn, d_grp_seg, d_grp_q0, d_seg_start, d_seg_len, d_seg_blk, n_groups, heads, blocks,
                                1.4426950408889634f / sqrtf(80.f), st, 128));
    else
      LCC_TRY(attn_vit_bf16(qkv, vt, attn, d_tile_seg, d_tile_q0, d_seg_start, d_seg_len, d_seg_blk, n_tiles, heads, blocks, d_grp_seg, d_grp_q0,
                            n_groups, st));
    g = GemmArgs(); g.w_packed = 1; g.A = attn; g.lda = E; g.W = L.proj_w; g.ldw = E; g.bias = L.proj_b; g.residual = x; g.ldr = E; g.C = x; g.ldc = E;
    g.M = P; g.N = E; g.K = E; g.epilogue = LCC_EPI_RESIDUAL;
    LCC_TRY(gemm_bf16(g, st));
    LCC_TRY(layernorm_bf16(x, L.ln2_w, L.ln2_b, xn, P, E, 1e-6f, st));
    g = GemmArgs(); g.w_packed = 1; g.A = xn; g.lda = E; g.W = L.fc1_w; g.ldw = E; g.bias = L.fc1_b; g.C = mlp; g.ldc = MLP; g.M = P; g.N = MLP; g.K = E;
    g.epilogue = LCC_EPI_QUICK_GELU;
    LCC_TRY(gemm_bf16(g, st));
    g = GemmArgs(); g.w_packed = 1; g.A = mlp; g.lda = MLP; g.W = L.fc2_w; g.ldw = MLP; g.bias = L.fc2_b; g.residual = x; g.ldr = E; g.C = x; g.ldc = E;
    g.M = P; g.N = E; g.K = MLP; g.epilogue = LCC_EPI_RESIDUAL;
    LCC_TRY(gemm_bf16(g, st));
    if (e->vit_taps) HIP_TRY(hipMemcpyAsync(e->vit_taps + (size_t)(l + 1) * tap_stride, x, (size_t)P * E * 2, hipMemcpyDeviceToDevice, st));
  }
  // merger: LN -> view [P/4, 4E] -> Linear + GELU -> Linear
  if (e->vit_over) HIP_TRY(hipMemcpyAsync(x, e->vit_over + (size_t)e->c.vit_depth * tap_stride, (size_t)P * E * 2, hipMemcpyDeviceToDevice, st));
  LCC_TRY(layernorm_bf16(x, e->mg_ln_w, e->mg_ln_b, xn, P, E, 1e-6f, st));
  g = GemmArgs(); g.w_packed = 1; g.A = xn; g.lda = 4 * E; g.W = e->mg_fc1_w; g.ldw = 4 * E; g.bias = e->mg_fc1_b; g.C = mg; g.ldc = 4 * E;
  g.M = P / 4; g.N = 4 * E; g.K = 4 * E; g.epilogue = LCC_EPI_GELU_ERF;
  LCC_TRY(gemm_bf16(g, st));
  g = GemmArgs(); g.w_packed = 1; g.A = mg; g.lda = 4 * E; g.W = e->mg_fc2_w; g.ldw = 4 * E; g.bias = e->mg_fc2_b; g.C = (bf16_t*)out_embeds; g.ldc = H;
  g.M = P / 4; g.N = H; g.K = 4 * E;
  LCC_TRY(gemm_bf16(g, st));
  if (own) { HIP_TRY(hipEventRecord(e->vmeta_ev[vslot], st)); e->vmeta_ev_used[vslot] = true; }
  return check_launch("lcc_vit_encode");
}

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
  b->partial = cv.take<float>(std::max((size_t)MAX_SPLIT * 16 * std::max<size_t>(e->qkvd, H), (size_t)4 * std::min<size_t>(S, 4096) * H));
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
  const int tp_o = cx.skinny ? 1 : gemm_tiled_num_splits(S, H, e->qd);
  const int tp_dn = cx.skinny ? 1 : gemm_tiled_num_splits(S, H, I);
  // q/k/v of a short prefill (one streaming chunk): split-K slabs consumed by the rope / KV-append kernel (267.5 -> 269.1 tok/s single
  // stream; LCC_PREFILL_QKV_SPLIT=0 restores the bf16 GEMM output)
  static const int qkv_split_on = [] { const char* v = getenv("LCC_PREFILL_QKV_SPLIT"); return v ? atoi(v) : 1; }();
  const int tp_qkv = (cx.skinny || !qkv_split_on || e->c.llm_fp8 || (size_t)4 * std::min<size_t>(e->lim.max_new_rows, 4096) * H <
                      (size_t)8 * S * e->qkvd) ? 1 : gemm_tiled_num_splits(S, e->qkvd, H);
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
int run_decode_layers_v2(lcc_engine* e, const LlmBuffers& b, int B, const int32_t* d_slots, int nsplit_attn, hipStream_t st) {
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
    LCC_TRY(attn_decode_bf16(b.q, b.attn, d_slots, e->d_kv_len, e->d_kv_base, e->lay, l, B, e->c.n_q_heads, nsplit_attn, b.ws_o, b.ws_ml, st));
    a = DgArgs(); a.W = L.o_w; a.wscale = L.o_s; a.M = B; a.N = H; a.K = e->qd; a.X = b.attn; a.ldx = e->qd; a.Hres = b.h; a.stats_out = b.stats;
    LCC_TRY(dgemv_resid(a, st));
    if (e->llm_taps) HIP_TRY(hipMemcpyAsync(e->llm_taps + (size_t)(2 * l + 1) * tap_stride, b.h, tap_bytes, hipMemcpyDeviceToDevice, st));
    a = DgArgs(); a.W = L.gate_up_w; a.wscale = L.gate_up_s; a.M = B; a.N = 2 * I; a.K = H; a.H = b.h; a.stats = b.stats; a.n_stat = H / 16; a.norm_w = L.post_norm;
    a.eps = eps; a.C = b.act; a.ldc = I;
    const bool prof = e->prof_on && l == e->c.n_layers / 2 && 2 * (e->prof_n + 1) <= (int)e->prof_ev.size();   // one sample per step
    if (prof) HIP_TRY(hipEventRecord(e->prof_ev[2 * e->prof_n], st));
    LCC_TRY(dgemv_norm_swiglu(a, st));
    if (prof) { HIP_TRY(hipEventRecord(e->prof_ev[2 * e->prof_n + 1], st)); e->prof_n++; }
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
  if (sp && sp->do_sample && sp->top_k != 1) {
    LCC_TRY(sample_topk_topp(logits, V, B, V, e->d_seen, e->words, d_slots, pen <= 0.f ? 1.0f : pen, thr_tok, use_thr, thr, sp->eos_token,
                             eos2, sp->suppress_eos, e->d_done, e->d_cur_tok, e->d_history, e->lim.max_history, e->d_hist_col,
                             sp->scores_out, sp->temperature, sp->top_k, sp->top_p, sp->seed, e->d_rng_ctr, st));
    if (e->forced != nullptr && B == e->forced_B && step_index < e->forced_steps)
      LCC_TRY(force_tokens(d_slots, e->forced + (size_t)step_index * B, B, e->d_cur_tok, e->d_history, e->lim.max_history, e->d_hist_col, st));
    return 0;
  }
  // top_k == 1 (the released generation_config): the top-k warper leaves one finite score -> the draw IS the argmax
  LCC_TRY(sample_greedy(logits, V, B, V, e->d_seen, e->words, d_slots, pen <= 0.f ? 1.0f : pen, thr_tok, use_thr, thr,
                        sp ? sp->eos_token : -1, eos2, sp ? sp->suppress_eos : 0, e->d_done, e->d_cur_tok, e->d_history, e->lim.max_history,
                        e->d_hist_col, sp ? sp->scores_out : nullptr, b.ws_ml, st));
  if (e->forced != nullptr && B == e->forced_B && step_index < e->forced_steps)
    LCC_TRY(force_tokens(d_slots, e->forced + (size_t)step_index * B, B, e->d_cur_tok, e->d_history, e->lim.max_history, e->d_hist_col, st));
  return 0;
}
}  // namespace

extern "C" int lcc_llm_prefill(lcc_engine* e, int n_streams, const int32_t* slots, const int32_t* n_new, const int32_t* ids,
                               const int32_t* vit_index, const void* vit_embeds, const int32_t* pos3, const lcc_sampling* sp,
                               void* stream) {
  LCC_TRY(ensure_ready(e));
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
  const int tile_rows = (mfma32 || (long)((S + 31) / 32) * e->c.n_q_heads >= 6144) ? 32 : 16;
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
    int max_kv = 0;
    for (int b = 0; b < n_streams; ++b) max_kv = std::max(max_kv, e->h_kv_len[slots[b]] + n_new[b]);
    const long waves = (long)n_tiles * e->c.n_q_heads;
    int ks = (int)std::min<long>(8, 3072 / std::max<long>(waves, 1));
    ks = std::min(ks, (max_kv / 32) / 16);          // >= 16 key tiles per split
    cx.kv_split = (S <= 1024 && ks >= 2) ? ks : 1;
    if (mfma32) {
      // one 8-wave block per CU and (tile, KV head, split).  Measured (tools/bench_attn.py, profiles/r03/attn_prefill_microbench.jsonl):
      // a split costs its fp32 partials twice (write + combine launch: 3,088 rows x 28 heads x 3 splits = 137 MB, 415 vs 373 us at 8
      // streams), so keys are split only while the unsplit grid cannot fill ONE round of the chip (one stream's chunk: 52 blocks ->
      // 4 splits, 68 vs 177 us); then the split count that fills whole rounds best, slightly preferring fewer splits.
      const int cus = e->cu_count;     // of the engine's device, queried once at create time
      const long base = (long)n_tiles * e->c.n_kv_heads;
      const int ks_max = (S <= 1024 && base < cus) ? std::max(1, std::min(8, (max_kv / 32) / 8)) : 1;   // >= 8 key tiles per split
      float best = -1.f; int best_ks = 1;
      for (int k = 1; k <= ks_max; ++k) {
        const long blocks = base * k, rounds = (blocks + cus - 1) / cus;
        const float u = (float)blocks / (float)(rounds * cus) - 0.015f * (float)k;
        if (u > best) { best = u; best_ks = k; }
      }
      static const int forced = [] { const char* v = getenv("LCC_ATTN32_SPLIT"); return v ? atoi(v) : 0; }();
      // a forced split obeys the same bound as the automatic one: the partial buffers hold min(S, 1024) x heads x 8 slots (carve_llm)
      cx.kv_split = (forced > 0 && S <= 1024) ? std::min(forced, std::max(1, std::min(8, (max_kv / 32) / 8))) : (forced > 0 ? 1 : best_ks);
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
  if (n_streams <= 0 || !slots || n_steps < 0) return fail(LCC_ERR_ARG, "bad argument");
  // <= 16 streams: one MFMA column tile of the weight-streaming GEMVs.  17..64: the rows go through the 64-row GEMM tiles of the
  // prefill path (every weight byte is still read once per step) with the decode attention; beyond that the caller splits.
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
  cx.S = n_streams; cx.skinny = n_streams <= 16; cx.tok_stream = d_slots; cx.tok_pos = nullptr; cx.slots = d_slots; cx.B = n_streams;
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
      LCC_TRY(run_decode_layers_v2(e, bf, n_streams, d_slots, nsplit, st));
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
namespace lcc { long long g_launch_counts[LC_COUNT] = {}; }
extern "C" int lcc_debug_launch_counts(int64_t* out, int n, int reset) {
  if (n < 0 || n > LC_COUNT || (n > 0 && !out)) return fail(LCC_ERR_ARG, "bad argument");
  for (int i = 0; i < n; ++i) out[i] = (int64_t)g_launch_counts[i];
  if (reset) for (int i = 0; i < LC_COUNT; ++i) g_launch_counts[i] = 0;
  return 0;
}
extern "C" int lcc_debug_set_fused_tails(int on) { g_fuse_tails = on ? 1 : 0; return 0; }
extern "C" int lcc_debug_set_decode_chain(int on) { g_decode_chain = on ? 1 : 0; return 0; }
extern "C" int lcc_debug_set_resid_waves(int mode) { return set_resid_waves(mode); }
extern "C" int lcc_debug_set_decode_path(int path) {
  if (path != 0 && path != 1) return fail(LCC_ERR_ARG, "decode path must be 0 (round-1 launch sequence) or 1 (v2)");
  g_decode_path = path;
  return 0;
}
// bit 0: engine uses the fused decode attention for batches of >= 16 (stream, KV head) pairs (default); bit 2: for every batch;
// bit 1: its key splits are merged in-launch (ticket) instead of by a combine launch
extern "C" int lcc_debug_set_fused_attn(int mode) {
  g_fused_attn = (mode & 4) ? 2 : (mode & 1);
  set_attn_fused_tail((mode & 2) ? 0 : 1);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// operator-level C-ABI wrappers
// ------------------------------------------------------------------------------------------------
static KvLayout to_lay(lcc_kv_layout l) { return KvLayout{l.n_layers, l.n_kv_heads, l.lmax, l.head_dim}; }
#define OP_RET(call, name)                                   \
  do {                                                       \
    int r__ = (call);                                        \
    if (r__ != 0) return fail(r__, "%s: invalid arguments (%d)", name, r__); \
    return check_launch(name);                               \
  } while (0)

extern "C" int lcc_debug_set_gemv_variant(int variant) { set_gemv_variant(variant); return 0; }
extern "C" int lcc_debug_set_gemm_variant(int variant) { set_gemm_variant(variant); return 0; }
extern "C" int lcc_debug_set_attn_variant(int variant) { set_attn_variant(variant); return 0; }
extern "C" int lcc_debug_set_fused_tails(int on);
extern "C" int lcc_gemm_bf16(const void* A, int lda, const void* W, int ldw, int w_layout, const void* bias, const void* residual,
                             int ldr, void* C, int ldc, int M, int N, int K, int epilogue, float* partial, int nsplit, void* stream) {
  if (!A || !W || (!C && !partial)) return fail(LCC_ERR_ARG, "lcc_gemm_bf16: null pointer");
  if (w_layout != 0 && w_layout != 1) return fail(LCC_ERR_ARG, "lcc_gemm_bf16: w_layout must be 0 or 1");
  GemmArgs g; g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.w_packed = w_layout; g.bias = (const bf16_t*)bias;
  g.residual = (const bf16_t*)residual; g.ldr = ldr; g.C = (bf16_t*)C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.epilogue = epilogue;
  g.partial = partial; g.nsplit = nsplit;
  if (partial && C == nullptr) g.C = (bf16_t*)partial;  // alignment check only
  OP_RET(gemm_bf16(g, (hipStream_t)stream), "lcc_gemm_bf16");
}
extern "C" int lcc_gemv_num_splits(int N, int K) { return gemv_num_splits(N, K); }
extern "C" int lcc_gemm_w8_bf16(const void* A, int lda, const void* W8, const float* wscale, const void* bias, const void* residual,
                                int ldr, void* C, int ldc, int M, int N, int K, int epilogue, float* partial, int nsplit,
                                void* dq_scratch, void* stream) {
  if (!A || !W8 || !wscale || (!C && !partial)) return fail(LCC_ERR_ARG, "lcc_gemm_w8_bf16: null pointer");
  if (M > 16 && !dq_scratch) return fail(LCC_ERR_ARG, "lcc_gemm_w8_bf16: M > 16 needs dq_scratch (N*K bf16)");
  GemmArgs g; g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W8; g.ldw = K; g.w_packed = 1; g.bias = (const bf16_t*)bias;
  g.residual = (const bf16_t*)residual; g.ldr = ldr; g.C = (bf16_t*)C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.epilogue = epilogue;
  g.partial = partial; g.nsplit = nsplit; g.w_fp8 = 1; g.wscale = wscale; g.dq_scratch = (bf16_t*)dq_scratch;
  if (partial && C == nullptr) g.C = (bf16_t*)partial;
  OP_RET(gemm_bf16(g, (hipStream_t)stream), "lcc_gemm_w8_bf16");
}
extern "C" int lcc_debug_mfma_probe(const void* A, const void* B, float* D, void* stream) {
  if (!A || !B || !D) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(mfma_probe((const bf16_t*)A, (const bf16_t*)B, D, (hipStream_t)stream), "lcc_debug_mfma_probe");
}
extern "C" int lcc_patchify_norm_u8(const uint8_t* frames, int layout, int T, int H, int W, const float mean255[3],
                                    const float std255[3], void* out, int ld, void* stream) {
  if (!frames || !out || !mean255 || !std255) return fail(LCC_ERR_ARG, "lcc_patchify_norm_u8: null pointer");
  OP_RET(patchify_norm_u8(frames, layout, T, H, W, mean255, std255, (bf16_t*)out, ld, (hipStream_t)stream), "lcc_patchify_norm_u8");
}
extern "C" int lcc_resize_bicubic_aa_u8(const uint8_t* src, int layout, int T, int Hin, int Win, uint8_t* dst, int Hout, int Wout,
                                        const int32_t* xmin, const int32_t* xsize, const float* wx, int kx, const int32_t* ymin,
                                        const int32_t* ysize, const float* wy, int ky, float* tmp, void* stream) {
  if (!src || !dst || !xmin || !xsize || !wx || !ymin || !ysize || !wy || !tmp) return fail(LCC_ERR_ARG, "lcc_resize_bicubic_aa_u8: null pointer");
  OP_RET(resize_bicubic_aa_u8(src, layout, T, Hin, Win, dst, Hout, Wout, xmin, xsize, wx, kx, ymin, ysize, wy, ky, tmp, (hipStream_t)stream),
         "lcc_resize_bicubic_aa_u8");
}
extern "C" int lcc_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream) {
  if (!in || !out) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(cast_f32_bf16(in, (bf16_t*)out, n, (hipStream_t)stream), "lcc_cast_f32_bf16");
}
extern "C" int lcc_layernorm_bf16(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps, void* stream) {
  if (!x || !w || !b || !y) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(layernorm_bf16((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, rows, dim, eps, (hipStream_t)stream), "lcc_layernorm_bf16");
}
extern "C" int lcc_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int dim, float eps, void* stream) {
  if (!x || !w || !y) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(rmsnorm_bf16((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rows, dim, eps, (hipStream_t)stream), "lcc_rmsnorm_bf16");
}
extern "C" int lcc_add_rmsnorm_bf16(void* h, const void* delta_bf16, const float* delta_partial, int nsplit, const void* w, void* y,
                                    int rows, int dim, float eps, void* stream) {
  if (!h || (w && !y)) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(add_rmsnorm_bf16((bf16_t*)h, (const bf16_t*)delta_bf16, delta_partial, nsplit, (const bf16_t*)w, (bf16_t*)y, rows, dim, eps,
                          (hipStream_t)stream), "lcc_add_rmsnorm_bf16");
}
extern "C" int lcc_swiglu_bf16(const void* gate, const void* up, void* out, int64_t n, void* stream) {
  if (!gate || !up || !out) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(swiglu_bf16((const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)out, n, (hipStream_t)stream), "lcc_swiglu_bf16");
}
extern "C" int lcc_vit_rope_vt_bf16(void* qkv, const float* cos, const float* sin, const int32_t* seg_of_patch, const int32_t* seg_start,
                                    const int32_t* seg_blk_start, void* vt, int P, int heads, int total_blocks, void* stream) {
  if (!qkv || !cos || !sin || !seg_of_patch || !seg_start || !seg_blk_start || !vt) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(vit_rope_vt_bf16((bf16_t*)qkv, cos, sin, seg_of_patch, seg_start, seg_blk_start, (bf16_t*)vt, P, heads, total_blocks,
                          (hipStream_t)stream), "lcc_vit_rope_vt_bf16");
}
extern "C" int lcc_attn_vit_bf16(const void* qkv, const void* vt, void* out, const int32_t* tile_seg, const int32_t* tile_q0,
                                 const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_tiles, int heads,
                                 int total_blocks, const int32_t* grp_seg, const int32_t* grp_q0, int n_groups, void* stream) {
  if (!qkv || !vt || !out || !tile_seg || !tile_q0 || !seg_start || !seg_len || !seg_blk_start) return fail(LCC_ERR_ARG, "null pointer");
  if (n_groups > 0 && (!grp_seg || !grp_q0)) return fail(LCC_ERR_ARG, "null group table");
  OP_RET(attn_vit_bf16((const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, tile_seg, tile_q0, seg_start, seg_len, seg_blk_start, n_tiles,
                       heads, total_blocks, grp_seg, grp_q0, n_groups, (hipStream_t)stream), "lcc_attn_vit_bf16");
}
extern "C" int lcc_attn_vit32_bf16(const void* qkv, const void* vt, void* out, const int32_t* grp_seg, const int32_t* grp_q0,
                                   const int32_t* seg_start, const int32_t* seg_len, const int32_t* seg_blk_start, int n_groups, int heads,
                                   int total_blocks, int group_rows, void* stream) {
  if (!qkv || !vt || !out || !grp_seg || !grp_q0 || !seg_start || !seg_len || !seg_blk_start) return fail(LCC_ERR_ARG, "null pointer");
  if (group_rows != 256 && group_rows != 128) return fail(LCC_ERR_ARG, "group_rows must be 256 or 128, got %d", group_rows);
  OP_RET(attn_vit32_launch((const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, grp_seg, grp_q0, seg_start, seg_len, seg_blk_start, n_groups, heads,
                           total_blocks, 1.4426950408889634f / sqrtf(80.f), (hipStream_t)stream, group_rows), "lcc_attn_vit32_bf16");
}
extern "C" int lcc_mrope_table(const int32_t* pos3, const float* inv_freq, int S, int sec_t, int sec_h, void* cos, void* sin, void* stream) {
  if (!pos3 || !inv_freq || !cos || !sin) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(mrope_table(pos3, inv_freq, S, sec_t, sec_h, (bf16_t*)cos, (bf16_t*)sin, (hipStream_t)stream), "lcc_mrope_table");
}
extern "C" int lcc_rope_kv_append_bf16(const void* qkv_bf16, const float* qkv_partial, int nsplit, const void* bias, const void* cos,
                                       const void* sin, const int32_t* tok_stream, const int32_t* tok_pos, const int32_t* kv_len,
                                       void* const* kv_base, lcc_kv_layout lay, int layer, void* q_out, int S, int n_q_heads, void* stream) {
  if ((!qkv_bf16 && !qkv_partial) || !cos || !sin || !tok_stream || !kv_base || !q_out) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(rope_kv_append_bf16((const bf16_t*)qkv_bf16, qkv_partial, nsplit, (const bf16_t*)bias, (const bf16_t*)cos, (const bf16_t*)sin,
                             tok_stream, tok_pos, kv_len, (bf16_t* const*)kv_base, to_lay(lay), layer, (bf16_t*)q_out, S, n_q_heads,
                             (hipStream_t)stream), "lcc_rope_kv_append_bf16");
}
extern "C" int lcc_attn_prefill_bf16(const void* q, void* out, const int32_t* tile_stream, const int32_t* tile_q0, const int32_t* tile_nq,
                                     const int32_t* tile_pos0, void* const* kv_base, lcc_kv_layout lay, int layer, int n_tiles,
                                     int n_q_heads, int tile_rows, int nsplit, int n_rows, float* ws_o, float* ws_ml, void* stream) {
  if (!q || !out || !tile_stream || !tile_q0 || !tile_nq || !tile_pos0 || !kv_base) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(attn_prefill_bf16((const bf16_t*)q, (bf16_t*)out, tile_stream, tile_q0, tile_nq, tile_pos0, (bf16_t* const*)kv_base, to_lay(lay),
                           layer, n_tiles, n_q_heads, tile_rows, nsplit, n_rows, ws_o, ws_ml, (hipStream_t)stream), "lcc_attn_prefill_bf16");
}
extern "C" int lcc_attn_decode_bf16(const void* q, void* out, const int32_t* slots, const int32_t* kv_len, void* const* kv_base,
                                    lcc_kv_layout lay, int layer, int B, int n_q_heads, int nsplit, float* ws_o, float* ws_ml, void* stream) {
  if (!q || !out || !slots || !kv_len || !kv_base || !ws_o || !ws_ml || nsplit < 1) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(attn_decode_bf16((const bf16_t*)q, (bf16_t*)out, slots, kv_len, (bf16_t* const*)kv_base, to_lay(lay), layer, B, n_q_heads, nsplit,
                          ws_o, ws_ml, (hipStream_t)stream), "lcc_attn_decode_bf16");
}
extern "C" int lcc_attn_decode_fused_bf16(const float* qkv_partial, int nsplit_qkv, const void* bias, const void* cos, const void* sin,
                                          const int32_t* slots, const int32_t* kv_len, void* const* kv_base, lcc_kv_layout lay, int layer,
                                          void* out, int B, int n_q_heads, int nsplit, float* ws_o, float* ws_ml, int32_t* counters,
                                          void* stream) {
  if (!qkv_partial || !bias || !cos || !sin || !slots || !kv_len || !kv_base || !out || !counters) return fail(LCC_ERR_ARG, "null pointer");
  if (nsplit > 1 && (!ws_o || !ws_ml)) return fail(LCC_ERR_ARG, "nsplit > 1 needs the partial workspaces");
  OP_RET(attn_decode_fused_bf16(qkv_partial, nsplit_qkv, (const bf16_t*)bias, (const bf16_t*)cos, (const bf16_t*)sin, slots, kv_len,
                                (bf16_t* const*)kv_base, to_lay(lay), layer, B, n_q_heads, nsplit, ws_o, ws_ml, counters, (bf16_t*)out,
                                (hipStream_t)stream), "lcc_attn_decode_fused_bf16");
}
// micro-benchmark of the decode attention chain of one layer, launched back to back `iters` times from C++ (a Python loop cannot
// issue 5-us kernels fast enough).  variant 0: rope_kv_append + attn_decode + combine (three launches, nsplit_sep key splits);
// 1: fused kernel + combine launch; 2: fused kernel with the in-launch merge.  Returns the average microseconds per chain.
extern "C" int lcc_debug_bench_attn_decode(int variant, int iters, const float* qkv_partial, int nsplit_qkv, const void* bias,
                                           const void* cos, const void* sin, const int32_t* slots, const int32_t* kv_len,
                                           void* const* kv_base, lcc_kv_layout lay, int layer, void* q_scratch, void* out, int B,
                                           int n_q_heads, int nsplit_sep, int nsplit_fused, float* ws_o, float* ws_ml,
                                           int32_t* counters, float* out_us, void* stream) {
  if (!qkv_partial || !bias || !cos || !sin || !slots || !kv_len || !kv_base || !q_scratch || !out || !ws_o || !ws_ml || !counters || !out_us)
    return fail(LCC_ERR_ARG, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  const KvLayout L = to_lay(lay);
  set_attn_fused_tail(variant == 2 ? 0 : 1);
  int rc = 0;
  for (int it = -3; it < iters && rc == 0; ++it) {
    if (it == 0) HIP_TRY(hipEventRecord(e0, st));
    if (variant == 0) {
      rc = rope_kv_append_bf16(nullptr, qkv_partial, nsplit_qkv, (const bf16_t*)bias, (const bf16_t*)cos, (const bf16_t*)sin, slots, nullptr,
                               kv_len, (bf16_t* const*)kv_base, L, layer, (bf16_t*)q_scratch, B, n_q_heads, st);
      if (rc == 0) rc = attn_decode_bf16((const bf16_t*)q_scratch, (bf16_t*)out, slots, kv_len, (bf16_t* const*)kv_base, L, layer, B, n_q_heads,
                                         nsplit_sep, ws_o, ws_ml, st);
    } else {
      rc = attn_decode_fused_bf16(qkv_partial, nsplit_qkv, (const bf16_t*)bias, (const bf16_t*)cos, (const bf16_t*)sin, slots, kv_len,
                                  (bf16_t* const*)kv_base, L, layer, B, n_q_heads, nsplit_fused, ws_o, ws_ml, counters, (bf16_t*)out, st);
    }
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *out_us = ms * 1000.f / (float)std::max(1, iters);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  set_attn_fused_tail(1);
  if (rc != 0) return fail(rc, "lcc_debug_bench_attn_decode: invalid arguments (%d)", rc);
  return check_launch("lcc_debug_bench_attn_decode");
}
// ---- what does a device-wide hand-off cost?  (the number the "one persistent launch per decode layer" design stands or falls with)
// mode 0: `iters` grid barriers inside ONE launch of `blocks` co-resident blocks (monotonic agent-scope counter: arrive = relaxed
//         fetch_add after a release fence, wait = acquire loads with s_sleep; every wait is BOUNDED -- a block that gives up counts
//         itself in *fails and leaves, so a mis-sized grid cannot hang the GPU);
// mode 1: `iters` dependent launches of a kernel of `blocks` blocks that touches one cache line per block (the kernel boundary);
// mode 2: mode 0 with a 16-KB streaming read per block between barriers (a barrier under memory load);
// mode 3 / 4: modes 0 / 2 with the XCD-hierarchical barrier of grid_sync.h (per-XCD arrival counters, one release fence per XCD leader,
//         per-XCD generation words) -- the form MI355X_MICROARCH.md prices at 4.1 us for 256 workgroups.
typedef __attribute__((ext_vector_type(4))) unsigned int bench_u32x4;
__global__ __launch_bounds__(256) void grid_barrier_bench_kernel(unsigned* counter, unsigned* fails, int iters, int nblocks, const bench_u32x4* stream_src,
                                                                 unsigned* sink) {
  unsigned acc = 0;
  for (int it = 1; it <= iters; ++it) {
    if (stream_src != nullptr) {
      const bench_u32x4 v = __builtin_nontemporal_load(stream_src + ((size_t)(blockIdx.x * 997 + it) % 4096) * 1024 + threadIdx.x * 4);
      acc += v.x ^ v.w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)it * (unsigned)nblocks;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 100000) { atomicAdd(fails, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// modes 3 / 4: the same loop on the XCD-hierarchical barrier of grid_sync.h (MI355X_MICROARCH.md "barrier-xcd")
__global__ __launch_bounds__(256) void grid_barrier_xcd_bench_kernel(GridSyncState* gs, int iters, const bench_u32x4* stream_src, unsigned* sink) {
  GridSync g = gs_begin(gs);
  if (!g.ok) return;
  unsigned acc = 0;
  for (int it = 1; it <= iters; ++it) {
    if (stream_src != nullptr) {
      const bench_u32x4 v = __builtin_nontemporal_load(stream_src + ((size_t)(blockIdx.x * 997 + it) % 4096) * 1024 + threadIdx.x * 4);
      acc += v.x ^ v.w;
    }
    if (!gs_barrier(g)) return;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void boundary_bench_kernel(unsigned* buf, int it) {
  if (threadIdx.x == 0) buf[blockIdx.x * 32] = buf[((blockIdx.x + 1) % gridDim.x) * 32] + (unsigned)it;
}
extern "C" int lcc_debug_bench_grid_barrier(int mode, int blocks, int iters, void* scratch, size_t scratch_bytes, float* out_us, int* out_fails,
                                            void* stream) {
  if (!scratch || !out_us || !out_fails || blocks < 1 || blocks > 1024 || iters < 1) return fail(LCC_ERR_ARG, "bad argument");
  if (mode < 0 || mode > 4) return fail(LCC_ERR_ARG, "mode must be 0..4");
  const bool streaming = mode == 2 || mode == 4;
  const size_t need = 4096 + (size_t)blocks * 128 + (streaming ? (size_t)4096 * 1024 * 16 : 0);
  if (scratch_bytes < need) return fail(LCC_ERR_ARG, "scratch too small: %zu bytes needed", need);
  hipStream_t st = (hipStream_t)stream;
  unsigned* ctr = (unsigned*)scratch;                       // [0] counter, [1] fails, [2] sink
  unsigned* buf = ctr + 1024;
  const bench_u32x4* src = streaming ? reinterpret_cast<const bench_u32x4*>((char*)scratch + 4096 + (size_t)blocks * 128) : nullptr;
  HIP_TRY(hipMemsetAsync(scratch, 0, 4096 + (size_t)blocks * 128, st));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  if (mode == 1) {
    for (int it = -8; it < iters; ++it) {
      if (it == 0) HIP_TRY(hipEventRecord(e0, st));
      boundary_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(buf, it);
    }
  } else if (mode >= 3) {
    static_assert(sizeof(GridSyncState) <= 3584, "GridSyncState must fit the first 3.5 KB of the scratch");
    GridSyncState* gs = (GridSyncState*)scratch;
    grid_barrier_xcd_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(gs, 8, src, ctr + 900);   // warm-up
    HIP_TRY(hipMemsetAsync(scratch, 0, 4096, st));
    HIP_TRY(hipEventRecord(e0, st));
    grid_barrier_xcd_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(gs, iters, src, ctr + 900);
  } else {
    grid_barrier_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(ctr, ctr + 1, 8, blocks, src, ctr + 2);   // warm-up
    HIP_TRY(hipMemsetAsync(scratch, 0, 64, st));
    HIP_TRY(hipEventRecord(e0, st));
    grid_barrier_bench_kernel<<<dim3(blocks), dim3(256), 0, st>>>(ctr, ctr + 1, iters, blocks, src, ctr + 2);
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  unsigned f = 0;
  HIP_TRY(hipMemcpy(&f, mode >= 3 ? &((GridSyncState*)scratch)->fail[0] : ctr + 1, 4, hipMemcpyDeviceToHost));
  *out_us = ms * 1000.f / (float)iters;
  *out_fails = (int)f;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return check_launch("lcc_debug_bench_grid_barrier");
}
// ---- decode pipeline v2 operators (decode_v2.hip) ----
extern "C" int lcc_decode_step_begin(const int32_t* slots, const int32_t* cur_tok, const int32_t* done, uint32_t* seen, int words_per_stream,
                                     const void* embed_table, void* h, float* stats, int dim, const int32_t* pos, const float* inv_freq,
                                     void* cos, void* sin, int B, void* stream) {
  if (!slots || !cur_tok || !seen || !embed_table || !h || !stats || !pos || !inv_freq || !cos || !sin) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(decode_step_begin(slots, cur_tok, done, seen, words_per_stream, (const bf16_t*)embed_table, (bf16_t*)h, stats, dim, pos, inv_freq,
                           (bf16_t*)cos, (bf16_t*)sin, B, (hipStream_t)stream), "lcc_decode_step_begin");
}
extern "C" int lcc_dgemv_norm_linear(const void* W_packed, const float* wscale, const void* h, const float* stats, const void* norm_w, float eps,
                                     const void* bias, void* C, int ldc, int M, int N, int K, int swiglu, void* stream) {
  if (!W_packed || !h || !stats || !norm_w || !C) return fail(LCC_ERR_ARG, "null pointer");
  DgArgs a; a.W = (const bf16_t*)W_packed; a.wscale = wscale; a.M = M; a.N = N; a.K = K; a.H = (const bf16_t*)h; a.stats = stats; a.n_stat = K / 16;
  a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.bias = (const bf16_t*)bias; a.C = (bf16_t*)C; a.ldc = ldc;
  if (swiglu) OP_RET(dgemv_norm_swiglu(a, (hipStream_t)stream), "lcc_dgemv_norm_linear");
  OP_RET(dgemv_norm_bf16(a, (hipStream_t)stream), "lcc_dgemv_norm_linear");
}
extern "C" int lcc_dgemv_resid(const void* W_packed, const float* wscale, const void* x, int ldx, void* h, float* stats_out, int M, int N, int K,
                               void* stream) {
  if (!W_packed || !x || !h || !stats_out) return fail(LCC_ERR_ARG, "null pointer");
  DgArgs a; a.W = (const bf16_t*)W_packed; a.wscale = wscale; a.M = M; a.N = N; a.K = K; a.X = (const bf16_t*)x; a.ldx = ldx; a.Hres = (bf16_t*)h; a.stats_out = stats_out;
  OP_RET(dgemv_resid(a, (hipStream_t)stream), "lcc_dgemv_resid");
}
extern "C" int lcc_dgemv_qkv_rope(const void* W_dec_packed, const float* wscale, const void* h, const float* stats, const void* norm_w, float eps,
                                  const void* bias,
                                  const void* cos, const void* sin, const int32_t* tok_stream, const int32_t* kv_len, void* const* kv_base,
                                  lcc_kv_layout lay, int layer, void* q_out, int n_q_heads, int M, int K, void* stream) {
  if (!W_dec_packed || !h || !stats || !norm_w || !bias || !cos || !sin || !tok_stream || !kv_len || !kv_base || !q_out) return fail(LCC_ERR_ARG, "null pointer");
  DgArgs a; a.W = (const bf16_t*)W_dec_packed; a.wscale = wscale; a.M = M; a.N = (n_q_heads + 2 * lay.n_kv_heads) * 128; a.K = K; a.H = (const bf16_t*)h;
  a.stats = stats; a.n_stat = K / 16; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.bias = (const bf16_t*)bias; a.cs = (const bf16_t*)cos; a.sn = (const bf16_t*)sin;
  a.tok_stream = tok_stream; a.kv_len = kv_len; a.kv_base = (bf16_t* const*)kv_base; a.lay = to_lay(lay); a.layer = layer; a.q_out = (bf16_t*)q_out;
  a.n_q_heads = n_q_heads;
  OP_RET(dgemv_qkv_rope(a, (hipStream_t)stream), "lcc_dgemv_qkv_rope");
}
extern "C" int lcc_dgemv_down_qkv(const void* W_down_packed, const void* x, int ldx, void* h, float* stats, int K_down,
                                  const void* W_qkv_dec_packed, const void* norm_w, float eps, const void* bias, const void* cos, const void* sin,
                                  const int32_t* tok_stream, const int32_t* kv_len, void* const* kv_base, lcc_kv_layout lay, int layer,
                                  void* q_out, int n_q_heads, int M, int hidden, uint32_t* counter, uint32_t counter_before, uint32_t* err,
                                  void* stream) {
  if (!W_down_packed || !x || !h || !stats || !W_qkv_dec_packed || !norm_w || !bias || !cos || !sin || !tok_stream || !kv_len || !kv_base ||
      !q_out || !counter || !err) return fail(LCC_ERR_ARG, "null pointer");
  DgArgs d; d.W = (const bf16_t*)W_down_packed; d.M = M; d.N = hidden; d.K = K_down; d.X = (const bf16_t*)x; d.ldx = ldx; d.Hres = (bf16_t*)h;
  d.stats_out = stats;
  DgArgs a; a.W = (const bf16_t*)W_qkv_dec_packed; a.M = M; a.N = (n_q_heads + 2 * lay.n_kv_heads) * 128; a.K = hidden; a.H = (const bf16_t*)h;
  a.stats = stats; a.n_stat = hidden / 16; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.bias = (const bf16_t*)bias; a.cs = (const bf16_t*)cos;
  a.sn = (const bf16_t*)sin; a.tok_stream = tok_stream; a.kv_len = kv_len; a.kv_base = (bf16_t* const*)kv_base; a.lay = to_lay(lay); a.layer = layer;
  a.q_out = (bf16_t*)q_out; a.n_q_heads = n_q_heads;
  OP_RET(dgemv_down_qkv(d, a, counter, counter_before + (uint32_t)(hidden / 16), err, (hipStream_t)stream), "lcc_dgemv_down_qkv");
}
extern "C" int lcc_embed_gather_bf16(const int32_t* ids, const int32_t* indirect, const int32_t* vit_index, const void* table,
                                     const void* vit_rows, void* out, int S, int dim, void* stream) {
  if (!ids || !table || !out) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(embed_gather_bf16(ids, indirect, vit_index, (const bf16_t*)table, (const bf16_t*)vit_rows, (bf16_t*)out, S, dim, (hipStream_t)stream),
         "lcc_embed_gather_bf16");
}
extern "C" int lcc_seen_set(uint32_t* seen, int words_per_stream, const int32_t* ids, const int32_t* slot_of_id, int n, void* stream) {
  if (!seen || !ids || !slot_of_id) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(seen_set(seen, words_per_stream, ids, slot_of_id, n, 0, nullptr, (hipStream_t)stream), "lcc_seen_set");
}
extern "C" int lcc_sample_greedy(const void* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream, const int32_t* stream_slot,
                                 float repetition_penalty, int thr_token, int use_thr, float thr_value, int eos_token, int eos_token2,
                                 int suppress_eos, int32_t* done, int32_t* out_tokens, int32_t* history, int hist_ld,
                                 int32_t* hist_col, float* scores_out, float* ws, void* stream) {
  if (!logits || !seen || !stream_slot || !out_tokens || (history && !hist_col)) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(sample_greedy((const bf16_t*)logits, ld, B, V, seen, words_per_stream, stream_slot, repetition_penalty, thr_token, use_thr,
                       thr_value, eos_token, eos_token2, suppress_eos, done, out_tokens, history, hist_ld, hist_col, scores_out, ws,
                       (hipStream_t)stream), "lcc_sample_greedy");
}
extern "C" int lcc_sample_topk_topp(const void* logits, int ld, int B, int V, uint32_t* seen, int words_per_stream,
                                    const int32_t* stream_slot, float repetition_penalty, int thr_token, int use_thr, float thr_value,
                                    int eos_token, int eos_token2, int suppress_eos, int32_t* done, int32_t* out_tokens,
                                    int32_t* history, int hist_ld, int32_t* hist_col, float* scores_out, float temperature, int top_k,
                                    float top_p, uint64_t seed, uint32_t* rng_ctr, void* stream) {
  if (!logits || !seen || !stream_slot || !out_tokens || (history && !hist_col)) return fail(LCC_ERR_ARG, "null pointer");
  OP_RET(sample_topk_topp((const bf16_t*)logits, ld, B, V, seen, words_per_stream, stream_slot, repetition_penalty, thr_token, use_thr,
                          thr_value, eos_token, eos_token2, suppress_eos, done, out_tokens, history, hist_ld, hist_col, scores_out,
                          temperature, top_k, top_p, seed, rng_ctr, (hipStream_t)stream), "lcc_sample_topk_topp");
}
