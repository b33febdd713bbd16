// Model-level engine: the forward/generate arithmetic of `Qwen2VLForConditionalGeneration` (HF
// modeling_qwen2_vl.py: visual.forward 700-729, Qwen2VLModel.forward 1144-1204, Qwen2VLTextModel.forward 762-844,
// lm_head 1320-1323, GenerationMixin._sample utils.py:2783-2960) as one C++ launch sequence per call over
// the HIP kernels, with every per-stream state (KV, lengths, rope position, seen-id bitmap, generated ids)
// resident in HBM.  Host code only enqueues; the decode loop runs n steps without a host round trip.
//
// Memory comes from the caller (PyTorch allocations): weights, one KV arena per stream slot, an activation
// workspace, a small device state block, and a pinned-host + device "meta" ring through which the per-call
// integer tables (ids, positions, tile tables) travel in ONE async copy per call.
#include "engine_internal.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int lcc_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int lcc_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return lcc_fail(LCC_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

std::atomic<int> g_calls_in_flight{0};
int lcc_knob_guard(const char* name) {
  const int n = g_calls_in_flight.load(std::memory_order_acquire);
  if (n > 0) return lcc_fail(LCC_ERR_STATE, "%s refused: %d model-level call(s) in flight (the knob is process-global launch-routing state)", name, n);
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

size_t lcc_engine::llm_ws_bytes() const {
  const size_t S = lim.max_new_rows, B = lim.max_slots;
  const size_t H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size;
  size_t t = 0;
  t += align_up(S * H * 2) * 2;              // h, xn
  t += align_up(S * qkvd * 2);               // qkv
  t += align_up(S * qd * 2) * 2;             // q, attn
  t += align_up(S * I * 2);                  // act
  t += align_up(S * 64 * 2) * 2;             // cos, sin
  t += align_up(std::max((size_t)MAX_SPLIT * 64 * std::max<size_t>(qkvd, H), (size_t)6 * std::min<size_t>(S, 4096) * H) * 4);  // split-K slabs (up to 6 of S x H floats)
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
  const size_t vit = (P + P / 4 + 8 * (P / 16 + 64) + 128) * 4;     // + P / 4: grp_off of the fused q|k|v projection
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
  return align_up((P + P / 4 + 8 * (P / 16 + 64) + 128) * 4 + 4096, 4096) * 2;
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
    { auto iw = e->w.find(p + "qkv_w_rope"), ib = e->w.find(p + "qkv_b_rope");
      L.qkv_w_rope = iw == e->w.end() ? nullptr : (const bf16_t*)iw->second; L.qkv_b_rope = ib == e->w.end() ? nullptr : (const bf16_t*)ib->second; }
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
int lcc_ensure_ready(lcc_engine* e) {
  if (!e) return fail(LCC_ERR_ARG, "null engine");
  if (!e->ws) return fail(LCC_ERR_STATE, "buffers not bound");
  if (!e->weights_resolved) {
    std::string m;
    if (resolve_weights(e, &m) != 0) return fail(LCC_ERR_STATE, "missing weights: %.400s", m.c_str());
  }
  return 0;
}

int meta_begin(lcc_engine* e, MetaWriter* mw) {
  const int s = e->meta_next;
  e->meta_next = (s + 1) % META_RING;
  if (e->meta_ev_used[s]) HIP_TRY(hipEventSynchronize(e->meta_ev[s]));
  mw->e = e; mw->slot = s; mw->host = e->meta_host + (size_t)s * e->meta_slot_bytes; mw->dev = e->meta_dev + (size_t)s * e->meta_slot_bytes;
  mw->off = 0; mw->cap = e->meta_slot_bytes;
  return 0;
}
int meta_commit(MetaWriter* mw, hipStream_t st) {
  if (mw->off == 0) return 0;
  HIP_TRY(hipMemcpyAsync(mw->dev, mw->host, mw->off, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(mw->e->meta_ev[mw->slot], st));
  mw->e->meta_ev_used[mw->slot] = true;
  return 0;
}

extern "C" int lcc_slot_reset(lcc_engine* e, int slot, void* stream) {
  if (!e || slot < 0 || slot >= e->lim.max_slots) return fail(LCC_ERR_ARG, "bad slot");
  if (!e->state) return fail(LCC_ERR_STATE, "buffers not bound");
  std::lock_guard<std::mutex> lk(e->mu_llm);
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
  std::lock_guard<std::mutex> lk(e->mu_llm);
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
  std::lock_guard<std::mutex> lk(e->mu_llm);
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
