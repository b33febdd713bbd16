// Private header of the engine translation units (engine.hip: object + buffers + slots, engine_vit.hip: vision tower, engine_llm.hip:
// prefill / decode, ops_abi.hip: operator-level C-ABI shims).  Not part of the C-ABI (include/livecc_amd.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/livecc_amd.h"
#include "kernels.h"
#include "grid_sync.h"

using namespace lcc;

// ---- error plumbing (defined in engine.hip; the message is thread-local: lcc_last_error) ----
int lcc_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int lcc_check_launch(const char* what);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wformat-security"
template <class... Args>
static inline int fail(int code, const char* fmt, Args... args) { return lcc_fail(code, fmt, args...); }
#pragma clang diagnostic pop
static inline int check_launch(const char* what) { return lcc_check_launch(what); }
#define HIP_TRY(x)                                                                          \
  do {                                                                                      \
    hipError_t e__ = (x);                                                                   \
    if (e__ != hipSuccess) return fail(LCC_ERR_HIP, "%s: %s", #x, hipGetErrorString(e__)); \
  } while (0)
#define LCC_TRY(x)                                                            \
  do {                                                                        \
    int r__ = (x);                                                            \
    if (r__ != 0) {                                                           \
      if (lcc_last_error()[0] == 0 || r__ != LCC_ERR_HIP) fail(r__, "%s failed (%d)", #x, r__); \
      return r__;                                                             \
    }                                                                         \
  } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct VitLayerW { const bf16_t *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  const bf16_t *qkv_w_rope = nullptr, *qkv_b_rope = nullptr;   // optional copies in the rotation-pair row order (EPI_VIT_QKV), nullptr when absent
};
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
  int vit_grid_cap = 0;                    // lcc_engine_set_vit_grid_cap: workgroup budget of the tower's tile kernels (0 = whole chip)
  const int32_t* forced = nullptr; int forced_steps = 0, forced_B = 0;   // teacher forcing (lcc_debug_set_forced_tokens)
  // host mirrors
  std::vector<int> h_kv_len, h_pos;
  std::vector<void*> h_kv_base;

  // host-side serialisation of the model-level calls (round 6): lcc_llm_prefill / lcc_llm_decode / the slot calls share the LLM workspace,
  // the meta ring and the host mirrors (mu_llm); lcc_vit_encode has its own workspace + ring (mu_vit) and may run beside them on a second
  // stream.  The Python surface additionally holds ONE lock per model across a whole generate (modeling.py); these make the C-ABI itself
  // safe for callers that do not.
  std::mutex mu_llm, mu_vit;

  size_t llm_ws_bytes() const;
  size_t vit_ws_bytes() const;
};

// Model-level calls in flight on ANY engine of this process (host threads inside lcc_vit_encode / lcc_llm_prefill / lcc_llm_decode).  The
// process-global routing knobs (lcc_debug_set_*) are read by those calls while they enqueue: changing one under a running call would mix
// two kernel families inside one forward pass, so the setters refuse with LCC_ERR_STATE while this is non-zero (engine.hip).
extern std::atomic<int> g_calls_in_flight;
struct CallScope {
  CallScope() { g_calls_in_flight.fetch_add(1, std::memory_order_acq_rel); }
  ~CallScope() { g_calls_in_flight.fetch_sub(1, std::memory_order_acq_rel); }
  CallScope(const CallScope&) = delete;
  CallScope& operator=(const CallScope&) = delete;
};
int lcc_knob_guard(const char* name);     // 0, or LCC_ERR_STATE (with message) while a model-level call is in flight

// shared between the translation units
int lcc_ensure_ready(lcc_engine* e);      // buffers bound + every weight resolved
static inline int ensure_ready(lcc_engine* e) { return lcc_ensure_ready(e); }
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
int meta_begin(lcc_engine* e, MetaWriter* mw);
int meta_commit(MetaWriter* mw, hipStream_t st);
extern int g_decode_chain;        // engine_llm.hip; lcc_slot_read_tokens turns it off after a failed hand-off
