// Vision tower of the engine (HF modeling_qwen2_vl.py: PatchEmbed 251-274, Qwen2VLVisionBlock 425-449 x depth, PatchMerger 277-290,
// visual.forward 700-729) as one launch sequence: patchify + normalise -> patch-embed GEMM -> per block [LN, qkv GEMM, RoPE + V
// transpose, per-slice attention, proj GEMM + residual, LN, fc1 + quick_gelu, fc2 + residual] -> merger.
#include "engine_internal.h"

// ------------------------------------------------------------------------------------------------
// ViT
// ------------------------------------------------------------------------------------------------
static int g_vit_fused_qkv = [] { const char* v = getenv("LCC_VIT_FUSED_QKV"); return v ? atoi(v) : 1; }();
extern "C" int lcc_debug_set_vit_fused_qkv(int on) {      // returns the previous value; -1 = refused (a model-level call is in flight)
  if (lcc_knob_guard("lcc_debug_set_vit_fused_qkv")) return -1;
  const int old = g_vit_fused_qkv; g_vit_fused_qkv = on ? 1 : 0; return old;
}

extern "C" int lcc_engine_set_vit_grid_cap(lcc_engine* e, int cap) {
  if (!e || cap < 0) return fail(LCC_ERR_ARG, "bad engine / cap");
  e->vit_grid_cap = cap;
  return 0;
}
namespace {
struct GridCapScope {      // the cap is host-side launch state of THIS thread (gemm.hip: thread_local): it holds for the launches made inside this call only
  int prev;
  explicit GridCapScope(int cap) : prev(get_grid_cap()) { set_grid_cap(cap); }
  ~GridCapScope() { set_grid_cap(prev); }
};
}  // namespace

extern "C" int lcc_vit_encode(lcc_engine* e, int n_clips, const lcc_clip* clips, const float mean255[3], const float std255[3],
                              const float* rope_cos, const float* rope_sin, void* out_embeds, void* stream) {
  LCC_TRY(ensure_ready(e));
  CallScope in_flight;
  std::lock_guard<std::mutex> lk(e->mu_vit);
  GridCapScope cap_scope(e->vit_grid_cap);
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
  // V epilogue of the fused q|k|v projection: where the 4 patches 4i .. 4i+3 land inside a (head, channel) plane of vt
  std::vector<int32_t> grp_off((size_t)(P + 3) / 4);
  for (size_t i = 0; i < grp_off.size(); ++i) {
    const int p = (int)i * 4, sg = seg_of_patch[std::min(p, P - 1)], kl = p - seg_start[sg];
    grp_off[i] = (seg_blk[sg] + (kl >> 5)) * (80 * 32) + (kl & 31);
  }

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
  int32_t *d_seg_start, *d_seg_len, *d_seg_blk, *d_seg_of_patch, *d_tile_seg, *d_tile_q0, *d_grp_seg, *d_grp_q0, *d_g8_seg, *d_g8_q0, *d_grp_off;
  const int n_groups8 = (int)g8_seg.size();
  if (!mw.put(seg_start.data(), n_seg, &d_seg_start) || !mw.put(seg_len.data(), n_seg, &d_seg_len) ||
      !mw.put(seg_blk.data(), n_seg, &d_seg_blk) || !mw.put(seg_of_patch.data(), P, &d_seg_of_patch) ||
      !mw.put(tile_seg.data(), n_tiles, &d_tile_seg) || !mw.put(tile_q0.data(), n_tiles, &d_tile_q0) ||
      !mw.put(grp_seg.data(), n_groups, &d_grp_seg) || !mw.put(grp_q0.data(), n_groups, &d_grp_q0) ||
      !mw.put(g8_seg.data(), n_groups8, &d_g8_seg) || !mw.put(g8_q0.data(), n_groups8, &d_g8_q0) || !mw.put(grp_off.data(), (int)grp_off.size(), &d_grp_off))
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
  // q|k|v projection with RoPE + V transpose in its epilogue: head_dim 80, shapes of the 8-wave GEMM (K % 64 == 0, > 64 patches); every
  // segment is a multiple of 4 patches (H, W multiples of 28).  lcc_debug_set_vit_fused_qkv(0) / LCC_VIT_FUSED_QKV=0: separate launches.
  // (16-byte aligned cos / sin tables: the epilogue DMAs them into LDS; an offset view of the tables takes the separate launches -- ADVICE r4)
  const bool fused_qkv = g_vit_fused_qkv != 0 && e->vit_hd == 80 && gemm_vit_qkv_eligible(P, E, E) && ((((uintptr_t)rope_cos | (uintptr_t)rope_sin) & 15) == 0);
  for (int l = 0; l < e->c.vit_depth; ++l) {
    const VitLayerW& L = e->vit[l];
    if (e->vit_over) HIP_TRY(hipMemcpyAsync(x, e->vit_over + (size_t)l * tap_stride, (size_t)P * E * 2, hipMemcpyDeviceToDevice, st));
    LCC_TRY(layernorm_bf16(x, L.ln1_w, L.ln1_b, xn, P, E, 1e-6f, st));
    g = GemmArgs(); g.w_packed = 1; g.A = xn; g.lda = E; g.W = L.qkv_w; g.ldw = E; g.bias = L.qkv_b; g.C = qkv; g.ldc = 3 * E; g.M = P; g.N = 3 * E; g.K = E;
    if (fused_qkv && L.qkv_w_rope != nullptr && L.qkv_b_rope != nullptr) {
      // round 4: the projection's epilogues rotate q, k and write V blocked-transposed themselves (gemm.hip: vit_qk_epilogue / vit_v_epilogue):
      // over the weight copy in the rotation-pair row order (`vit.<i>.qkv_w_rope` / `qkv_b_rope`; V rows = rows 2E.. of the packed copy).  The V columns of `qkv` are not written.
      g.W = L.qkv_w_rope; g.bias = L.qkv_b_rope;
      g.vq.cs = rope_cos; g.vq.sn = rope_sin; g.vq.grp_off = d_grp_off; g.vq.vt = vt; g.vq.total_blocks = blocks; g.vq.E = E;
      if ((E & 127) == 0) {      // q, k and V column tiles never share a 256-column block: one launch
        g.epilogue = GEMM_EPI_VIT_QKV;
        LCC_TRY(gemm_bf16(g, st));
      } else {
        g.N = 2 * E; g.epilogue = GEMM_EPI_VIT_QK;
        LCC_TRY(gemm_bf16(g, st));
        g.W = L.qkv_w_rope + (size_t)2 * E * E; g.bias = L.qkv_b_rope + 2 * E; g.N = E; g.C = nullptr; g.epilogue = GEMM_EPI_VIT_V;
        LCC_TRY(gemm_bf16(g, st));
      }
    } else {
      LCC_TRY(gemm_bf16(g, st));
      LCC_TRY(vit_rope_vt_bf16(qkv, rope_cos, rope_sin, d_seg_of_patch, d_seg_start, d_seg_blk, vt, P, heads, blocks, st));
    }
    // 32x32x16 kernel (8 waves x 32 rows per block) once its grid fills the chip: 8 streams' chunks = 768 blocks, 170 vs 359 us per
    // block of the tower; ONE 2-frame chunk is only 6 groups x 16 heads = 96 blocks (49 us) -- there the 16-row-per-wave LDS-shared
    // kernel with twice the blocks stays (41 us)
    // (the 4-wave form of the 32x32x16 kernel -- 128-row groups, 192 blocks for one chunk -- measured the same as the 16-row kernel:
    // 261.6 vs 262.1 tokens/s without prefetch, profiles/r03/knob_sweeps_call11_13.txt; off unless LCC_VIT32_MIN_BLOCKS4 says otherwise)
    // Round 6: that instantiation had been compiled in the AGPR form (118 accumulator copies per region, tools/audit_agpr_copies.py); in the
    // VGPR form it beats the 16-row kernel on one chunk (192 blocks): tower 5.20 -> 5.02 ms (profiles/r06/tower_4wave_vgprform_ab.jsonl), so
    // grids of >= 128 four-wave blocks that are too small for the 8-wave form now take it.
    static const int vit32_min4 = [] { const char* v = getenv("LCC_VIT32_MIN_BLOCKS4"); return v ? atoi(v) : 128; }();
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
