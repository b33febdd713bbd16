"""-m gpu: the vision tower's q|k|v projection with the 2-D RoPE and the V transpose in its own epilogues (csrc/gemm.hip: vit_qk_epilogue /
vit_v_epilogue over `vit.<i>.qkv_w_rope`) against the separate-launch form (projection -> lcc_vit_rope_vt_bf16): BIT-IDENTICAL tower outputs.  The separate
form is the one every parity test against HF (Q2VL:225-248, 342-449) has exercised since round 1; it stays selectable
(lcc_debug_set_vit_fused_qkv(0) / LCC_VIT_FUSED_QKV=0)."""
import pytest
import torch

from tests.util import record

pytestmark = pytest.mark.gpu


def _model(cfg, dev, max_patches):
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    arena = WeightArena(cfg, dev).fill_random(seed=5)
    return LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=1024, max_new_rows=512, max_patches=max_patches, max_history=16)


def _clips(shapes, dev, seed):
    from livecc_amd import protocol
    return [dict(frames=torch.from_numpy(protocol.synth_frames(T, H, W, seed=seed + i, layout="TCHW")).to(dev), layout="TCHW")
            for i, (T, H, W) in enumerate(shapes)]


def _both(native, clips):
    from livecc_amd import ops
    old = ops.set_vit_fused_qkv(False)
    try:
        ops.launch_counts(reset=True)
        plain = native.engine.vit_encode(clips).clone()
        c0 = ops.launch_counts(reset=True)
        ops.set_vit_fused_qkv(True)
        fused = native.engine.vit_encode(clips).clone()
        c1 = ops.launch_counts(reset=True)
        torch.cuda.synchronize()
    finally:
        ops.set_vit_fused_qkv(bool(old))
    return plain, fused, c0, c1


# small config (E = 320, 4 heads): ragged segment lengths (not multiples of 32 keys), several clips, a patch count that is not a multiple of
# the 128-row tile, one- and multi-slice clips
@pytest.mark.parametrize("shapes", [[(2, 56, 84)], [(6, 112, 84)], [(2, 56, 56), (4, 84, 140), (2, 28, 56)], [(2, 280, 280), (2, 392, 728)]],
                         ids=["24p", "3x48p", "3clips", "400p+1456p"])
def test_fused_qkv_epilogue_is_bit_identical_to_the_separate_launches_small(dev, shapes):
    from livecc_amd.config import small
    cfg = small()
    native = _model(cfg, dev, max_patches=4096)
    plain, fused, c0, c1 = _both(native, _clips(shapes, dev, seed=11))
    patches = sum(((T + 1) // 2) * (H // 14) * (W // 14) for T, H, W in shapes)
    # up to 64 patches the projection stays on the weight-streaming kernels and the separate RoPE launch (gemm_vit_qkv_eligible)
    assert c0["gemm_vit_qkv"] == 0 and c1["gemm_vit_qkv"] == (2 * cfg.vit_depth if patches > 64 else 0), (c0, c1)      # E = 320: a q|k launch + a V launch per block
    assert torch.isfinite(fused.float()).all()
    assert torch.equal(plain, fused)


@pytest.mark.parametrize("streams", [1, 8])
def test_fused_qkv_epilogue_is_bit_identical_at_7b_shapes(dev, streams):
    """LiveCC-7B tower (E = 1280, 16 heads, 32 blocks) on the streaming chunk shape: 1 and 8 co-scheduled 2-frame 392x728 chunks
    (1,456 / 11,648 patches: the 128-row and the 256-row tile of the 8-wave kernel).  Also times both forms (evidence, not a bar)."""
    from livecc_amd.config import get_config
    cfg = get_config("livecc-7b")
    native = _model(cfg, dev, max_patches=16384)
    clips = _clips([(2, 392, 728)] * streams, dev, seed=21)
    plain, fused, c0, c1 = _both(native, clips)
    assert c0["gemm_vit_qkv"] == 0 and c1["gemm_vit_qkv"] == cfg.vit_depth         # E % 128 == 0: q, k and V tiles in ONE launch per block
    assert torch.equal(plain, fused)
    from livecc_amd import ops
    ms = {}
    for on in (False, True, False, True):
        old = ops.set_vit_fused_qkv(on)
        native.engine.vit_encode(clips)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            native.engine.vit_encode(clips)
        t1.record()
        torch.cuda.synchronize()
        ms.setdefault("fused" if on else "separate", []).append(t0.elapsed_time(t1) / 5)
        ops.set_vit_fused_qkv(bool(old))
    record(f"vit_tower_fused_qkv[{streams} streams]", dict(tower_ms_separate=ms["separate"], tower_ms_fused=ms["fused"]))


@pytest.mark.parametrize("streams,cap", [(1, 32), (1, 96), (8, 64), (8, 128)])
def test_tower_under_a_grid_cap_is_bit_identical(dev, streams, cap):
    """lcc_engine_set_vit_grid_cap (round 5): the tower's tile kernels launch at most `cap` workgroups and walk their tiles persistently (what a
    prefetched tower does under another turn's decode steps) -- the tile -> output mapping does not change, so the embeddings are the bits of
    the whole-chip launch.  LiveCC-7B tower on 1 and 8 chunks: the 4-wave LDS-DMA GEMM + 128-row 8-wave tiles + the LDS-shared vision
    attention (one chunk), the 256-row 8-wave tiles + the 32x32x16 vision attention (eight)."""
    from livecc_amd.config import get_config
    cfg = get_config("livecc-7b")
    native = _model(cfg, dev, max_patches=16384)
    clips = _clips([(2, 392, 728)] * streams, dev, seed=33)
    ref = native.engine.vit_encode(clips).clone()
    native.engine.set_vit_grid_cap(cap)
    try:
        got = native.engine.vit_encode(clips).clone()
        ms = []
        for _ in range(3):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            native.engine.vit_encode(clips)
            t1.record()
            torch.cuda.synchronize()
            ms.append(t0.elapsed_time(t1))
    finally:
        native.engine.set_vit_grid_cap(0)
    again = native.engine.vit_encode(clips)
    torch.cuda.synchronize()
    assert torch.equal(ref, got), f"{int((ref != got).sum())} of {ref.numel()} embedding values differ under a cap of {cap} workgroups"
    assert torch.equal(ref, again)
    record(f"vit_tower_grid_cap[{streams} streams, {cap} CUs]", dict(tower_ms=min(ms)))
