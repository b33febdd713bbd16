"""Host-side routing of the GEMM entry point at the engine's shapes (lcc_debug_gemm_plan: no launch, no GPU): which kernel family serves a
call and how many split-K slabs the engine's prefill asks for.  Pins the round-5 table of DESIGN section 4 so that a change of a tile
score, of a predicate or of the split rule shows up here before it shows up as a slower bench line."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, I, QKV = 3584, 18944, 4608


@pytest.fixture(scope="module")
def ops(built_lib):          # conftest: builds the git-ignored library when this file runs alone / on a fresh checkout
    from livecc_amd import ops as o
    return o


def test_eight_chunks_take_the_big_tiles(ops):
    assert ops.gemm_plan(3088, 2 * I, H, ops.EPI_SWIGLU)[0] == 272          # variable-height tiles: 12 x 148 blocks instead of 13 x 148
    assert ops.gemm_plan(3088, H, I, ops.EPI_RESIDUAL)[0] == 192            # 17 x 14 = 238 blocks fill one round
    assert ops.gemm_plan(3088, H, H, ops.EPI_RESIDUAL)[0] == 192
    assert ops.gemm_plan(3088, QKV, H, ops.EPI_NONE)[0] == 256
    assert ops.gemm_plan(9048, 2 * I, H, ops.EPI_SWIGLU)[0] in (256, 272)
    assert ops.gemm_plan(24058, 2 * I, H, ops.EPI_SWIGLU)[0] == 256         # 1504 fragments = 94 x 16: the 256-row tiles are the same thing


def test_one_chunk_takes_the_tall_kernel_and_the_small_variable_height_slabs(ops):
    assert ops.gemm_plan(386, 2 * I, H, ops.EPI_SWIGLU) == (448, 1)         # gate/up: one block row, 237 blocks
    for n, k, want in ((H, H, 6), (H, I, 6), (QKV, H, 4)):                  # o, down, q|k|v: 3 row tiles x 14 / 14 / 18 column tiles
        tile, splits = ops.gemm_plan(386, n, k, ops.EPI_NONE)
        assert splits == want, (n, k, splits)
        assert ops.gemm_plan(386, n, k, ops.EPI_NONE, nsplit=splits)[0] == 144
    assert ops.gemm_plan(386, H, I, ops.EPI_NONE, nsplit=4)[0] == 128       # with four slabs the 128-row tiles keep the call (224 blocks beat 168)
    assert ops.gemm_plan(386, H, I, ops.EPI_NONE, w_fp8=True) == (0, 4)     # fp8 weights: the round-4 split counts


def test_decode_batches_stream_the_weights(ops):
    for m in (1, 8, 16, 32, 64):
        assert ops.gemm_plan(m, 2 * I, H, ops.EPI_SWIGLU)[0] == 16
    assert ops.gemm_plan(65, 2 * I, H, ops.EPI_SWIGLU)[0] != 16


def test_other_row_counts_keep_their_round_4_routes(ops):
    assert ops.gemm_plan(1131, 2 * I, H, ops.EPI_SWIGLU)[0] == 256          # the first turn: 5 row tiles, the last half skipped
    assert ops.gemm_plan(122, H, I, ops.EPI_NONE)[1] >= 1                   # 8 fragments: no small class (a multiple of 8), any split count is legal
    assert ops.gemm_plan(122, H, I, ops.EPI_NONE, nsplit=4)[0] != 144
    assert ops.gemm_plan(1456, 1280, 5120, ops.EPI_RESIDUAL)[0] == 64       # one-stream vision fc2: the 4-wave 64 x 128 tiles (230 blocks)
    assert ops.gemm_plan(11648, 5120, 1280, ops.EPI_QUICK_GELU)[0] in (256, 272)


def test_the_switch_restores_the_128_row_slabs(built_lib):
    code = ("import sys; sys.path.insert(0, %r)\nfrom livecc_amd import ops\n"
            "s = ops.gemm_plan(386, 3584, 18944, ops.EPI_NONE)[1]\nprint(s, ops.gemm_plan(386, 3584, 18944, ops.EPI_NONE, nsplit=s)[0])" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, LCC_GEMM_VH_SMALL="0"), timeout=120)
    assert r.returncode == 0, r.stderr[-400:]
    assert r.stdout.strip() == "4 128", r.stdout      # four slabs on the 128-row tiles: the round-4 form


def test_prefill_attention_plan_at_the_engines_shapes(built_lib):
    """Tile height x key-split count of the LLM prefill attention (lcc_debug_attn_plan: the function lcc_llm_prefill itself calls, no launch):
    one block per CU and (tile, KV head, split), relative time = rounds of blocks x keys per block; measured in
    profiles/r06/attn_tall_tiles_ab.jsonl and attn_tall_tiles_splits_probe.jsonl."""
    from livecc_amd import _lib, ops
    assert ops.attn_tile_rows(28, 4) == 36 and ops.attn_tile_rows(64, 8) == 32 and ops.attn_tile_rows(12, 2) == 42 and ops.attn_tile_rows(16, 1) == 32
    plan = lambda n_new, kv, hq=28, hkv=4, cus=256: ops.attn_plan(n_new, kv, hq, hkv, cus)      # noqa: E731
    assert plan([386], 6586) == (36, 5)              # one streaming chunk: 11 tiles x 4 KV heads x 5 splits = 220 blocks, one round (65 us; (32, 4): 73)
    assert plan([386] * 2, 6586) == (32, 2)          # 208 blocks, one round: 113.7 us ((36, 8) 126, (32, 7) 129: what "fill the last round" chose)
    assert plan([386] * 3, 6586) == (32, 1)          # 1,158 rows: no split above 1,024 rows (the partial buffers); 156 / 132 blocks: one round, a tie
    assert plan([386] * 4, 6586) == (32, 1)          # 208 blocks unsplit, a tie with the tall tile
    assert plan([386] * 5, 6586) == (36, 1)          # 260 blocks = two rounds, 220 tall ones = one
    assert plan([300] * 3, 6586) == (32, 2)          # 900 rows may split: 30 tiles x 4 x 2 = 240 blocks of half the keys, one round
    assert plan([386] * 8, 6586) == (32, 1)          # 416 / 352 blocks: two rounds either way
    assert plan([1131] * 8, 1131) == (36, 1)         # 1152 -> 1024 blocks: five rounds -> four
    assert plan([4096], 12288) == (32, 1)            # a one-shot piece: 512 / 456 blocks, two rounds either way
    assert plan([386], 200) == (32, 1)               # too few keys to split (>= 8 key tiles per split): 52 / 44 blocks, one round: a tie
    assert plan([386], 26000) == (36, 5)             # a long history changes nothing for one chunk
    assert plan([386], 6586, 64, 8) == (32, 2)       # 72B heads (G = 8): the tall tile does not exist; 13 x 8 x 2 = 208 blocks
    assert plan([386], 6586, 32, 2) == (0, 0)        # 16 query heads per KV head: another kernel family
    with pytest.raises(_lib.LccError):
        plan([386, 0], 6586)
    with pytest.raises(_lib.LccError):
        plan([386], 6586, 28, 5)
