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
