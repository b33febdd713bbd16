"""bench.py's launcher and report logic without a GPU: `--gpus N` starts N ranks by itself (one process per GPU under
torch.distributed.run; here gloo ranks with a stand-in model), refuses a launcher/--gpus mismatch, and the parity / roofline
helpers compute what they say."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--standin", "--steps", "1", "--warmup", "0", "--config", "tiny", "--frames", "10", "--height", "56", "--width", "84"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


@pytest.mark.timeout(300)
def test_gpus_2_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, printed by rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["launcher"] == "self" and out["scaling"] == "weak"
    assert out["config"]["streams"] == 2 and len(out["tokens_per_s_per_rank"]) == 2
    # 10 frames = 1 + 2 turns of 16 tokens per stream, 2 streams, one step
    assert out["value"] == pytest.approx(2 * 3 * 16 / (out["ms_per_step"] / 1e3), rel=1e-3)
    # N > 1: every rank also runs its 8-stream share and the line carries the north_star job (BASELINE.json configs[2] scaled to N GPUs)
    c2 = out["configs2"]
    assert "configs2_share" not in out and c2["streams"] == 16 and c2["streams_per_gpu"] == 8 and c2["n_gpus"] == 2
    assert len(c2["tokens_per_s_per_rank"]) == 2 and c2["data_path_collectives"] == 0 and "xgmi_bound_s" in c2["weight_broadcast"]
    assert c2["tokens_per_s_per_stream"] == pytest.approx(min(c2["tokens_per_s_per_rank"]) / 8, rel=1e-2)
    assert c2["value"] == pytest.approx(16 * 3 * 16 * c2["steps"] / (c2["ms_per_replay"] * c2["steps"] / 1e3), rel=1e-2)
    assert c2["roofline"]["peak"] == 5000.0 and c2["roofline"]["bound"] == "mfma"
    assert c2["stream_ids_rank0"] == [0, 2, 4, 6, 8, 10, 12, 14] and c2["stream_id_sum_per_rank"] == [56, 64]       # stream s -> rank s % 2


@pytest.mark.timeout(600)
def test_gpus_8_runs_the_north_star_job_shape_on_gloo():
    """the first real 8-GPU run must have no first-time code.  `bench.py --gpus 8 --standin` = 8 gloo ranks: 64 streams,
    stream s on rank s % 8, 8-way gather of the per-rank counters, one broadcast, no collective in the data path, `configs2.streams == 64`."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"] + SMALL, env=dict(_env(), OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, printed by rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["config"]["streams"] == 8 and len(out["tokens_per_s_per_rank"]) == 8
    assert len(out["decode_step_ms_per_rank"]) == 8 and len(out["weight_broadcast_s_per_rank"]) == 8
    c2 = out["configs2"]
    assert c2["streams"] == 64 and c2["streams_per_gpu"] == 8 and c2["n_gpus"] == 8 and c2["data_path_collectives"] == 0
    assert "BASELINE.json configs[2]" in c2["workload"] and "scaled" not in c2["workload"]
    assert c2["stream_ids_rank0"] == [0, 8, 16, 24, 32, 40, 48, 56]                                  # s % 8 == 0
    assert c2["stream_id_sum_per_rank"] == [8 * r + 224 for r in range(8)]                           # rank r holds {r, r + 8, ..., r + 56}
    assert len(c2["tokens_per_s_per_rank"]) == 8 and all(x > 0 for x in c2["tokens_per_s_per_rank"])
    assert c2["tokens_per_s_per_stream"] == pytest.approx(min(c2["tokens_per_s_per_rank"]) / 8, rel=1e-2)
    assert c2["value"] == pytest.approx(64 * 3 * 16 * c2["steps"] / (c2["ms_per_replay"] * c2["steps"] / 1e3), rel=1e-2)
    assert c2["roofline"]["peak"] == 8 * 2500.0 and len(c2["weight_broadcast"]["seconds_per_rank"]) == 8
    assert out["value_no_prefetch"] is not None and out["value_no_prefetch"] > 0


@pytest.mark.timeout(300)
def test_one_failing_rank_does_not_hang_the_job():
    """a rank that throws while it builds its 8-stream share (HBM, first replay) used to leave the healthy ranks blocked in
    the share's next collective for ever.  The ranks now agree on a success flag before the first collective of the share: every rank
    leaves with an error block and the main line is still printed."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=dict(_env(), LCC_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "error" in out["configs2"] and out["configs2"]["ranks_failed"] == 1


@pytest.mark.timeout(120)
def test_launcher_world_size_mismatch_is_refused():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_parity_report_and_kv_schedule():
    sys.path.insert(0, ROOT)
    import bench
    from livecc_amd import protocol
    from livecc_amd.config import tiny
    rng = np.random.RandomState(0)
    l32 = rng.randn(2, 4, 64).astype(np.float32) * 3
    l16 = l32 + rng.randn(2, 4, 64).astype(np.float32) * 0.02
    nat = l32 + rng.randn(2, 4, 64).astype(np.float32) * 0.02
    tok = l16.argmax(-1)
    rep = bench.parity_report(tok, nat, dict(logits=l16, own_argmax=tok), dict(logits=l32, own_argmax=l32.argmax(-1)))
    assert rep["tokens_equal"] == 8 and rep["tokens_total"] == 8 and rep["turns_compared"] == 2
    assert 0 < rep["rel_dlogit_vs_bf16"] < 0.05 and 0.3 < rep["err_ratio_vs_fp32"] < 3.0
    assert rep["tokens_equal_where_decided"] <= rep["tokens_decided_by_margin"] <= 8
    cfg = tiny()
    kv = bench.kv_lengths_of_decode_steps(cfg, 10, 56, 84, 4, protocol)
    b = protocol.TurnBuilder(cfg, seed=0)
    n0 = len(b.turn_ids(0, protocol.num_video_tokens(protocol.grid_of(6, 56, 84, cfg), cfg)))
    assert kv[:3] == [n0 + 1, n0 + 2, n0 + 3] and len(kv) == 3 * 3


@pytest.mark.timeout(300)
def test_the_drivers_own_torchrun_command_line():
    """The driver launches N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`: the ranks must take WORLD_SIZE from that launcher (no second self-launch)
    and rank 0 alone prints the one JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["launcher"] == "torchrun" and out["steps"] == 1 and out["warmup"] == 0
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                "cpu_baseline"):
        assert key in out, key
