"""bench.py's contract: the single-GPU line carries `roofline` and `cpu_baseline`-shaped objects, and the multi-rank line the
diagnostics a disappointing 8-GPU run would have to be explained with (rehearsed here with two ranks time-slicing this box's
one GPU over gloo -- the 8-GPU run itself is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_two_rank_rehearsal_reports_channel_ranks_allreduce_and_rank_spread():
    # invoked PLAINLY, the way the driver runs `--gpus 1`: bench.py is its own launcher when WORLD_SIZE is unset (round 4; it
    # used to exit with "must be launched with torch.distributed.run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(AMDSPEECH_BENCH_SHARE_GPU="1", AMDSPEECH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 64
    assert line["value"] > 0 and abs(line["value"] - 2 * 32 * 1001 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    m = line["multi_gpu"]
    # the rehearsal goes through torch.distributed (two ranks cannot share one GPU in an RCCL communicator); on the real
    # node the same field reads "c-abi-rccl" and a fallback is an ERROR (AMDSPEECH_COMM=rccl, tests/test_cpu_host.py)
    assert m["device_channel"] == "torch-gloo" and line["config"]["device_channel"] == "torch-gloo"
    assert m["rccl_ranks"] is None and m["rccl_ranks_consistent"] is False
    assert m["allreduce_bytes"] == 4 * 6359808 or m["allreduce_bytes"] >= 4 * 6359632
    assert m["allreduce_ms"] > 0 and m["allreduce_ms_max_over_ranks"] >= m["allreduce_ms"] * 0.5
    assert 0 < m["ms_per_step_min_rank"] <= m["ms_per_step_max_rank"] <= line["ms_per_step"] * 1.001 + 1e-6
    assert line["roofline"]["flops_per_time_step"] > 0 and "flops_per_launch" not in line["roofline"]
    # the first real SCALE run has something to say about RAGGED data -- the same step on lengths ~U[600, T] dealt
    # by dataparallel.shard_bucketed (global buckets) and by rank-local buckets, and the beam decoder's host cores per rank
    rg = m["ragged"]
    g, loc = rg["global_buckets"], rg["rank_local_buckets"]
    for d in (g, loc):
        assert d["valid_frames_per_s"] > 0 and d["ms_per_step"] > 0
        assert 2 * 32 * 600 <= d["valid_frames_per_step"] <= 2 * 32 * 1001
        assert 0 <= d["longest_per_rank_spread"] <= d["longest_per_rank_spread_max"] <= 401
    # dealt from global buckets the ranks' longest utterances are neighbours in the sorted order; rank-local buckets are unrelated
    assert g["longest_per_rank_spread"] <= 16 < loc["longest_per_rank_spread_max"]
    assert g["frames_run_for_the_slowest_rank_per_step"] < loc["frames_run_for_the_slowest_rank_per_step"]
    bd = rg["dropin_beam_decoder"]
    assert "error" not in bd, bd
    assert len(bd["host_cores_busy_per_rank"]) == 2 and all(v > 0 for v in bd["host_cores_busy_per_rank"])
    assert len(bd["ms_per_step_per_rank"]) == 2 and all(v > 0 for v in bd["ms_per_step_per_rank"])


def test_single_gpu_ragged_section():
    """`bench.py --ragged` on one GPU: the same section with a world of one rank (spread 0), the headline untouched."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-alt",
                          "--ragged"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line["n_gpus"] == 1 and "multi_gpu" not in line
    rg = line["ragged"]
    for k in ("global_buckets", "rank_local_buckets"):
        assert rg[k]["longest_per_rank_spread"] == 0 and rg[k]["valid_frames_per_s"] > 1e6
        # a bucket stops at its longest utterance: fewer frames run than the padded 1001 of the headline step
        assert rg[k]["ms_per_step"] < line["ms_per_step"] * 1.02
    assert "error" not in rg["dropin_beam_decoder"], rg["dropin_beam_decoder"]


def test_single_gpu_line_shape_and_cfg3_extra():
    """The default command's line (short): roofline + extras incl. configs[2] timed by a child process."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line["metric"] == "audio_frames_per_sec_train_3x512_lstm_ctc" and line["n_gpus"] == 1 and line["dtype"] == "f32"
    assert line["config"]["device_channel"] == "none" and "multi_gpu" not in line
    r = line["roofline"]
    assert r["bound"] == "mfma" and 0.2 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["launch_ms"] * 1e3 / line["config"]["time_steps"] - r["avg_time_step_us"]) < 1e-3
    c3 = line["extras"]["cfg3"]
    assert "error" not in c3, c3
    assert c3["metric"] == "audio_frames_per_sec_train_5x1024_lstm_ctc" and 50 < c3["ms_per_step"] < 400
    assert "lstm_bwd_big" in c3["roofline"]["kernel"] and 0.2 < c3["roofline"]["frac"] < 1.0
    # the opt-in split-precision mode at the H = 1024 configurations: separate alt_* entries, faster than exact f32
    a3, a5 = line["extras"]["alt_bf16x3_cfg3"], line["extras"]["alt_bf16x3_cfg5_bidirectional"]
    assert "error" not in a3 and "error" not in a5, (a3, a5)
    assert "bf16x3" in a3["dtype"] and a3["ms_per_step"] < 0.8 * c3["ms_per_step"]
    assert a5["metric"] == "audio_frames_per_sec_train_5x1024_bidirectional_lstm_ctc" and a3["ms_per_step"] < a5["ms_per_step"]
    # ... and (round 4) as plain bf16 operands, one MFMA per product: its own alt_* entries, faster again, never the headline
    b3, b5 = line["extras"]["alt_bf16_cfg3"], line["extras"]["alt_bf16_cfg5_bidirectional"]
    assert "error" not in b3 and "error" not in b5, (b3, b5)
    assert "bf16 MFMA" in b3["dtype"] and "bf16x3" not in b3["dtype"] and b3["ms_per_step"] < a3["ms_per_step"] < c3["ms_per_step"]
    assert b5["ms_per_step"] < a5["ms_per_step"] and line["dtype"] == "f32"
