"""bench.py's output contract on the GPU box: exactly ONE line on stdout, a JSON record with the fields the driver
reads -- in the single-process mode and in the multi-process mode (one rank per GPU over RCCL).

The multi-process mode cannot be launched at N > 1 on a 1-GPU box, so it is driven at world size 1 through the same
code path (ORBX_BENCH_FORCE_DIST=1: torch + a 1-rank RCCL process group).  Two things broke there once and were only
found by doing this: librccl prints a version banner to stdout AFTER the record (C stdio, flushed at exit), and
initialising torch + RCCL before the extractor handle took the hardware queues its four streams rely on (a rank ran
at 105 k instead of 129 k frames/s; measured A/B: tools/dist_order.sh)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def run_bench(extra_env, extra_args=()):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29537", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    env.update(extra_env)
    # (the live-stream / drop-in / tracking side blocks are a dozen sub-processes: one dedicated test below keeps them, the
    # contract tests do not pay for them four times over -- round 4's suite took 695 s at the driver that way)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-replay",
           "--no-host-path", "--no-live-streams", "--no-tracking-path", "--pool", "2"]
    return subprocess.run(cmd + list(extra_args), capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)


@pytest.mark.timeout(700)
@pytest.mark.parametrize("mode", ["single", "rccl_world1"])
def test_stdout_is_one_json_record(gpu, mode):
    out = run_bench({"ORBX_BENCH_FORCE_DIST": "1"} if mode == "rccl_world1" else {})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, "stdout must be the record alone, got %d lines: %r" % (len(lines), out.stdout[:400])
    rec = json.loads(lines[0])
    for k in REQUIRED:
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["warmup"] == 2
    assert rec["unit"] == "frames/s" and rec["higher_is_better"] is True and rec["scaling"] == "weak"
    assert rec["dtype"] == "u8" and rec["vs_baseline"] is None and "workload" in rec["config"]
    assert rec["value"] > 0 and rec["keypoints_last_frame"][0] > 1000 and rec["matches_last_frame"][0] > 100
    r = rec["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if mode == "rccl_world1":
        assert "RCCL version" not in out.stdout  # the banner belongs on stderr
        assert rec["stats_gather"].startswith("rccl all_gather")   # the record went over an RCCL communicator (made after the timed region)


def _plain_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


@pytest.mark.timeout(900)
def test_self_launch_one_rank_per_visible_gpu(gpu):
    """`python3 bench.py --gpus N` the way the driver invokes it (no RANK/WORLD_SIZE), N = every visible GPU: bench.py
    launches its own ranks over RCCL.  On a 1-GPU box this is the plain single-process run."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpu), "--steps", "20", "--warmup", "2",
           "--no-cpu-baseline", "--no-replay", "--no-host-path", "--no-live-streams", "--no-tracking-path", "--pool", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=_plain_env(), cwd=ROOT, timeout=850)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == gpu and len(rec["keypoints_last_frame"]) == gpu
    assert all(k > 1000 for k in rec["keypoints_last_frame"]) and all(m > 100 for m in rec["matches_last_frame"])
    if gpu > 1:  # independent streams: different scenes, different counts
        assert len(set(rec["matches_last_frame"])) > 1


@pytest.mark.timeout(900)
def test_self_launch_two_ranks_real_extractor(gpu):
    """two self-launched ranks with the real extractor.  With >= 2 GPUs: one each over RCCL.  On a 1-GPU box: the
    plumbing mode (gloo, both ranks on GPU 0) -- the process layout, the stream -> rank map and the single record are
    the real ones, only the device is shared (so the figure is not a scaling number and is not used as one)."""
    env = _plain_env()
    if gpu < 2:
        env["ORBX_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--no-cpu-baseline", "--no-replay", "--no-host-path", "--pool", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=850)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:400]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["streams"] == 2
    kp, nm = rec["keypoints_last_frame"], rec["matches_last_frame"]
    assert len(kp) == 2 and min(kp) > 1000 and min(nm) > 100 and (kp[0], nm[0]) != (kp[1], nm[1])  # two different camera streams


@pytest.mark.timeout(600)
def test_c2_config_has_a_bench_line(gpu):
    out = run_bench({}, ["--config", "c2"])
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout)
    assert "640x480" in rec["metric"] and rec["config"]["name"] == "c2" and rec["keypoints_last_frame"][0] > 500
    # the TUM shape on photographs (the stand-in for configs[0]'s inputs): three 640x480 sequences, each held to the oracle
    nat = rec["natural"]
    for seq in ("camera", "brick", "grass"):
        assert nat[seq]["parity_ok"] is True and nat[seq]["fps"] > 50000 and nat[seq]["keypoints_last_frame"] > 800
    assert nat["grass"]["cells_retry"] < 0.05 < nat["camera"]["cells_retry"]


@pytest.mark.timeout(600)
def test_default_record_carries_the_side_blocks_under_one_budget(gpu):
    """the record the DRIVER gets (no --no-* flags but the CPU baseline's): the side blocks run under one wall-clock budget
    and say what they took; SURVEY.md 8(d)'s PCIe-inclusive figures and the timed-region stamp ride in `config` as scalars;
    the drop-in classes' Tracking-shaped loop is in the record and made no device state per matcher object."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--pool", "4", "--side-budget", "120"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=_plain_env(), cwd=ROOT, timeout=550)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout)
    sb = rec["side_blocks"]
    assert sb["budget_s"] == 120 and sum(sb["spent_s"].values()) <= 150 and set(sb["spent_s"]) >= {"live_streams", "dropin_classes", "tracking_path", "host_path"}
    c = rec["config"]
    assert c["io_in_timed_region"] is False and c["timed_region_version"] == 2 and c["closing_barrier_inside_interval"] is False
    assert c["host_inclusive_fps_pinned"] > 20000 and c["host_inclusive_fps_pageable"] > 20000 and rec["value"] > c["host_inclusive_fps_pinned"]
    d = rec["dropin_classes"]
    assert d["steady_state"]["none_made"] is True and 0 < d["median_motion_model_frame_ms"] < 1.5 == (c["dropin_classes_tracking_frame_ms_median"] < 1.5) * 1.5
    assert rec["live_streams"]["one_robot"]["track"]["ms_median"] < 0.5
    assert rec["parity_check"]["ok"] is True
    # round 6: the protocol figure at the top level, the natural-statistics block, the per-rank self-diagnosis
    assert rec["value_host_inclusive"] == c["host_inclusive_fps_pinned"] and rec["value_host_inclusive_pageable"] == c["host_inclusive_fps_pageable"]
    nat = rec["natural"]
    for seq in ("retina_pan", "mosaic", "hubble"):
        assert nat[seq]["parity_ok"] is True and nat[seq]["fps"] > 50000 and 0 <= nat[seq]["cells_retry"] <= 1 and nat[seq]["k_fast_avg_launch_ms"] > 0
    assert nat["retina_pan"]["cells_retry"] > 0.9 > 0.01 > nat["synthetic_headline_frames"]["cells_retry"]
    pr = rec["per_rank"]
    assert len(pr) == 1 and pr[0]["device"] == 0 and pr[0]["parity_ok"] is True and len(pr[0]["pci_bus_id"]) >= 12
    assert 1000 < pr[0]["shader_clock_mhz_after_region"] < 3000 and abs(pr[0]["fps"] - rec["value"]) < 1e-6 * rec["value"]
    assert rec["per_rank_fps"]["distinct_devices"] == 1 and rec["per_rank_fps"]["spread"] == 0


@pytest.mark.timeout(300)
def test_side_budget_of_zero_skips_every_side_block_and_keeps_the_headline(gpu):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-replay", "--pool", "2", "--side-budget", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=_plain_env(), cwd=ROOT, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout)
    assert rec["value"] > 50000 and rec["parity_check"]["ok"] is True
    assert set(rec["side_blocks"]["skipped"]) >= {"live_streams", "dropin_classes", "tracking_path", "host_path"}
    assert "live_streams" not in rec and "host_path" not in rec and rec["config"]["host_inclusive_fps_pinned"] is None
