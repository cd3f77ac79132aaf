"""bench.py's N > 1 path without a GPU: the REAL bench.main code (launcher, rank -> stream / device map, barrier-bracketed
timed region, gather, one JSON line) over gloo, with tests/bench_stub.py standing in for the extractor.

`python3 bench.py --gpus N` is how the driver invokes the bench; round 1's version asserted out unless it ran under
torch.distributed.run.  Both launch forms are exercised here at N = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "6", "--warmup", "2", "--batch", "4", "--pool", "3", "--config", "c2", "--no-cpu-baseline"]


def _env(tmp_path, **extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ORBX_BENCH_EXTRACTOR="tests.bench_stub:StubExtractor", ORBX_DIST_BACKEND="gloo", ORBX_DIST_EXCLUSIVE="1",
               ORBX_BENCH_STUB_DEVICES="2", ORBX_BENCH_STUB_DUMP=str(tmp_path), PYTHONPATH=ROOT)
    env.update(extra)
    return env


def _check_record(out, tmp_path):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:500]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 6 and rec["warmup"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["name"] == "c2" and rec["config"]["streams"] == 2
    assert rec["keypoints_last_frame"] == [1500, 1501] and rec["matches_last_frame"] == [700, 701]  # rank order
    # whole-job frames over the SLOWEST rank's time: rank 1 sleeps 4 ms per step, rank 0 2 ms
    assert rec["ms_per_step"] >= 4.0 * 0.9
    assert abs(rec["value"] - 2 * 4 * 6 / (rec["ms_per_step"] * 6 / 1e3)) < 1e-6 * rec["value"]
    dumps = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    for r, d in enumerate(dumps):
        assert d["rank"] == r and d["device"] == r and d["world"] == 2  # one process per GPU, rank r on GPU r
        assert d["steps"] == 3 + 2 + 6 and len(d["uploads"]) == 3      # pool validation + warmup + EXACTLY the timed steps
    assert dumps[0]["uploads"] != dumps[1]["uploads"]                    # rank r feeds camera stream r, not a shared one
    return rec


@pytest.mark.timeout(300)
def test_plain_invocation_launches_its_own_ranks(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, capture_output=True, text=True,
                         env=_env(tmp_path), cwd=ROOT, timeout=280)
    _check_record(out, tmp_path)


@pytest.mark.timeout(300)
def test_under_torch_distributed_run(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS
    out = subprocess.run(cmd, capture_output=True, text=True, env=_env(tmp_path), cwd=ROOT, timeout=280)
    _check_record(out, tmp_path)


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launch(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, capture_output=True, text=True,
                         env=_env(tmp_path, ORBX_BENCH_STUB_FAIL_RANK="1"), cwd=ROOT, timeout=280)
    assert out.returncode != 0 and out.stdout.strip() == "", (out.returncode, out.stdout[:300])
    assert "rank 1 failed" in out.stderr


@pytest.mark.timeout(300)
def test_two_rank_record_carries_the_cpu_baseline_and_the_interval_stamp(tmp_path):
    """the CPU figure beside EVERY number -- the N > 1 record too (rank 0 times a bounded one-thread
    sample of the oracle after the gather; round 4 emitted it at N = 1 only)."""
    args = [a for a in ARGS if a != "--no-cpu-baseline"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True,
                         env=_env(tmp_path, ORBX_BENCH_CPU_SECONDS="1.0"), cwd=ROOT, timeout=280)
    rec = _check_record(out, tmp_path)
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "frames/s" and cb["value"] > 1 and "640x480" in cb["sample"]
    assert "one_thread_per_stream" not in cb   # the threaded leg is the N = 1 record's
    c = rec["config"]
    assert c["io_in_timed_region"] is False and c["timed_region_version"] == 2 and c["closing_barrier_inside_interval"] is False
    assert c["rccl_up_before_timed_region"] is False and rec["stats_gather"] == "gloo all_gather"


@pytest.mark.timeout(600)
def test_eight_ranks_eight_disjoint_streams_one_record(tmp_path):
    """BASELINE.json configs[4] without the hardware: `bench.py --gpus 8` the way the driver invokes it, eight ranks over
    gloo on a free port, the stub standing in for the GPUs.  Eight different camera streams on eight different devices,
    ONE record, with `roofline` and `cpu_baseline` in it (MultipleRobotsScenario/Examples/Monocular/mono_kitti.cc:80-101:
    eight robots, nothing shared but the counters)."""
    args = [a for a in ARGS if a != "--no-cpu-baseline"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + args, capture_output=True, text=True,
                         env=_env(tmp_path, ORBX_BENCH_STUB_DEVICES="8", ORBX_BENCH_CPU_SECONDS="1.0", ORBX_BENCH_STUB_PROFILE="1"), cwd=ROOT, timeout=550)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:500]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["config"]["streams"] == 8 and rec["scaling"] == "weak"
    assert rec["keypoints_last_frame"] == [1500 + r for r in range(8)] and rec["matches_last_frame"] == [700 + r for r in range(8)]
    dumps = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(8)]
    assert [d["device"] for d in dumps] == list(range(8)) and all(d["world"] == 8 for d in dumps)
    assert len({tuple(d["uploads"]) for d in dumps}) == 8            # eight disjoint streams
    assert all(d["steps"] == 3 + 2 + 6 for d in dumps)
    # whole-job frames over the slowest rank (rank 7 sleeps 16 ms per step)
    assert rec["ms_per_step"] >= 16.0 * 0.9 and abs(rec["value"] - 8 * 4 * 6 / (rec["ms_per_step"] * 6 / 1e3)) < 1e-6 * rec["value"]
    assert rec["cpu_baseline"]["value"] > 1 and rec["cpu_baseline"]["cores"] == 1
    r = rec["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert rec["stats_gather"] == "gloo all_gather"   # (the GPU box: "rccl all_gather ..." -- tests/test_gpu_bench_contract.py)
    # the self-diagnosing part of the record: one entry per rank with the device it drove, where it was pinned,
    # its own time and rate, the shader clock behind its region and its parity verdict; min / max / spread over the ranks
    pr = rec["per_rank"]
    assert [q["rank"] for q in pr] == list(range(8)) and [q["device"] for q in pr] == list(range(8))
    for q in pr:
        assert set(q) >= {"rank", "device", "pci_bus_id", "numa_node", "pinned_to_numa_node", "ms_per_step", "fps",
                          "shader_clock_mhz_after_region", "parity_ok", "keypoints_last_frame", "matches_last_frame"}
        assert q["ms_per_step"] >= 2.0 * (q["rank"] + 1) * 0.9 and abs(q["fps"] - 4 / (q["ms_per_step"] / 1e3)) < 1e-6 * q["fps"]
    f = rec["per_rank_fps"]
    assert f["min"] == min(q["fps"] for q in pr) and f["max"] == max(q["fps"] for q in pr) and 0.8 < f["spread"] < 0.9   # rank 7 is 8 x slower
    assert f["distinct_devices"] == 8


def test_world_size_must_match_gpus(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, capture_output=True, text=True,
                         env=_env(tmp_path, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT, timeout=120)
    assert out.returncode == 2 and out.stdout == "" and "WORLD_SIZE=1" in out.stderr


def test_single_rank_with_the_stub_prints_one_line(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, capture_output=True, text=True,
                         env=_env(tmp_path), cwd=ROOT, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout)
    assert rec["n_gpus"] == 1 and "torch" not in out.stderr.lower()


def test_device_of_rank():
    from orbslamm_amd import streams
    assert [streams.device_of_rank(r, 8) for r in range(8)] == list(range(8))
    with pytest.raises(ValueError):
        streams.device_of_rank(1, 1)  # one process per GPU: a second rank on a 1-GPU box is an error, not a share
    assert [streams.device_of_rank(r, 2, exclusive=False) for r in range(4)] == [0, 1, 0, 1]
    assert streams.stream_of_rank(3) == [3]
