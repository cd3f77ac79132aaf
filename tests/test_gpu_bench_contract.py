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
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-replay"]
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
