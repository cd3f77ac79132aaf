"""examples/multi_robot on the GPU box: the reference's multi-robot main loop (MultipleRobotsScenario/Examples/Monocular/
mono_kitti.cc:83-125 -- one thread per robot, one frame per robot per iteration) on the C ABI, with the RCCL all-gather of
the robots' counters.  The program itself holds no oracle; what is checked here is that every way of driving the same
frames through the library gives the SAME results (the per-robot checksum covers keypoint counts, descriptor words and
match-table entries of every frame) -- bit-exactness against the oracle of the same entry points is
tests/test_gpu_tracking.py::test_live_stream_chain_attached_to_the_extractor."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    import __graft_entry__ as ge
    exe = ge.build_examples()
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([exe, "--json", "--frames", "40", "--warmup", "6", "--interval", "15"] + [str(a) for a in args],
                         capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_one_robot_every_mode_and_the_same_results_however_driven(gpu):
    base = _run("--mode", "track", "--w", 640, "--h", 480, "--nfeat", 1000)
    assert base["keypoints_mean"] > 900 and base["matches_mean"] > 300 and base["frames_per_s"] > 500
    assert base["rccl_allgathers"] >= 2 and base["gathered_frames"] == 40   # the final gather sees every timed frame
    for extra, env in ((["--depth", 2], None), (["--pinned", 0], None), (["--attach", 0], None), (["--depth", 2, "--attach", 0], None),
                       ([], {"ORBX_LAT_STREAMS": "1"}), ([], {"ORBX_LAT_STREAMS": "2"})):
        r = _run("--mode", "track", "--w", 640, "--h", 480, "--nfeat", 1000, *extra, env=env)
        assert r["checksum"] == base["checksum"], (extra, env)
        assert r["matches_mean"] == base["matches_mean"]
    bf = _run("--mode", "bf", "--w", 640, "--h", 480, "--nfeat", 1000)
    assert bf["matches_mean"] > 200
    ex = _run("--mode", "extract", "--w", 640, "--h", 480, "--nfeat", 1000)
    assert ex["keypoints_mean"] == base["keypoints_mean"] and ex["matches_mean"] == 0


def test_several_robots_on_one_gpu(gpu):
    """K robots = K threads, K extractor / matcher / frame-set handles on the one GPU; robot r's stream does not depend on
    who runs beside it: the XOR of the per-robot checksums of robots 0..3 run together equals that of the four run in
    two processes' worth of pairs (two cameras per call) and alone."""
    four = _run("--mode", "track", "--robots", 4, "--w", 640, "--h", 480, "--nfeat", 1000)
    assert four["gathered_frames"] == 160 and four["matches_mean"] > 300
    pairs = _run("--mode", "track", "--robots", 2, "--per-call", 2, "--w", 640, "--h", 480, "--nfeat", 1000)
    assert pairs["gathered_frames"] == 160
    assert abs(pairs["keypoints_mean"] - four["keypoints_mean"]) < 1e-9 and abs(pairs["matches_mean"] - four["matches_mean"]) < 1e-9
