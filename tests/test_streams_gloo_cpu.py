"""world_size-2 CPU test (gloo) of the N>1 path bench.py takes on the GPU box:
independent streams per rank, barrier-bracketed timed region, all_gather of the
statistics record, MAX-over-ranks time, whole-job aggregation."""
import json
import os
import socket
import time

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from orbslamm_amd import streams
    r, w, lr = streams.init("gloo")
    assert (r, w) == (rank, world)
    mine = streams.stream_of_rank(rank)
    counter = {"n": 0}

    def step():
        counter["n"] += 1
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow one

    dt = streams.timed_region(step, 5, lambda: None, world)
    assert counter["n"] == 5
    gathered, dt_max = streams.gather_stats((64 * 5, 2000 + rank, 900 + rank, dt), world)
    fps, total = streams.aggregate(gathered, dt_max)
    with open(os.path.join(outdir, "rank%d.json" % rank), "w") as f:
        json.dump(dict(streams=mine, gathered=gathered, dt=dt, dt_max=dt_max, fps=fps, total=total), f)
    streams.finalize(world)


def test_two_rank_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert res[0]["streams"] == [0] and res[1]["streams"] == [1]  # disjoint streams, no sharing
    for r in res:
        assert r["total"] == 640  # frames of BOTH ranks
        assert [g[1] for g in r["gathered"]] == [2000.0, 2001.0] and [g[2] for g in r["gathered"]] == [900.0, 901.0]
        assert r["dt_max"] >= max(x["dt"] for x in res) - 1e-6  # MAX over ranks, not the local time
        assert abs(r["fps"] - 640 / r["dt_max"]) < 1e-6
    # the barrier on both sides makes every rank's window cover the slowest rank's 5 steps
    # a rank's own interval ends at ITS sync (the closing barrier follows outside it): the fast rank reads its own time, the
    # job's time is the MAX over ranks and every rank holds the same one
    assert res[0]["dt"] >= 5 * 0.01 * 0.9 and res[1]["dt"] >= 5 * 0.02 * 0.9 and res[0]["dt"] < res[1]["dt"]
    assert all(r["dt_max"] >= 5 * 0.02 * 0.9 for r in res) and abs(res[0]["dt_max"] - res[1]["dt_max"]) < 1e-9


def test_single_process_is_a_noop():
    from orbslamm_amd import streams
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    assert streams.env_rank()[:2] == (0, 1)
    g, dt = streams.gather_stats((10, 1, 2, 0.5), 1)
    assert g == [[10.0, 1.0, 2.0, 0.5]] and dt == 0.5
    assert streams.aggregate(g, dt) == (20.0, 10.0)
