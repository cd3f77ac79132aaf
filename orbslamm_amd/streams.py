"""Multi-GPU layout of the hot path: independent camera streams, one per GPU
(MultipleRobotsScenario: one System/Tracking thread per robot,
/root/reference/MultipleRobotsScenario/Examples/Monocular/mono_kitti.cc:83-98).

There is no data-path collective: a stream's frames, pyramid and previous-frame
descriptors never leave its GPU.  torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) is used for exactly two things:
the barrier/MAX around the timed region and one all_gather of a small statistics
record per rank.

Backend "nccl": the RCCL communicator is made AFTER the timed region, for the gather it exists for.  A rank that brings
RCCL up before its steps runs them 4 % slower (148-150 k against 154-156 k frames/s in a 20-step run on an MI355X; torch
and a CPU process group alone cost 2.7 % against 159-160 k for a process without torch) -- the communicator's own streams
and proxy take part in the GPU's queue rotation although the data path never uses them -- and the driver would read that
as scaling loss of a path that has no collective.  So the barriers around the timed region go over a gloo group (a
barrier is a barrier), and the statistics record is gathered over RCCL / xGMI as SURVEY.md 8e says.
ORBX_DIST_EAGER_NCCL=1 brings the communicator up first (the old order: the A/B)."""
import os

STATS_FIELDS = ("frames", "keypoints_last", "matches_last", "seconds")
_state = {"stats_backend": None, "stats_group": None, "device": None}


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend, device=None):
    """returns (rank, world, local_rank); no-op for a single process"""
    import torch.distributed as dist
    rank, world, local_rank = env_rank()
    # ORBX_BENCH_FORCE_DIST=1: a 1-rank process group, to exercise the N > 1 initialisation order on a 1-GPU box
    force = os.environ.get("ORBX_BENCH_FORCE_DIST") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one node, ranks meet on the loopback: gloo need not resolve the container's hostname to pick an interface
        if os.environ.get("MASTER_ADDR") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        kw = {}
        eager = os.environ.get("ORBX_DIST_EAGER_NCCL") == "1"
        if backend == "nccl" and not eager:
            # barriers over gloo now, the RCCL group when the record is gathered (see the head of this file)
            _state.update(stats_backend="nccl", device=device)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def _stats_group():
    """the group the statistics record is gathered over: the default one, or -- backend "nccl" -- an RCCL group of all
    ranks made at first use"""
    import torch.distributed as dist
    if _state["stats_backend"] != "nccl":
        return None
    if _state["stats_group"] is None:
        kw = {}
        if _state["device"] is not None:
            kw["device_id"] = _state["device"]
        try:
            _state["stats_group"] = dist.new_group(backend="nccl", **kw)
        except TypeError:   # a torch without new_group(device_id=...)
            _state["stats_group"] = dist.new_group(backend="nccl")
    return _state["stats_group"]


def stream_of_rank(rank, streams_per_rank=1):
    """camera streams owned by a rank: stream s lives on GPU s mod world"""
    return [rank * streams_per_rank + i for i in range(streams_per_rank)]


def device_of_rank(local_rank, ndev, exclusive=True):
    """GPU of a rank: one process per GPU (stream s -> GPU s mod 8).  exclusive=False is the plumbing mode
    of the tests (more ranks than GPUs, backend gloo): ranks wrap around the visible devices."""
    if exclusive:
        if ndev > 0 and local_rank >= ndev:
            raise ValueError("rank %d has no GPU of its own (%d visible)" % (local_rank, ndev))
        return local_rank
    return local_rank % max(ndev, 1)


def barrier(world):
    if world > 1 or os.environ.get("ORBX_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()


def timed_region(step, steps, sync, world):
    """barrier + device sync on both sides, EXACTLY `steps` steps in between.  A rank's time ends when ITS device has
    drained (the closing barrier follows, outside the interval): the job's time is the MAX over ranks, taken in
    gather_stats -- with the barrier inside, every rank would read the slowest one's time plus the barrier's own
    latency (a CPU barrier over sockets is tenths of a millisecond, a few percent of a 20-step region)."""
    import time
    sync()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    barrier(world)
    return dt


def gather_stats(stats, world, device="cpu"):
    """all_gather of the per-rank record (STATS_FIELDS) and MAX of the elapsed time"""
    forced = os.environ.get("ORBX_BENCH_FORCE_DIST") == "1"
    if world == 1 and not forced:  # no torch in a single-process run
        return [[float(v) for v in stats]], float(stats[3])
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [[float(v) for v in stats]], float(stats[3])
    g = _stats_group()
    on_gpu = g is not None or dist.get_backend() == "nccl"
    t = torch.tensor([float(v) for v in stats], dtype=torch.float64, device=device if on_gpu else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=g)
    tmax = t[3:4].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=g)
    return [o.cpu().tolist() for o in out], float(tmax.item())


def stats_transport():
    """what carried the statistics record (for the bench record)"""
    try:
        import torch.distributed as dist
        if not dist.is_initialized():
            return "none (single process)"
        if _state["stats_group"] is not None:
            return "rccl all_gather (communicator made after the timed region; barriers over gloo)"
        return "%s all_gather" % ("rccl" if dist.get_backend() == "nccl" else dist.get_backend())
    except Exception:
        return "none (single process)"


def aggregate(gathered, dt_max):
    """whole-job frames/s: frames of ALL ranks over the slowest rank's time"""
    total_frames = sum(g[0] for g in gathered)
    return total_frames / dt_max, total_frames


def finalize(world):
    if world > 1 or os.environ.get("ORBX_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        _state.update(stats_backend=None, stats_group=None, device=None)
