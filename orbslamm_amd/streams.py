"""Multi-GPU layout of the hot path: independent camera streams, one per GPU
(MultipleRobotsScenario: one System/Tracking thread per robot,
/root/reference/MultipleRobotsScenario/Examples/Monocular/mono_kitti.cc:83-98).

There is no data-path collective: a stream's frames, pyramid and previous-frame
descriptors never leave its GPU.  torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) is used for exactly two things:
the barrier/MAX around the timed region and one all_gather of a small statistics
record per rank."""
import os

STATS_FIELDS = ("frames", "keypoints_last", "matches_last", "seconds")


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
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


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
    """barrier + device sync on both sides, EXACTLY `steps` steps in between"""
    import time
    sync()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    barrier(world)
    return time.perf_counter() - t0


def gather_stats(stats, world, device="cpu"):
    """all_gather of the per-rank record (STATS_FIELDS) and MAX of the elapsed time"""
    if world == 1:  # no torch in a single-process run
        return [[float(v) for v in stats]], float(stats[3])
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in stats], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    tmax = t[3:4].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return [o.cpu().tolist() for o in out], float(tmax.item())


def aggregate(gathered, dt_max):
    """whole-job frames/s: frames of ALL ranks over the slowest rank's time"""
    total_frames = sum(g[0] for g in gathered)
    return total_frames / dt_max, total_frames


def finalize(world):
    if world > 1 or os.environ.get("ORBX_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
