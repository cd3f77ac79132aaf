#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match-vs-previous-frame on MI355X.

Workload (BASELINE.json configs[2]/[3], `--config c3`, the default): KITTI-shape 1241x376 uint8 mono frames, 2000
features, 8 levels, scale 1.2, FAST 20/7; every frame is extracted and brute-force Hamming matched against the previous
frame of its stream.  `--config c2` is BASELINE.json configs[1] (640x480, 1000 features) with the same pipeline.
A "step" is one pass of the hot path over one batch of `--batch` (default 64) frames that are already resident in HBM.
The frames of consecutive steps are consecutive frames of the camera stream, drawn from a pool of `--pool` batches
(default 16 = 1024 distinct frames, 0.5 GB: twice the 256 MiB Infinity Cache, so a step's input comes from HBM and not
from a cache that a 64-frame loop would sit in).

One process per GPU; each rank owns an independent camera stream (MultipleRobotsScenario: one tracking thread per
robot, mono_kitti.cc:80-101) -- no data-path collective; RCCL only gathers the match statistics after the timed region.

`python bench.py --gpus N` launches itself: without RANK/WORLD_SIZE in the environment and N > 1 it spawns N ranks
(one per GPU, 127.0.0.1 rendezvous), relays rank 0's record and exits non-zero if any rank failed.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are the launcher's.

Prints ONE JSON line on rank 0 (see the contract in the task statement).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

CONFIGS = {
    # name: width, height, features, device row stride (64-byte aligned rows)
    "c3": dict(w=1241, h=376, nfeat=2000, stride=1280,
               metric="frames/s ORB extract+match, 1241x376 @2000 kp; bit-exact kp/desc vs CPU",
               workload="KITTI-shape 1241x376 mono, 2000 features, 8 levels x1.2, FAST 20/7, extract + brute-force Hamming "
                        "match vs previous frame (BASELINE.json configs[2]/[3])"),
    "c2": dict(w=640, h=480, nfeat=1000, stride=640,
               metric="frames/s ORB extract+match, 640x480 @1000 kp; bit-exact kp/desc vs CPU",
               workload="TUM-shape 640x480 mono, 1000 features, 8 levels x1.2, FAST 20/7, extract + brute-force Hamming "
                        "match vs previous frame (BASELINE.json configs[1])"),
}
HBM_PEAK_GBS = 8000.0
SIMDS = 1024  # 256 CUs x 4
DOMINANT = "k_fast"  # bracketed inside the timed region; the serialized replay of every run re-derives the dominant kernel


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 0.1 s timed: a 12 ms region (20 steps) is mostly pipeline ramp-up
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--pool", type=int, default=16, help="distinct batches of stream frames resident in HBM, cycled by the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the (untimed) host-buffer entry measurements")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle comparison of two frames of the last timed step")
    ap.add_argument("--no-natural", action="store_true", help="skip the (untimed) 64-frame steps over natural-image sequences (tests/golden/natural.npz)")
    ap.add_argument("--no-tracking-path", action="store_true", help="skip the (untimed) Tracking-shaped matcher measurements")
    ap.add_argument("--no-live-streams", action="store_true", help="skip the (untimed) one-frame-per-robot-per-call measurements (examples/multi_robot)")
    ap.add_argument("--live-full", action="store_true", help="every live-stream configuration (default: the five the record's targets and DESIGN.md quote)")
    ap.add_argument("--no-dropin-classes", action="store_true", help="skip the (untimed) Tracking-shaped loop over the drop-in C++ classes (examples/tracking_loop)")
    ap.add_argument("--side-budget", type=float, default=150.0, help="seconds of wall-clock ALL untimed side blocks together may take (live_streams, dropin_classes, "
                    "tracking_path, host_path); a block that would start beyond it is skipped and the record says so -- the headline never waits for them longer than this")
    ap.add_argument("--no-profile", action="store_true", help="do not record HIP events inside the timed region")
    ap.add_argument("--no-replay", action="store_true", help="skip the untimed serialized replay (roofline.isolated); used under rocprofv3 so that its per-kernel averages are those of the timed launches")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, timeout=None):
    """Spawn n ranks of this script (one process per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), relay rank 0's
    stdout -- the one JSON record -- on ours, everything else on stderr.  Returns the exit code: 0 only if every rank
    exited 0 and rank 0 printed exactly one line."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), ORBX_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    t0 = time.time()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if timeout is not None and time.time() - t0 > timeout:
            failed = -1
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:  # our own children, by PID
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write("bench.py: rank %s failed (exit codes %s)\n" % ("timeout" if failed < 0 else failed, [p.poll() for p in procs]))
        return 1
    out = procs[0].stdout.read().decode()
    lines = [l for l in out.splitlines() if l.strip()]
    if len(lines) != 1:
        sys.stderr.write("bench.py: rank 0 printed %d lines instead of one record\n%s\n" % (len(lines), out[:2000]))
        return 1
    rec = json.loads(lines[0])
    if rec.get("n_gpus") != n:
        sys.stderr.write("bench.py: record says n_gpus=%r, launched %d ranks\n" % (rec.get("n_gpus"), n))
        return 1
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------------------------------------ reporting helpers
def algorithmic_bytes(level_px, nfeat):
    """staged dataflow, each stage reads its input once and writes its output once.
    level_px: pixels of each pyramid level (from the product's own geometry).  Per-frame bytes per kernel + total."""
    S = sum(level_px)
    per = {
        "k_pyramid": (S - level_px[-1]) + (S - level_px[0]),
        "k_fast": S,
        "k_blur": 2 * S,
        "k_orient_desc": 749 * nfeat + 512 * nfeat + (32 + 28) * nfeat,
        "k_distribute": 0,  # works on candidate records, not counted in the SURVEY figure
        "k_match_mfma": 2 * 32 * nfeat + 8 * nfeat,
        "k_match_accept": 0,
        "k_match_prune": 0,
    }
    return per, sum(per.values())


def _latest_profile(pattern):
    """newest committed profiles/rNN_<pattern> (the PMC passes are separate rocprofv3 runs; bench.py replays their summary)"""
    hits = sorted(glob.glob(os.path.join(_ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    return hits[-1] if hits else None


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def cpu_baseline(cfg, frames, seconds_budget=12.0, max_frames=320, streams_n=8):
    """The CPU checker (a port of the reference algorithm: kind="port") timed on this host, on a bounded sample of the
    same workload: (i) ONE thread -- the reference extractor and matchers are single-threaded per stream
    (Frame.cc:191); (ii) one thread per stream for min(streams_n, cores) streams -- the MultipleRobotsScenario layout."""
    import threading
    from oracle import binding as ob
    try:
        so = ob.build(march_native=True, out_dir="/tmp")
        L = ob.lib(path=so)
    except Exception:
        L = ob.lib()
    W, H, NFEAT = cfg["w"], cfg["h"], cfg["nfeat"]

    def run_stream(budget, out, idx):
        ex = ob.Extractor(NFEAT, 1.2, 8, 20, 7, L=L)
        prev = None
        t0 = time.perf_counter()
        n = 0
        for f in range(max_frames):
            r = ex(frames[(f + 7 * idx) % len(frames)])
            if prev is not None:
                ob.match_bruteforce(r["desc"], r["kps"]["angle"], prev["desc"], prev["kps"]["angle"], 0.7, 50, True, L=L)
            prev = r
            n += 1
            if time.perf_counter() - t0 > budget:
                break
        out[idx] = (n, time.perf_counter() - t0)

    res = {}
    run_stream(seconds_budget, res, 0)
    n1, dt1 = res[0]
    info = host_info()
    if streams_n <= 0:
        return {"value": n1 / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": "%d synthetic %dx%d frames (%.1f s), extract + brute-force match vs previous frame, 1 thread, gcc -O3 -march=native -ffp-contract=off" % (n1, W, H, dt1),
                "nproc": info["nproc"], "cpu_model": info["cpu_model"]}
    nthr = max(1, min(streams_n, info["nproc"] or 1))
    resn = {}
    t0 = time.perf_counter()
    th = [threading.Thread(target=run_stream, args=(seconds_budget * 0.4, resn, i)) for i in range(nthr)]  # ctypes releases the GIL
    for t in th:
        t.start()
    for t in th:
        t.join()
    dtn = time.perf_counter() - t0
    nn = sum(v[0] for v in resn.values())
    return {"value": n1 / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d synthetic %dx%d frames (%.1f s), extract + brute-force match vs previous frame, 1 thread, gcc -O3 -march=native -ffp-contract=off" % (n1, W, H, dt1),
            "one_thread_per_stream": {"value": nn / dtn, "unit": "frames/s", "cores": nthr, "streams": nthr,
                                      "sample": "%d frames over %d threads (%.1f s wall)" % (nn, nthr, dtn)},
            "nproc": info["nproc"], "cpu_model": info["cpu_model"]}


def make_extractor(cfg, B, dev_index):
    """the product's extractor -- or, for the CPU plumbing tests of the N > 1 path, the class named by
    ORBX_BENCH_EXTRACTOR=module:Class (tests/bench_stub.py), which has the same methods and needs no GPU"""
    spec = os.environ.get("ORBX_BENCH_EXTRACTOR")
    if spec:
        import importlib
        mod, cls = spec.split(":")
        return getattr(importlib.import_module(mod), cls)(cfg["nfeat"], 1.2, 8, 20, 7, max_width=cfg["w"], max_height=cfg["h"],
                                                          max_batch=B, device=dev_index)
    from orbslamm_amd import ORBextractor
    return ORBextractor(cfg["nfeat"], 1.2, 8, 20, 7, max_width=cfg["w"], max_height=cfg["h"], max_batch=B, device=dev_index)


def device_count():
    if os.environ.get("ORBX_BENCH_EXTRACTOR"):
        return int(os.environ.get("ORBX_BENCH_STUB_DEVICES", "1"))
    from orbslamm_amd import _lib
    return _lib.lib().orbx_device_count()


def host_path(ex, cfg, frames, seconds=1.5):
    """The host-buffer entries (never `value`): what a drop-in caller sees.
    b1: Frame::ExtractORB's entry -- one pageable host frame per call (orbx_extract), keypoints + descriptors back on the
        host, then match vs previous frame and the match table back on the host; per-frame wall time, reported as the
        reference examples report theirs (median and mean, mono_tum.cc:113-122).
    b64: orbx_extract_batch with 64 host frames per call, results on the host (PCIe-inclusive throughput)."""
    import numpy as np
    W, H = cfg["w"], cfg["h"]
    out = {}
    lat = []
    ex.reset_stream()
    t_end = time.perf_counter() + seconds
    i = 0
    while time.perf_counter() < t_end or len(lat) < 20:
        f = frames[i % len(frames)]
        t0 = time.perf_counter()
        ex.extract_match_host(f[None])
        lat.append(time.perf_counter() - t0)
        i += 1
    lat = np.array(lat[5:]) * 1e3
    out.update(b1_ms_median=float(np.median(lat)), b1_ms_mean=float(lat.mean()), b1_frames=int(len(lat)),
               b1_fps=float(1e3 / lat.mean()), b1_what="orbx_extract_match_batch, B=1: pageable host frame in, keypoints + descriptors + match table on the host, per call")
    B = min(ex.max_batch, len(frames))
    batch = frames[:B]
    ex.reset_stream()
    ex.extract_match_host(batch)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ex.extract_match_host(batch)
        n += B
    dt = time.perf_counter() - t0
    out.update(b64_fps=n / dt, b64_batch=B, b64_what="orbx_extract_match_batch, B=%d host frames per call (synchronous), results on the host" % B)
    # pipelined form: submit batch n+1 while batch n computes and batch n-1 downloads
    if hasattr(ex, "submit_host"):
        ex.reset_stream()
        tickets = []
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            tickets.append(ex.submit_host(batch))
            if len(tickets) > 2:
                ex.collect_host(tickets.pop(0))
            n += B
        while tickets:
            ex.collect_host(tickets.pop(0))
        dt = time.perf_counter() - t0
        out.update(pipelined_fps=n / dt, pipelined_what="orbx_submit_batch / orbx_collect_view + orbx_release, depth 3, B=%d pageable host frames per ticket, results in the pinned host buffer" % B,
                   pcie_gbs=n / dt * (W * H + 0.0) / 1e9)
        # results straight into caller-owned pinned arrays (orbx_submit_batch_into / orbx_collect): what a drop-in with
        # containers of its own does -- no second pass on the host
        if hasattr(ex, "submit_host_into"):
            outs = [ex.alloc_pinned_results(B) for _ in range(3)]
            ex.reset_stream()
            n = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                tickets.append(ex.submit_host_into(batch, outs[(n // B) % 3]))
                if len(tickets) > 2:
                    ex.collect_into(tickets.pop(0))
                n += B
            while tickets:
                ex.collect_into(tickets.pop(0))
            dt = time.perf_counter() - t0
            out.update(pipelined_into_fps=n / dt, pipelined_into_what="orbx_submit_batch_into / orbx_collect, depth 3, B=%d pageable host frames per ticket, "
                       "keypoints + descriptors + match tables written by the device straight into the caller's own pinned arrays" % B)
        # the same from pinned frames in the device layout (orbx_host_alloc_frames), results read in place (orbx_collect_view)
        if hasattr(ex, "alloc_pinned_frames"):
            pins = [ex.alloc_pinned_frames(B, W, H) for _ in range(3)]
            for pf in pins:
                pf.fill(batch)
            ex.reset_stream()
            n = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                tickets.append(ex.submit_host(pins[(n // B) % 3]))
                if len(tickets) > 2:
                    ex.collect_host(tickets.pop(0), view=True)
                n += B
            while tickets:
                ex.collect_host(tickets.pop(0), view=True)
            dt = time.perf_counter() - t0
            out.update(pipelined_pinned_fps=n / dt, pipelined_pinned_what="the same from pinned frames in the device layout (orbx_host_alloc_frames), results read in place (orbx_collect_view)")
            for pf in pins:
                pf.free()
        # the link's own ceiling on this box for exactly these sizes: plain pinned hipMemcpyAsync of one batch's frames up and
        # one batch's results down, alone and both at once (what the pipeline asks of it), and the fraction reached
        if hasattr(ex, "link_rate"):
            try:
                stride = (W + 63) // 64 * 64
                up_b = B * stride * H
                down_b = B * (ex.max_keypoints - 64) * 64   # ~n x (28 + 32 + 4) bytes per frame
                h2d, d2h, bu, bd = ex.link_rate(up_b, down_b, 40)
                peak_fps = bu * 1e9 / (stride * H)
                out.update(pcie_peak_gbs={"h2d_alone": h2d, "d2h_alone": d2h, "h2d_beside_d2h": bu, "d2h_beside_h2d": bd,
                                          "what": "pinned hipMemcpyAsync of %d B up / %d B down per repetition (one batch), GB/s; the two 'beside' figures are each direction's bytes over the time of the loop that issues one copy down per copy up -- the rates the pipeline asks for together, not each direction's limit (tools/ubench/link_duplex.hip has those: 54.6 up beside 42.9 down)" % (up_b, down_b)},
                           pcie_frames_per_s_ceiling=peak_fps,
                           pcie_frac_pinned=out.get("pipelined_pinned_fps", 0.0) / peak_fps,
                           pcie_frac_pageable=out.get("pipelined_fps", 0.0) / peak_fps)
            except Exception as e:
                out["pcie_peak_gbs"] = {"error": repr(e)}
    return out


def tracking_path(ex, cfg, frames, dargs, seconds=1.0):
    """What Tracking runs per frame behind the extractor (never `value`; SURVEY.md 8d: "the grid-windowed variant ...
    because that is what Tracking actually runs per frame"): Frame::Frame's tail (UndistortKeyPoints +
    AssignFeaturesToGrid, Frame.cc:196-210) and SearchByProjection(CurrentFrame, LastFrame, th = 15, bMono)
    (ORBmatcher.cc:1330-1472, Tracking.cc:925-936) with the identity pose, octave window +-1, every LastFrame feature a
    query.  b1: one pageable host frame per call, keypoints + descriptors + the match table back on the host.
    batched: B frames per step, frames resident in HBM, match tables written to pinned host memory.
    dropin: the flattened host-array entry the ORBmatcher adapter calls (queries AND train side from host memory).
    The CPU oracle is timed on the same pair beside it."""
    import numpy as np
    from orbslamm_amd import ORBmatcher, make_grid
    from oracle import binding as ob
    W, H = cfg["w"], cfg["h"]
    B = ex.max_batch
    out = {"what": "SearchByProjection(Cur, Last): identity pose, th=15, octave +-1, TH_HIGH=100, rotation check; mvKeysUn + 64x48 grid built on the device"}
    m = ORBmatcher(0.9, True, device=ex.device)
    sf = np.array(ex.GetScaleFactors(), np.float32)
    g = make_grid(0.0, 0.0, float(W), float(H))
    K, D0, bounds = [718.856, 718.856, 607.1928, 185.2157], [0, 0, 0, 0, 0], [0.0, float(W), 0.0, float(H)]
    fs = m.frame_set(2 * B, ex.max_keypoints, K, D0, g, bounds, sf)
    # ---- b1: host frame -> extract -> frame -> search -> match table on the host, one frame per call
    ex.reset_stream()
    lat, lat_search = [], []
    nm = []
    t_end = time.perf_counter() + seconds
    i = 0
    while time.perf_counter() < t_end or len(lat) < 30:
        f = frames[i % len(frames)]
        slot, prev = i & 1, (i & 1) ^ 1
        t0 = time.perf_counter()
        kps, desc, n, _, _ = ex.extract_match_host(f[None], match=False)
        t1 = time.perf_counter()
        fs.build_from_extractor(slot, ex)
        if i:
            fs.track([slot], [prev], th=15.0)
            a, k = fs.results()
            nm.append(int(k[0]))
        else:
            fs.sync()
        t2 = time.perf_counter()
        lat.append(t2 - t0); lat_search.append(t2 - t1)
        i += 1
    lat, lat_search = np.array(lat[5:]) * 1e3, np.array(lat_search[5:]) * 1e3
    out["b1"] = {"ms_median": float(np.median(lat)), "ms_mean": float(lat.mean()), "fps": float(1e3 / lat.mean()), "frames": int(len(lat)),
                 "search_ms_median": float(np.median(lat_search)), "search_ms_mean": float(lat_search.mean()),
                 "matches_mean": float(np.mean(nm)), "rounds_last": fs.stats(0)[0], "candidates_last": fs.stats(0)[1],
                 "what": "round 3's call sequence, kept for comparison: orbx_extract_match_batch(B=1, no brute-force match), THEN orbm_frameset_build_from_extractor + "
                         "orbm_track_frames + orbm_track_results on the matcher's stream (a host hop and a cross-stream hand-over in between); search_ms = the last three"}
    # the same through the round-4 form: the frame set attached to the extractor, frame + build + search submitted together (one
    # chain on the device), still through this Python mirror and from a pageable frame (live_streams has the native caller)
    if hasattr(fs, "attach") and hasattr(ex, "submit_host"):
        fs.attach(ex)
        lat2 = []
        for i in range(400):
            f = frames[i % len(frames)]
            slot, prev = i & 1, (i & 1) ^ 1
            t0 = time.perf_counter()
            tk = ex.submit_host(f[None], match=False)
            fs.build_from_extractor(slot, ex)
            if i:
                fs.track([slot], [prev], th=15.0)
            ex.collect_host(tk, view=True)
            if i:
                fs.results()
            lat2.append(time.perf_counter() - t0)
        fs.attach(None)
        lat2 = np.array(lat2[20:]) * 1e3
        out["b1_attached"] = {"ms_median": float(np.median(lat2)), "ms_mean": float(lat2.mean()), "fps": float(1e3 / lat2.mean()),
                              "what": "orbm_frameset_attach, then per frame orbx_submit_batch(B=1) + orbm_frameset_build_from_extractor + orbm_track_frames + "
                                      "orbx_collect_view + orbm_track_results (Python mirror, pageable frame)"}
    # ---- batched: B resident frames per step; pair p = (frame p, frame p-1), the first against the previous step's last
    pool = len(dargs)

    def step(n, extract=True):
        base = (n & 1) * B
        if extract:
            ex.extract_batch_device(*dargs[n % pool])
        fs.build_from_extractor(base, ex)
        cur = np.arange(base, base + B)
        last = np.concatenate([[((n & 1) ^ 1) * B + B - 1], cur[:-1]])
        fs.track(cur, last, th=15.0)

    for n in range(4):
        step(n)
    fs.results()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step(n)
        if n > 1:
            fs.results(back=2)  # the consumer reads step n-2's tables while steps n-1 and n are in flight
        n += 1
    a, k = fs.results()
    ex.sync()
    dt = time.perf_counter() - t0
    out["batched"] = {"pairs_per_s": n * B / dt, "ms_per_step": dt / n * 1e3, "batch": B, "steps": n, "matches_mean": float(np.mean(k)),
                      "what": "per step: orbx_extract_batch_device(B) + frame-set build + orbm_track_frames(B pairs), tables read two steps behind"}
    # the tracking kernels alone on the last extracted batch (k_frame_build + k_track_fused, both B workgroups)
    for _ in range(3):
        step(0, extract=False)
    fs.results()
    reps, t0 = 200, time.perf_counter()
    for r in range(reps):
        step(r, extract=False)
    fs.results()
    dt = time.perf_counter() - t0
    out["batched"]["search_only_pairs_per_s"] = reps * B / dt
    out["batched"]["search_only_ms_per_step"] = dt / reps * 1e3
    # ---- dropin: everything from host arrays, per call (orbm_search_by_projection; what ORBmatcherT<...> calls)
    kc, dc = fs.download(B - 1)
    kl, dl = fs.download(B - 2)
    uvr = np.stack([kl["x"], kl["y"], (np.float32(15.0) * sf[kl["octave"]]).astype(np.float32)], axis=1).astype(np.float32)
    lvl = np.stack([kl["octave"] - 1, kl["octave"] + 1], axis=1).astype(np.int8)
    occ0, a0 = np.zeros(len(kc), np.uint8), np.full(len(kc), -1, np.int32)
    ts = []
    for _ in range(60):
        t0 = time.perf_counter()
        ga, _, gn = m.SearchByProjection(4, 100, uvr, lvl, dl, kl["angle"], None, None, g, kc, dc, occ0, a0)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e3
    out["dropin"] = {"ms_median": float(np.median(ts)), "ms_mean": float(ts.mean()), "matches": int(gn), "rounds": m.last_search_stats()[0],
                     "what": "orbm_search_by_projection: queries and train frame from pageable host memory, tables back on the host, per call (includes the ctypes wrapper's array checks)"}
    # ---- the CPU oracle on the same pair (grid build + sequential search), and the check that the GPU said the same
    gp = ob.make_grid_params(0.0, 0.0, float(W), float(H))
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        start, idx = ob.grid_build(gp, kc)
        wa, _, wn = ob.search_by_projection(4, 0.9, True, 100, uvr, lvl, dl, kl["angle"], None, None, gp, kc, start, idx, dc, occ0, a0)
    dt = (time.perf_counter() - t0) / reps
    out["cpu_oracle"] = {"ms_per_pair": dt * 1e3, "pairs_per_s": 1.0 / dt, "cores": 1, "kind": "port", "matches": int(wn)}
    out["parity_ok"] = bool(wn == gn and np.array_equal(wa, ga) and int(k[B - 1]) == wn and np.array_equal(a[B - 1, :len(kc)], wa))
    # ---- Tracking::SearchLocalPoints' search (Tracking.cc:1242-1249 -> ORBmatcher.cc:45-129, mode 3): the local map's
    # MapPoints in view -- here 3 000 made from the two previous frames' keypoints, jittered, radius by viewing angle -- against
    # the frame resident in the set; one pinned upload, two launches, flag-polled table.  The CPU oracle beside it.
    try:
        rng = np.random.default_rng(11)
        src_k = np.concatenate([fs.download(B - 2)[0], fs.download(B - 3)[0]])
        src_d = np.concatenate([fs.download(B - 2)[1], fs.download(B - 3)[1]])
        lm = {"what": "orbm_track_local_points (SearchByProjection(Frame, local MapPoints, th), nnratio 0.8, TH_HIGH 100) against a frame resident in the set: "
                      "queries from host memory in one pinned upload, table read behind the resolve kernel's flag; per call"}
        for th in (1.0, 3.0):
            sel = rng.permutation(len(src_k))[:3000]
            ks, qd = src_k[sel], np.ascontiguousarray(src_d[sel])
            lv = ks["octave"].astype(np.int32)
            r = (np.where(rng.random(len(ks)) < 0.5, np.float32(2.5), np.float32(4.0)) * np.float32(th)).astype(np.float32) * sf[lv]
            uvr3 = np.stack([ks["x"] + rng.normal(0, 1.2, len(ks)), ks["y"] + rng.normal(0, 1.2, len(ks)), r], axis=1).astype(np.float32)
            ql3 = np.stack([lv - 1, lv], axis=1).astype(np.int8)
            ts3 = []
            for _ in range(80):
                t0 = time.perf_counter()
                fs.track_local_points(B - 1, uvr3, ql3, qd)
                ga3, gn3 = fs.results()
                ts3.append(time.perf_counter() - t0)
            ts3 = np.array(ts3[8:]) * 1e3
            start3, idx3 = ob.grid_build(gp, kc)
            t0 = time.perf_counter()
            for _ in range(5):
                wa3, _, wn3 = ob.search_by_projection(3, 0.8, True, 100, uvr3, ql3, qd, None, None, None, gp, kc, start3, idx3, dc, occ0, a0)
            cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
            # batched: 16 searches in flight (a server tracking many robots' local maps), tables read three calls behind
            nrep, t0 = 200, time.perf_counter()
            for i in range(nrep):
                fs.track_local_points(B - 1, uvr3, ql3, qd)
                if i >= 3:
                    fs.results(back=3)
            fs.results()
            pipelined_ms = (time.perf_counter() - t0) / nrep * 1e3
            lm["th%d" % int(th)] = {"queries": int(len(ks)), "ms_median": float(np.median(ts3)), "ms_mean": float(ts3.mean()), "pipelined_ms_per_call": pipelined_ms,
                                    "matches": int(gn3[0]), "rounds": fs.stats(0)[0], "candidates": fs.stats(0)[1],
                                    "cpu_oracle_ms": cpu_ms, "parity_ok": bool(int(gn3[0]) == wn3 and np.array_equal(ga3[0, :len(kc)], wa3))}
        out["local_map"] = lm
        # TrackWithMotionModel's search with the caller's projections (orbm_track_frame_projected): LastFrame's descriptors
        # and angles stay in HBM, 14 bytes per feature go up -- here the identity projection, so that the table must equal
        # orbm_track_frames'
        uvrp = np.stack([kl["x"], kl["y"], (np.float32(15.0) * sf[kl["octave"]]).astype(np.float32)], axis=1).astype(np.float32)
        lvlp = np.stack([kl["octave"] - 1, kl["octave"] + 1], axis=1).astype(np.int8)
        tp = []
        for _ in range(80):
            t0 = time.perf_counter()
            fs.track_projected(B - 1, B - 2, uvrp, lvlp)
            gap, gnp = fs.results()
            tp.append(time.perf_counter() - t0)
        tp = np.array(tp[8:]) * 1e3
        out["projected_pose"] = {"ms_median": float(np.median(tp)), "ms_mean": float(tp.mean()), "matches": int(gnp[0]),
                                 "parity_ok": bool(int(gnp[0]) == gn and np.array_equal(gap[0, :len(kc)], ga)),
                                 "what": "orbm_track_frame_projected: SearchByProjection(Cur, Last, th = 15) with the caller's projections of LastFrame's features "
                                         "(u, v, radius, octave window from host memory), both frames resident; per call, table behind the flag"}
    except Exception as e:  # never lose the record over the secondary block
        out["local_map"] = {"error": repr(e)}
    # ---- TrackReferenceKeyFrame's pair (Tracking.cc:805-812): Frame::ComputeBoW on the new frame, then
    # SearchByBoW(KeyFrame = the previous frame, Frame), on a synthetic vocabulary of ORBvoc's shape (k = 10, L = 6)
    try:
        from orbslamm_amd import ORBVocabulary, synth
        voc = synth.make_vocabulary(10, 6)
        G = ORBVocabulary(10, 6, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=ex.device)
        # the frame set still holds the last batch in slots base .. base + B - 1 (search-only loop above: n = 199 -> odd half)
        base = (199 & 1) * B
        fs.compute_bow(G, base, B, 4)
        fs.sync()
        # b1: per frame, ComputeBoW of the new frame + SearchByBoW against the previous one, table on the host
        tt = []
        for i in range(60):
            t0 = time.perf_counter()
            fs.compute_bow(G, base + B - 1, 1, 4)
            fs.search_by_bow([base + B - 2], [base + B - 1], 0.7, True)
            gm, gnb = fs.bow_results()
            tt.append(time.perf_counter() - t0)
        gm, gnb = gm[0].copy(), int(gnb[0])
        tt = np.array(tt[5:]) * 1e3
        # batched: B frames per step
        kfs = np.arange(base, base + B - 1)
        curs = kfs + 1
        for _ in range(3):
            fs.compute_bow(G, base, B, 4); fs.search_by_bow(kfs, curs, 0.7, True)
        fs.bow_results()
        reps, t0 = 100, time.perf_counter()
        for _ in range(reps):
            fs.compute_bow(G, base, B, 4); fs.search_by_bow(kfs, curs, 0.7, True)
        fs.bow_results()
        dtb = (time.perf_counter() - t0) / reps
        O = ob.Vocabulary(10, 6, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
        t0 = time.perf_counter()
        for _ in range(3):
            _, fvc = O.transform(dc, 4)
        t1 = time.perf_counter()
        _, fvl = O.transform(dl, 4)
        t2 = time.perf_counter()
        for _ in range(5):
            wm, wnb = ob.search_by_bow(dl, kl["angle"], None, fvl, dc, kc["angle"], None, fvc, 0.7, True, True)
        t3 = time.perf_counter()
        out["bow"] = {"b1_ms_median": float(np.median(tt)), "b1_ms_mean": float(tt.mean()), "matches": gnb,
                      "batched_frames_per_s": B / dtb, "batched_ms_per_step": dtb * 1e3, "batch": B,
                      "cpu_oracle": {"transform_ms": (t1 - t0) / 3 * 1e3, "search_by_bow_ms": (t3 - t2) / 5 * 1e3, "cores": 1, "kind": "port"},
                      "parity_ok": bool(gnb == wnb and np.array_equal(gm[:len(kc)], wm)),
                      "what": "orbm_frameset_compute_bow (vocabulary descent + BowVector/FeatureVector, all in HBM) + orbm_bow_frames (SearchByBoW(KeyFrame = previous "
                              "frame, Frame), nnratio 0.7, rotation check) + orbm_bow_results (table in pinned host memory); b1 = one frame per call, batched = B frames "
                              "and B - 1 pairs per step; vocabulary: synthetic complete tree k=10, L=6 (1.1 M nodes)"}
        G.close()
        # the same on a RAGGED tree of the published ORBvoc.txt's size (1.06 M nodes, words at different depths; the file
        # itself is absent from the reference checkout) and on a strongly ragged one (2 .. 10 children, 10 % early leaves)
        for tag, rg in (("orbvoc_sized_ragged", 0.994), ("strongly_ragged", True)):
            vr = synth.make_vocabulary(10, 6, ragged=rg)
            Gr = ORBVocabulary(10, 6, 0, 0, vr["parent"], vr["is_leaf"], vr["desc"], vr["weight"], device=ex.device)
            fs.compute_bow(Gr, base, B, 4)
            fs.sync()
            tr = []
            for i in range(40):
                t0 = time.perf_counter()
                fs.compute_bow(Gr, base + B - 1, 1, 4)
                fs.search_by_bow([base + B - 2], [base + B - 1], 0.7, True)
                gmr, gnr = fs.bow_results()
                tr.append(time.perf_counter() - t0)
            gmr, gnr = gmr[0].copy(), int(gnr[0])
            reps, t0 = 50, time.perf_counter()
            for _ in range(reps):
                fs.compute_bow(Gr, base, B, 4); fs.search_by_bow(kfs, curs, 0.7, True)
            fs.bow_results()
            dtr = (time.perf_counter() - t0) / reps
            Or = ob.Vocabulary(10, 6, 0, 0, vr["parent"], vr["is_leaf"], vr["desc"], vr["weight"])
            _, fvc_r = Or.transform(dc, 4)
            _, fvl_r = Or.transform(dl, 4)
            wmr, wnr = ob.search_by_bow(dl, kl["angle"], None, fvl_r, dc, kc["angle"], None, fvc_r, 0.7, True, True)
            out["bow"][tag] = {"nodes": int(len(vr["parent"])), "words": int(vr["is_leaf"].sum()), "b1_ms_median": float(np.median(np.array(tr[5:]) * 1e3)),
                               "batched_frames_per_s": B / dtr, "matches": gnr, "parity_ok": bool(gnr == wnr and np.array_equal(gmr[:len(kc)], wmr))}
            Gr.close()
    except Exception as e:  # never lose the record over the secondary block
        out["bow"] = dict(out.get("bow") or {}, error=repr(e))
    fs.close()
    m.close()
    return out


class SideBudget:
    """One wall-clock allowance for every untimed side block of a run (the side blocks could outlast the
    driver's limit before a single timed step ran).  A block asks for time before it starts; what it is refused it skips,
    and the record carries what ran, what was skipped and how long each took."""

    def __init__(self, seconds):
        self.total = float(seconds)
        self.t0 = time.time()
        self.spent = {}
        self.skipped = []

    def left(self):
        return self.total - (time.time() - self.t0)

    def allows(self, name, need=5.0):
        if self.left() >= need:
            return True
        self.skipped.append(name)
        return False

    def charge(self, name, t_start):
        self.spent[name] = round(self.spent.get(name, 0.0) + time.time() - t_start, 1)

    def record(self):
        return {"budget_s": self.total, "spent_s": self.spent, "skipped": self.skipped,
                "what": "wall-clock of the untimed side blocks; none of it is inside the timed region"}


def dropin_classes(cfg, budget):
    """The drop-in CLASSES on the clock (never `value`): examples/tracking_loop -- per frame ORBextractor::operator() and
    ORBmatcher STACK TEMPORARIES exactly where Tracking builds them (Tracking.cc:809, 914, 1242), on mock Frame / KeyFrame /
    MapPoint objects at this configuration's size, per-frame median / mean as mono_tum.cc:113-122 prints them.  Per member:
    total = device (inside the C ABI) + adapter (object-graph walk), the same arrays through the raw C ABI, and the round-4
    ownership pattern (a device handle per matcher object) beside it; the program fails if its steady-state loop made a
    matcher handle or a device / pinned allocation."""
    import __graft_entry__ as ge
    try:
        exe = ge.build_tracking_loop()
        r = subprocess.run([exe, "--json", "--frames", "200", "--warmup", "30", "--w", str(cfg["w"]), "--h", str(cfg["h"]), "--features", str(cfg["nfeat"])],
                           capture_output=True, text=True, timeout=max(10.0, min(120.0, budget.left())))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-300:], "rc": r.returncode}
        d = json.loads(line[-1])
        d["what"] = "examples/tracking_loop: ORBextractor::operator() + ORBmatcher temporaries per frame in Tracking's shape, mock SLAM objects; never `value`"
        return d
    except Exception as e:  # never lose the record over a secondary block
        return {"error": repr(e)}


def live_streams(cfg, dev_index=0, frames=400, budget=None, full=False):
    """The mode the reference actually runs (never `value`): ONE live frame per robot per iteration
    (MultipleRobotsScenario/Examples/Monocular/mono_kitti.cc:80-101 -> Tracking::GrabImageMonocular, Tracking.cc:240-267;
    the search of frame t needs only the pose predicted from frame t-1, Tracking.cc:905-936).  Measured by the native
    program a maintainer would write against the C ABI (examples/multi_robot.cpp: one thread per robot, one extractor +
    matcher + frame-set handle per robot, frames from a pinned camera ring unless stated), each configuration in a process
    of its own, per-frame median / mean printed like mono_tum.cc:113-122:
      one_robot.track       extract + Frame tail + SearchByProjection(Cur, Last, th = 15), the frame set attached to the
                            extractor (one chain on the device), keypoints + descriptors + match table on the host
      one_robot.track_d2    the same with the next frame submitted before this one is collected (tickets two deep)
      one_robot.track_d2_two_queues   two deep, frame set NOT attached: the next frame's extraction on the extractor's
                            queue runs beside this frame's search on the matcher's
      one_robot.track_plus_local_map   track, then SearchByProjection(Frame, 3 000 local MapPoints) (Tracking.cc:1242-1249) on the same frame
      one_robot.bf          extract + brute-force match vs previous frame (BASELINE.json's pair), one frame per call
      robots_on_one_gpu     K robots = K threads on the one GPU, one frame per robot per call, and 8 cameras fed two per
                            call by 4 threads (the GPU runs about four queues at a time: tools/live_scale_probe.sh); 8 and 16
                            robot threads behind four orbslamm::CameraHub's (the frames that wait together share a chain)"""
    import __graft_entry__ as ge
    exe = ge.build_examples()
    base = [exe, "--json", "--interval", "0", "--w", str(cfg["w"]), "--h", str(cfg["h"]), "--nfeat", str(cfg["nfeat"]), "--gpus", "1"]
    env = dict(os.environ)   # (single-GPU runs only: the program takes device 0 of what this process sees)
    keep = ("frames_per_s", "ms_median", "ms_mean", "ms_p99", "keypoints_mean", "matches_mean", "host_us_submit", "host_us_enqueue")

    budget = budget or SideBudget(600.0)

    def run(*extra, n=frames, more_env=None, timeout=120, only_full=False):
        if only_full and not full:
            return {"skipped": "--live-full"}
        if budget.left() < 4.0:
            budget.skipped.append("live_streams:" + " ".join(str(a) for a in extra))
            return {"skipped": "side budget spent"}
        try:
            e = dict(env, **more_env) if more_env else env
            r = subprocess.run(base + ["--frames", str(n), "--warmup", "40"] + [str(a) for a in extra], capture_output=True, text=True,
                               timeout=max(4.0, min(timeout, budget.left())), env=e)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                return {"error": (r.stderr or r.stdout)[-300:]}
            d = json.loads(line[-1])
            return {k: d[k] for k in keep}
        except Exception as e:  # never lose the record over a secondary block
            return {"error": repr(e)}

    # The program links librccl (its counters' all-gather), and this is the first process of the bench to map it: on a fresh
    # box the image pages it in on first use -- minutes, now and then (one refresh run lost its first configuration to the
    # 180 s limit that way).  One short untimed run takes that, with a limit to match.
    t_first = time.time()
    first = run("--mode", "extract", n=5, timeout=budget.left())
    t_first = time.time() - t_first
    out = {"what": "examples/multi_robot (C ABI, one thread per robot, B = 1 per call, pinned camera ring unless stated); never `value`",
           "first_run_s": round(t_first, 1), "first_run_ok": "error" not in first,
           "one_robot": {
               "track": run("--mode", "track"),
               "track_pageable_frames": run("--mode", "track", "--pinned", 0, only_full=True),
               "track_d2": run("--mode", "track", "--depth", 2, only_full=True),
               # (ORBX_LAT_PRIO=0: the extractor's queue at the matcher's priority -- a high- and a normal-priority queue busy
               # at the same time are time-sliced in ~50 us quanta, tools/live_d2_trace.sh: 2.7 k frames/s instead of 8.2 k)
               "track_d2_two_queues": run("--mode", "track", "--depth", 2, "--attach", 0, more_env={"ORBX_LAT_PRIO": "0"}, only_full=True),
               # the whole front-end of one Tracking iteration: the chain above + Tracking::SearchLocalPoints' search of 3 000 local
               # MapPoints against the same resident frame, collected one after the other as Tracking needs them
               "track_plus_local_map": run("--mode", "full"),
               "bf": run("--mode", "bf"),
               "bf_pageable_frames": run("--mode", "bf", "--pinned", 0, only_full=True),
               "extract": run("--mode", "extract", only_full=True)},
           "robots_on_one_gpu": {}}
    for k in (2, 4, 8):
        out["robots_on_one_gpu"]["track_%d_threads_x1" % k] = run("--mode", "track", "--robots", k, n=300, only_full=(k != 4))
    out["robots_on_one_gpu"]["track_4_threads_x2_cameras"] = run("--mode", "track", "--robots", 4, "--per-call", 2, n=300)
    out["robots_on_one_gpu"]["bf_4_threads_x1"] = run("--mode", "bf", "--robots", 4, n=300, only_full=True)
    # more robots than the GPU runs queues: one thread and one blocking call per robot as before, but the frames of the robots
    # waiting together go through ONE chain (include/orbslamm_hub.hpp, orbx_create_live) -- four hubs per GPU
    keep = keep + ("hub_batch_mean",)
    out["robots_on_one_gpu"]["track_8_threads_x1_hubs_of_2"] = run("--mode", "track", "--robots", 8, "--hub", 2, n=300)
    out["robots_on_one_gpu"]["track_16_threads_x1_hubs_of_4"] = run("--mode", "track", "--robots", 16, "--hub", 4, n=300, only_full=True)
    # the same program's offline-sequence mode (--mode batch: this file's own step through the device-resident entries, one
    # thread + one handle per GPU): the native cross-check of `value`, measured in a process of its own
    r = run("--mode", "batch", "--steps", 200, "--warmup", 5, only_full=True)
    out["native_offline_batch"] = {k: r[k] for k in ("frames_per_s", "keypoints_mean", "matches_mean") if k in r} if ("error" not in r and "skipped" not in r) else dict(r)
    out["native_offline_batch"]["what"] = "examples/multi_robot --mode batch: 64 frames per step, 8 batches resident in HBM, 200 steps -- bench.py's `value` from the C++ caller"
    t = out["one_robot"]["track"]
    if "ms_median" in t:
        out["target"] = {"track_ms_median_le_0.20": bool(t["ms_median"] <= 0.20), "bf_ms_median_le_0.15": bool(out["one_robot"]["bf"].get("ms_median", 9) <= 0.15),
                         "eight_cameras_ge_25k": bool(out["robots_on_one_gpu"]["track_4_threads_x2_cameras"].get("frames_per_s", 0) >= 25000),
                         "eight_robot_threads_ge_25k": bool(out["robots_on_one_gpu"]["track_8_threads_x1_hubs_of_2"].get("frames_per_s", 0) >= 25000)}
    return out


def pci_code(bus_id):
    """"0000:c1:00.0" -> a number that survives the float64 all_gather (-1: unknown)"""
    try:
        dom, bus, rest = bus_id.split(":")
        dev, fn = rest.split(".")
        return float((int(dom, 16) << 20) | (int(bus, 16) << 12) | (int(dev, 16) << 4) | int(fn, 16))
    except (AttributeError, ValueError):
        return -1.0


def pci_text(code):
    if code is None or code < 0:
        return None
    c = int(code)
    return "%04x:%02x:%02x.%x" % (c >> 20, (c >> 12) & 0xFF, (c >> 4) & 0xFF, c & 0xF)


def pin_to_gpu_numa(dev_index):
    """Pin this rank to the cores of its GPU's NUMA node (os.sched_setaffinity): the HBM-resident headline does not
    care, the host-fed entries on a two-socket box do (staging copies and pinned buffers on the far socket cross the
    inter-socket link).  Returns what was done, for the record."""
    if os.environ.get("ORBX_BENCH_EXTRACTOR") or os.environ.get("ORBX_BENCH_NO_PIN") == "1":
        return {"pinned": False, "why": "disabled"}
    from orbslamm_amd import _lib
    node, cpus = _lib.numa_cpus_of_device(dev_index)
    if not cpus:
        return {"pinned": False, "why": "the platform names no NUMA node for the device"}
    try:
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return {"pinned": False, "numa_node": node, "why": "none of the node's cores is in this process's cpuset"}
        os.sched_setaffinity(0, allowed)
        return {"pinned": True, "numa_node": node, "cores": len(allowed)}
    except OSError as e:
        return {"pinned": False, "numa_node": node, "why": repr(e)}


def parity_check(ex, cfg, host_frames, frames_idx):
    """`metric` says "bit-exact kp/desc vs CPU": checked here, in this very process, on frames of the LAST timed step --
    keypoints (28-byte records), descriptors and the match table of each checked frame against the CPU oracle run on the
    same input bytes.  host_frames[i] = the step's frame i on the host."""
    import numpy as np
    from oracle import binding as ob
    oex = ob.Extractor(cfg["nfeat"], 1.2, 8, 20, 7)
    ref = {}
    for f in sorted(set(frames_idx) | {i - 1 for i in frames_idx}):
        ref[f] = oex(host_frames[f])
    bad = []
    for f in frames_idx:
        kps, desc = ex.download(f)
        m, nm = ex.download_matches(f)
        r, rp = ref[f], ref[f - 1]
        wm, wn = ob.match_bruteforce(r["desc"], r["kps"]["angle"], rp["desc"], rp["kps"]["angle"], 0.7, 50, True)
        ok = (len(kps) == len(r["kps"]) and kps.tobytes() == r["kps"].tobytes() and desc.tobytes() == r["desc"].tobytes()
              and nm == wn and np.array_equal(m[:len(wm)], wm))
        if not ok:
            bad.append(int(f))
    return {"frames": len(frames_idx), "ok": not bad, "mismatching_frames": bad, "checked": [int(f) for f in frames_idx],
            "what": "keypoints, descriptors and match table of these frames of the last timed step, byte for byte against the CPU oracle on the same input"}


def natural_sequences(W, H, B):
    """The 64-frame step on frames with natural statistics (the reference's inputs are photographs,
    mono_tum.cc:60-78 -- absent here).  From tests/golden/natural.npz (scikit-image photographs, 1241x376 only):
      retina_pan  a camera panning back and forth over the retina photograph at native resolution (origin as synth.py pans
                  its canvas): LOW texture -- nearly every cell comes back empty at iniThFAST and runs again at minThFAST
      mosaic      camera | astronaut | gravel side by side, a static camera with +-1 sensor noise per frame: mixed
      hubble      the Hubble deep field (x1.241), static camera with +-1 noise: point-like texture everywhere
    Returns name -> [B, H, W] uint8 (deterministic: seeded noise)."""
    import numpy as np
    from orbslamm_amd import synth
    path = os.path.join(_ROOT, "tests", "golden", "natural.npz")
    if not os.path.exists(path) or (W, H) not in ((1241, 376), (640, 480)):
        return {}
    z = np.load(path)
    seqs = {}
    if (W, H) == (640, 480):
        # --config c2 (BASELINE configs[1], the TUM shape of configs[0]): three of the fixture's 640x480 photographs, a static camera
        # with +-1 sensor noise per frame -- camera (mixed), brick (dense regular texture), grass (dense fine texture, no cell retries)
        for name, seed in (("camera", 21), ("brick", 22), ("grass", 23)):
            base = z["c2_" + name].astype(np.int16)
            rng = np.random.Generator(np.random.PCG64(0x4E41 + seed))
            seqs[name] = np.stack([np.clip(base + rng.integers(-1, 2, size=base.shape, dtype=np.int16), 0, 255).astype(np.uint8) for _ in range(B)])
        return seqs
    cv = z["c3_canvas"]
    seqs["retina_pan"] = np.stack([np.ascontiguousarray(cv[synth._tri(t, 16):synth._tri(t, 16) + H, synth._tri(2 * t, 64):synth._tri(2 * t, 64) + W]) for t in range(B)])
    for name, seed in (("mosaic", 11), ("hubble", 12)):
        base = z["c3_" + name].astype(np.int16)
        rng = np.random.Generator(np.random.PCG64(0x4E41 + seed))
        seqs[name] = np.stack([np.clip(base + rng.integers(-1, 2, size=base.shape, dtype=np.int16), 0, 255).astype(np.uint8) for _ in range(B)])
    return seqs


def natural_block(ex, cfg, B, stride, steps=40, warmup=8):
    """Untimed side block: the headline's step (extract + brute-force match of B frames resident in HBM) on each natural
    sequence -- frames/s, k_fast's average launch inside those steps, what fraction of the cells the reference's loop runs
    a second time at minThFAST (ORBextractor.cc:808-816; counted by the oracle on one frame), keypoints per frame, and
    two frames of the last step byte for byte against the oracle."""
    import numpy as np
    from oracle import binding as ob
    W, H = cfg["w"], cfg["h"]
    out = {"what": "the timed region's step on %d-frame batches of natural images (tests/golden/natural.npz), untimed side block: %d steps after %d warm-up steps; "
                   "cells_retry = share of FAST cells empty at iniThFAST and run again at minThFAST (oracle count on one frame)" % (B, steps, warmup)}
    oex = ob.Extractor(cfg["nfeat"], 1.2, 8, 20, 7)
    for name, frames in natural_sequences(W, H, B).items():
        d = ex.upload_frames(frames, stride=stride)

        def step():
            ex.extract_batch_device(*d)
            ex.match_prev_batch_device(0.7, 50, True)
        for _ in range(warmup):
            step()
        ex.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ex.sync()
        dt = time.perf_counter() - t0
        ex.profile_select("k_fast")
        ex.profile_enable(True)
        ex.profile_read(reset=True)
        for _ in range(8):
            step()
        pr = ex.profile_read(reset=True)
        ex.profile_enable(False)
        ex.profile_select(None)
        kf = pr.get("k_fast", (0.0, 0))
        # the batch repeats, so the last step's frame f followed frame f - 1 of the same batch (f >= 1)
        par = parity_check(ex, cfg, {f: frames[f] for f in (B - 3, B - 2, B - 1)}, [B - 2, B - 1])
        cells, retry, empty = oex.cell_stats(frames[B // 2])
        kps, _ = ex.download(B - 1)
        _, nm = ex.download_matches(B - 1)
        out[name] = {"fps": B * steps / dt, "ms_per_step": dt / steps * 1e3, "k_fast_avg_launch_ms": kf[0] / kf[1] if kf[1] else None,
                     "k_fast_ms_per_step": kf[0] / 8.0 if kf[1] else None, "cells": cells, "cells_retry": retry / max(cells, 1),
                     "cells_empty_at_both_thresholds": empty / max(cells, 1), "keypoints_last_frame": int(len(kps)), "matches_last_frame": int(nm),
                     "parity_ok": bool(par["ok"])}
    return out


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import numpy as np
    from orbslamm_amd import streams, synth

    cfg = CONFIGS[args.config]
    W, H, NFEAT, STRIDE = cfg["w"], cfg["h"], cfg["nfeat"], cfg["stride"]
    rank, world, local_rank = streams.env_rank()
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 2
    torch = None
    device = "cpu"
    force_dist = os.environ.get("ORBX_BENCH_FORCE_DIST") == "1"  # world 1 through the N > 1 initialisation path
    distributed = world > 1 or force_dist
    backend = os.environ.get("ORBX_DIST_BACKEND", "nccl")  # "gloo": plumbing test of N ranks on a box with fewer GPUs
    # The live-stream block runs FIRST, before this process has as much as initialised the HIP runtime: its robots are to
    # be measured the way they would be deployed -- their process alone on the device.  (With this process's runtime up,
    # even before it owns a stream, 8 cameras read 18.6 k frames/s instead of 30 k: the GPU runs about four queues at a
    # time and every queue a process holds takes part in the rotation, idle or not -- docs/experiments.md, round 4.)
    # It is no part of the timed region either way.
    ls = None
    dic = None
    budget = SideBudget(args.side_budget)
    alone = world == 1 and not distributed and not os.environ.get("ORBX_BENCH_EXTRACTOR")
    if alone and not args.no_live_streams and not args.no_tracking_path and budget.allows("live_streams", 20.0):
        t_blk = time.time()
        ls = live_streams(cfg, 0, budget=budget, full=args.live_full)
        budget.charge("live_streams", t_blk)
    if alone and not args.no_dropin_classes and not args.no_tracking_path and budget.allows("dropin_classes", 10.0):
        t_blk = time.time()
        dic = dropin_classes(cfg, budget)   # (a process of its own, like the robots above: the classes as a maintainer's binary runs them)
        budget.charge("dropin_classes", t_blk)
    ndev = device_count()
    if backend == "nccl" and distributed and local_rank >= max(ndev, 1):
        sys.stderr.write("bench.py: rank %d has no GPU of its own (%d visible); one process per GPU\n" % (rank, ndev))
        return 2
    dev_index = streams.device_of_rank(local_rank, ndev, exclusive=(backend == "nccl" or not distributed or os.environ.get("ORBX_DIST_EXCLUSIVE") == "1"))

    # stdout carries exactly ONE line, the JSON record.  librccl prints a version banner to fd 1 when its first
    # communicator comes up (and C stdio flushes it at exit, i.e. AFTER anything Python printed), so in a
    # multi-process run fd 1 is pointed at stderr for the life of the process and the record goes to the saved fd.
    json_fd = None
    if distributed:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)

    # The extractor handle comes FIRST: orbx_create makes its four streams and binds them to the process's first four
    # hardware queues (orbslamm_hip.hip).  With torch + RCCL initialised before it they take those queues, the
    # handle's streams share what is left and a rank runs at 82 % of the single-process rate (105 k against 129 k
    # frames/s, tools/dist_order.sh) -- which the driver would read as 0.82 scaling efficiency of a path that has no
    # data-path collective.
    B = args.batch
    stream_id = streams.stream_of_rank(rank)[0]  # this rank's camera stream
    pool = max(1, args.pool)
    pin = pin_to_gpu_numa(dev_index) if distributed else {"pinned": False, "why": "single process: left to the caller's cpuset"}
    ex = make_extractor(cfg, B, dev_index)
    canvas = synth.make_scene(W, H, stream_id)
    dargs = []
    first_batch = None
    for p in range(pool):  # frames t = p*B .. p*B+B-1 of the stream, resident in HBM before the timed region
        fr = np.stack([synth.frame_from_scene(canvas, W, H, p * B + t, stream_id) for t in range(B)])
        if p == 0:
            first_batch = fr
        dargs.append(ex.upload_frames(fr, stride=STRIDE))
    del fr

    if distributed:
        # torch only for torch.distributed (backend "nccl" = RCCL over xGMI); a 1-GPU run needs no torch at all
        # (its first import on a cold box can take minutes)
        import torch
        if backend == "nccl":
            torch.cuda.set_device(dev_index)
            device = torch.device("cuda", dev_index)
        streams.init(backend, device if backend == "nccl" else None)
        import torch.distributed as dist
        assert dist.get_world_size() == args.gpus, "the process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus)

    SAMPLE = 4  # in the timed region the dominant kernel's launches are bracketed in one step out of SAMPLE
    state = {"i": 0, "sample": False, "n": 0}

    def step():
        if state["sample"]:
            ex.profile_enable(state["i"] % SAMPLE == 0)
            state["i"] += 1
        ex.extract_batch_device(*dargs[state["n"] % pool])
        ex.match_prev_batch_device(0.7, 50, True)
        state["n"] += 1

    def sync():
        ex.sync()  # orbx_sync: every stream of the handle (all GPU work of this process hangs off it)
        if torch is not None and backend == "nccl":
            torch.cuda.synchronize()

    import gc
    # Everything that is not a step happens BEFORE the warm-up (collector, profiler events, result arrays), so that the
    # W warm-up steps run straight into the barrier-bracketed timed region: an idle GPU drops its clocks within
    # milliseconds and takes tens of milliseconds of load to bring them back -- with 8192 profiler events created between
    # warm-up and timing a 20-step run (10 ms) read 122 k frames/s where the steady state is 133 k.
    gc.collect()
    gc.disable()  # no collector pause inside the timed region (the host only enqueues, ~0.15 ms per step)
    if not args.no_profile:
        # inside the timed region only the dominant kernel's launches are bracketed with HIP events: event records
        # between the dependent launches of all kernels cost 4 % of the throughput, this kernel's alone ~1 %
        # (and those of one step in SAMPLE: ~1 %)
        ex.profile_select(DOMINANT)
        ex.profile_enable(True)
        state["sample"] = True
    # Pool validation (untimed, before the W warm-up steps): every resident batch goes through the pipeline once, the
    # device error flag is read (a scratch overflow on any batch would otherwise surface in the timed region) and each
    # batch must yield keypoints.  It also leaves the stream with a previous frame and both result sets in use.  Stated
    # openly because it matters for short runs: a step is 0.5 ms, so the driver's W = 5 is 2.4 ms of load, not enough for
    # the GPU to reach its steady clocks (20 timed steps read 124 k frames/s after 5 warm-up steps alone, 134 k after 50;
    # tools/short_run.sh) -- with this pass in front the warm-up starts on a GPU that has 8 ms of load behind it.
    for p in range(pool):
        step()
    sync()   # orbx_sync returns the device error flag
    state["n"] = 0
    kp_check, _ = ex.download(B - 1)
    if len(kp_check) == 0:
        sys.stderr.write("bench.py: the pool's last batch yields no keypoints\n")
        return 3
    for _ in range(args.warmup):
        step()
    if not args.no_profile:
        ex.profile_read(reset=True)  # drops the validation's and the warm-up's spans; synchronises, like the sync() below
        state["i"] = 0
    dt = streams.timed_region(step, args.steps, sync, world)
    state["sample"] = False
    # Self-diagnosis of a rank (the first 8-GPU run is the first time `device != 0` executes): the shader
    # clock right behind the region -- a cold (2.0 GHz) or throttled GPU reads differently from a slow pipeline --, which
    # device this rank really drove (ordinal + PCI bus id) and where it was pinned.  All of it rides in the gathered record.
    diag = {"device": dev_index, "pci": None, "clock_mhz": None}
    if not os.environ.get("ORBX_BENCH_EXTRACTOR"):
        from orbslamm_amd import _lib as _l
        diag["clock_mhz"] = _l.shader_clock_mhz(dev_index)
        diag["pci"] = _l.device_pci_bus_id(dev_index)
    prof = ex.profile_read(reset=True) if not args.no_profile else {}
    ex.profile_enable(False)
    ex.profile_select(None)
    steps_bracketed = (args.steps + SAMPLE - 1) // SAMPLE
    gc.enable()

    # match statistics of the last frame; RCCL all_gather over xGMI (not on the data path)
    _, nmatch_last = ex.download_matches(B - 1)
    kps_last, _ = ex.download(B - 1)
    real = not os.environ.get("ORBX_BENCH_EXTRACTOR")
    par = None
    if real and not args.no_parity_check and B >= 3:
        p_last = (args.warmup + args.steps - 1) % pool   # the batch the last timed step extracted
        hf = {f: synth.frame_from_scene(canvas, W, H, p_last * B + f, stream_id) for f in (B - 3, B - 2, B - 1)}
        par = parity_check(ex, cfg, hf, [B - 2, B - 1])
        if not par["ok"]:
            sys.stderr.write("bench.py: rank %d: frames %s of the last step differ from the CPU oracle\n" % (rank, par["mismatching_frames"]))
    nat = None
    if real and world == 1 and not args.no_natural and B >= 3 and budget.allows("natural", 8.0):
        t_blk = time.time()
        nat = natural_block(ex, cfg, B, STRIDE)
        # the synthetic headline frames, for the same two statistics
        from oracle import binding as _ob
        c_, r_, e_ = _ob.Extractor(NFEAT, 1.2, 8, 20, 7).cell_stats(first_batch[B // 2])
        nat["synthetic_headline_frames"] = {"cells": c_, "cells_retry": r_ / max(c_, 1), "cells_empty_at_both_thresholds": e_ / max(c_, 1)}
        budget.charge("natural", t_blk)
    # The Tracking-shaped block runs BEFORE the host-buffer entries: those give the handle its copy streams (two of them on
    # hardware queues of their own), and with more than ~4 queues alive in the process the GPU rotates them -- the
    # batched extraction + search step then reads 113 k pairs/s instead of 150 k (idle queues count)
    trk = None
    if real and world == 1 and not args.no_tracking_path and hasattr(ex, "extract_match_host") and budget.allows("tracking_path", 10.0):
        t_blk = time.time()
        trk = tracking_path(ex, cfg, first_batch, dargs)
        budget.charge("tracking_path", t_blk)
    hp = None
    # (N > 1: every rank measures its own link at the same time and the figures ride in the gathered record, so the block is
    # not optional there)
    if real and not args.no_host_path and hasattr(ex, "extract_match_host") and (distributed or budget.allows("host_path", 10.0)):
        t_blk = time.time()
        hp = host_path(ex, cfg, first_batch)
        budget.charge("host_path", t_blk)
    gathered, dt_max = streams.gather_stats((B * args.steps, len(kps_last), nmatch_last, dt,
                                             -1.0 if par is None else float(par["ok"]),
                                             0.0 if hp is None else hp.get("pipelined_fps", 0.0),
                                             0.0 if hp is None else hp.get("b1_ms_median", 0.0),
                                             float(diag["device"]), pci_code(diag["pci"]),
                                             -1.0 if pin.get("numa_node") is None else float(pin["numa_node"]), float(bool(pin.get("pinned"))),
                                             -1.0 if diag["clock_mhz"] is None else float(diag["clock_mhz"])), world, device)

    # serialized replay (untimed): the same steps with every kernel alone on the GPU, to tell
    # kernel cost from overlap.  `value` above is NOT affected by it.
    prof_serial, prof_all = {}, {}
    if rank == 0 and not args.no_profile and not args.no_replay:
        ex.profile_enable(True)  # the same overlapped steps, every kernel bracketed (untimed)
        ex.profile_read(reset=True)
        for _ in range(args.steps):
            step()
        prof_all = ex.profile_read(reset=True)
        ex.profile_enable(False)
        ex.set_serial(True)
        step()
        ex.sync()
        ex.profile_enable(True)
        ex.profile_read(reset=True)
        for _ in range(args.steps):
            step()
        prof_serial = ex.profile_read(reset=True)
        ex.profile_enable(False)
        ex.set_serial(False)

    if rank == 0:
        fps, total_frames = streams.aggregate(gathered, dt_max)
        level_px = [a * b for a, b in ex.level_sizes()]
        per, bytes_frame = algorithmic_bytes(level_px, NFEAT)
        out = {
            "metric": cfg["metric"],
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            # SURVEY.md 8(d)'s protocol figure (H2D of the frames and D2H of keypoints, descriptors and matches INSIDE the clock;
            # pipelined 64-frame tickets from pinned host frames) beside `value`, which is the HBM-resident rate of BASELINE config 4
            "value_host_inclusive": None if hp is None else hp.get("pipelined_pinned_fps"),
            "value_host_inclusive_pageable": None if hp is None else hp.get("pipelined_fps"),
            "config": {"workload": cfg["workload"], "name": args.config,
                       "frames_per_step_per_gpu": B, "streams": world, "parallelism": "1 independent stream per GPU",
                       "resident_pool_frames_per_gpu": pool * B, "resident_pool_mb_per_gpu": pool * B * STRIDE * H / 1e6,
                       "pool_validation_steps_before_warmup": pool,
                       # SURVEY.md 8(d) asks for wall-clock including H2D of frames and D2H of results (the reference times the whole
                       # TrackMonocular call, mono_tum.cc:80-97); `value` is the HBM-resident rate the task's contract defines, and the
                       # PCIe-inclusive figures of the same workload ride here, as scalars, so that they survive every parser:
                       "io_in_timed_region": False,
                       "host_inclusive_fps_pinned": None if hp is None else hp.get("pipelined_pinned_fps"),
                       "host_inclusive_fps_pageable": None if hp is None else hp.get("pipelined_fps"),
                       "host_inclusive_one_frame_per_call_ms": None if hp is None else hp.get("b1_ms_median"),
                       "dropin_classes_tracking_frame_ms_median": None if not dic or "error" in dic else dic.get("median_motion_model_frame_ms"),
                       # what the timed interval is (orbslamm_amd/streams.py: timed_region): records of rounds <= 3 had the closing
                       # barrier inside the interval and, at N > 1, RCCL up before it -- not comparable silently
                       "timed_region_version": 2, "closing_barrier_inside_interval": False,
                       "rccl_up_before_timed_region": bool(distributed and backend == "nccl" and os.environ.get("ORBX_DIST_EAGER_NCCL") == "1")},
            "stats_gather": streams.stats_transport() if distributed else "none (single process)",
            "keypoints_last_frame": [int(g[1]) for g in gathered],
            "matches_last_frame": [int(g[2]) for g in gathered],
        }
        traffic_src = _latest_profile("pmc_traffic.json")
        sq_src = _latest_profile("pmc_sq.json")
        traffic_tab = json.load(open(traffic_src)) if traffic_src else {}
        sq_tab = json.load(open(sq_src)) if sq_src else {}
        # the replayed PMC tables carry the hash of the kernel sources they were measured on
        sys.path.insert(0, os.path.join(_ROOT, "tools"))
        from kernels_sha import kernels_sha
        sha_now = kernels_sha()
        replay = {"kernels_sha16_now": sha_now,
                  "traffic": {"source": os.path.basename(traffic_src) if traffic_src else None, "kernels_sha16": traffic_tab.pop("_kernels_sha16", None)},
                  "issue": {"source": os.path.basename(sq_src) if sq_src else None, "kernels_sha16": sq_tab.get("kernels_sha16")}}
        for k in ("traffic", "issue"):
            replay[k]["stale"] = replay[k]["kernels_sha16"] != sha_now   # (tables older than round 4 carry no hash: stale by definition)
        if replay["traffic"]["stale"] or replay["issue"]["stale"]:
            sys.stderr.write("bench.py: the replayed PMC tables (%s, %s) were not taken on the current kernel sources: roofline.traffic / roofline.issue are from an older build\n"
                             % (replay["traffic"]["source"], replay["issue"]["source"]))

        def roofline_of(prof, dom=None, nsteps=args.steps):
            kern = {k: v for k, v in prof.items() if v[1] > 0 and k.startswith("k_")}
            if dom is None:
                dom = max(kern, key=lambda k: kern[k][0])
            avg_ms = kern[dom][0] / kern[dom][1]
            frames_per_launch = B * nsteps / kern[dom][1]
            alg_launch = per.get(dom, 0) * frames_per_launch
            achieved = alg_launch / (avg_ms * 1e-3) / 1e9
            r = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms,
                 "frames_per_launch": frames_per_launch, "algorithmic_bytes_per_launch": alg_launch,
                 "launches_bracketed": int(kern[dom][1]), "kernel_ms_per_step": {k: v[0] / nsteps for k, v in kern.items()}}
            t = traffic_tab.get(dom) if args.config == "c3" else None
            if t:  # PMC pass (separate rocprofv3 --pmc runs, not this run), KB per 64-frame step as reported
                r["traffic"] = (t["fetch_kb"] * t.get("fetch_scale", 1.0) + t["write_kb"]) * 1024.0 * frames_per_launch / t["frames_per_launch"]
                r["traffic_source"] = os.path.relpath(traffic_src, _ROOT) + " (replayed: FETCH_SIZE/WRITE_SIZE passes of an earlier run of this command)"
            q = sq_tab.get("kernels", {}).get(dom) if args.config == "c3" else None
            if q:
                # the yardstick for an issue-bound kernel: VALU wave-instructions x measured cycles per instruction over the
                # SIMD-cycles the launch had: (1024 SIMDs x clock x time).  Counts from the SQ_* passes (per 64-frame step).
                cyc = sq_tab["cycles_per_valu_instr"]
                clk = sq_tab["clock_ghz"] * 1e9
                winstr = q["valu"] * frames_per_launch / sq_tab["frames_per_step"]
                r["issue"] = {"valu_wave_instr_per_launch": winstr, "cycles_per_instr": cyc, "clock_ghz": sq_tab["clock_ghz"], "simds": SIMDS,
                              "frac": winstr * cyc / (SIMDS * clk * avg_ms * 1e-3),
                              "salu_wave_instr_per_launch": q.get("salu", 0) * frames_per_launch / sq_tab["frames_per_step"],
                              "lds_wave_instr_per_launch": q.get("lds", 0) * frames_per_launch / sq_tab["frames_per_step"],
                              "lds_bank_conflict_cycles_per_launch": q.get("lds_bank_conflict", 0) * frames_per_launch / sq_tab["frames_per_step"],
                              "source": os.path.relpath(sq_src, _ROOT) + " (replayed: SQ_INSTS_* passes, kernels alone on the GPU)"}
            return r

        if prof_serial:
            # the dominant kernel is picked where kernels run alone; its figure inside the timed
            # (overlapped) region is reported as `roofline`, the isolated one beside it
            iso = roofline_of(prof_serial)
            timed = prof if prof.get(iso["kernel"], (0, 0))[1] > 0 else prof_all  # the replay names another kernel than DOMINANT
            out["roofline"] = roofline_of(timed, iso["kernel"], steps_bracketed if timed is prof else args.steps) if timed else iso
            out["roofline"]["measured_in_timed_region"] = timed is prof
            out["roofline"]["kernel_ms_per_step_all_bracketed"] = {k: v[0] / args.steps for k, v in prof_all.items() if v[1] > 0 and k.startswith("k_")}
            out["roofline"]["overlapped_streams"] = True
            out["roofline"]["isolated"] = {k: iso[k] for k in ("achieved", "frac", "avg_launch_ms", "frames_per_launch", "kernel_ms_per_step", "issue") if k in iso}
            out["roofline"]["pipeline_bytes_per_frame"] = bytes_frame
            out["roofline"]["pipeline_achieved_GBs"] = fps / world * bytes_frame / 1e9
            out["roofline"]["pipeline_frac"] = fps / world * bytes_frame / 1e9 / HBM_PEAK_GBS
            # The one GEMM-shaped kernel gets the matrix-core yardstick: v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 operands) issued
            # per step (query workgroups of 256 x 4 waves x train tiles of 32 x 8 MFMAs per tile and wave) x 131 072 ops each,
            # over its time, against the dense fp4 peak of the guide (~10 PFLOP/s; its micro-benchmark floor is 9 099).  An HBM
            # fraction says nothing about it.  (The first half of round 3 ran it on v_mfma_i32_32x32x32_i8: twice the
            # instructions for the same 135 GOP, priced against 3 944 TOP/s.)
            nkp = float(np.mean([g[1] for g in gathered]))
            mfmas = B * int(np.ceil(nkp / 256.0)) * 4 * int(np.ceil(nkp / 32.0)) * 8
            FP4_PEAK = 10000.0
            mm = {"kernel": "k_match_mfma", "bound": "mfma", "peak": FP4_PEAK, "unit": "TFLOP/s", "mfma_per_step": mfmas, "ops_per_step": mfmas * 131072.0,
                  "what": "exact +-1 fp4 (E2M1) product, block scales 2^5: 1024 (256 - 2 Hamming) in the f32 accumulator, which IS the match key; "
                          "2 x 32 x 32 x 64 ops per v_mfma_scale_f32_32x32x64_f8f6f4"}
            for tag, kms in (("isolated", iso["kernel_ms_per_step"]), ("overlapped", out["roofline"].get("kernel_ms_per_step_all_bracketed", {}))):
                if kms.get("k_match_mfma"):
                    a = mfmas * 131072.0 / (kms["k_match_mfma"] * 1e-3) / 1e12
                    mm[tag] = {"ms_per_step": kms["k_match_mfma"], "achieved": a, "frac": a / FP4_PEAK}
            out["roofline"]["mfma"] = mm
            if sq_tab and args.config == "c3":
                cyc, clk = sq_tab["cycles_per_valu_instr"], sq_tab["clock_ghz"] * 1e9
                # (the counter passes also see the clock probe of the per-rank diagnosis: not a kernel of the step)
                tot = sum(v["valu"] for k, v in sq_tab["kernels"].items() if k != "k_clock_probe") * B / sq_tab["frames_per_step"]
                out["roofline"]["pipeline_issue_frac"] = tot * cyc / (SIMDS * clk * dt_max / args.steps)
        elif prof:
            out["roofline"] = roofline_of(prof, DOMINANT, steps_bracketed)  # --no-replay: the kernel named by the isolated runs so far
            out["roofline"]["overlapped_streams"] = True
        if hp is not None:
            out["host_path"] = hp
            if world > 1:  # every rank fed its own GPU at the same time, pinned to its NUMA node
                out["host_path"]["per_rank_pipelined_fps"] = [g[5] for g in gathered]
                out["host_path"]["per_rank_b1_ms_median"] = [g[6] for g in gathered]
        out["numa"] = pin
        # one line per rank: read THIS first when the driver's scaling efficiency is below ~0.95
        per_rank = []
        for r, g in enumerate(gathered):
            per_rank.append({"rank": r, "device": int(g[7]), "pci_bus_id": pci_text(g[8]), "numa_node": None if g[9] < 0 else int(g[9]),
                             "pinned_to_numa_node": bool(g[10] > 0), "ms_per_step": g[3] / args.steps * 1e3, "fps": g[0] / g[3] if g[3] > 0 else 0.0,
                             "shader_clock_mhz_after_region": None if g[11] < 0 else g[11], "parity_ok": None if g[4] < 0 else bool(g[4] > 0),
                             "keypoints_last_frame": int(g[1]), "matches_last_frame": int(g[2])})
        out["per_rank"] = per_rank
        fr_ = [q["fps"] for q in per_rank]
        out["per_rank_fps"] = {"min": min(fr_), "max": max(fr_), "spread": (max(fr_) - min(fr_)) / max(fr_) if max(fr_) > 0 else 0.0,
                               "distinct_devices": len({(q["device"], q["pci_bus_id"]) for q in per_rank})}
        if par is not None:
            out["parity_check"] = par
            if world > 1:
                out["parity_check"]["per_rank_ok"] = [bool(g[4] > 0) for g in gathered]
        if nat is not None:
            out["natural"] = nat
        if trk is not None:
            out["tracking_path"] = trk
        out["replayed_pmc"] = replay
        if ls is not None:
            out["live_streams"] = ls
        if dic is not None:
            out["dropin_classes"] = dic
        out["side_blocks"] = budget.record()
        if not args.no_cpu_baseline:
            # the CPU figure beside EVERY number.  N = 1: ~12 s on one thread + the one-thread-per-stream leg;
            # N > 1 (rank 0, the other ranks wait in finalize's barrier): a 6 s one-thread sample, so that a scaling sweep
            # does not spend its time on the host
            cpu_s = float(os.environ.get("ORBX_BENCH_CPU_SECONDS", "12.0" if world == 1 else "6.0"))   # (the CPU suite asks for less)
            out["cpu_baseline"] = cpu_baseline(cfg, first_batch, seconds_budget=cpu_s) if world == 1 else cpu_baseline(cfg, first_batch, seconds_budget=cpu_s, streams_n=0)
        line = json.dumps(out) + "\n"
        if json_fd is not None:
            os.write(json_fd, line.encode())
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    streams.finalize(world)
    if par is not None and not par["ok"]:
        return 4  # the metric promises bit-exact results
    if rank == 0 and any(0 <= g[4] < 0.5 for g in gathered):
        sys.stderr.write("bench.py: ranks %s differ from the CPU oracle\n" % [r for r, g in enumerate(gathered) if 0 <= g[4] < 0.5])
        return 4  # ... on EVERY rank: the launch fails when any GPU's results differ
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
