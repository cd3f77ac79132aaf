"""The host-buffer entries (orbx_extract_match_batch, orbx_submit_batch / orbx_collect_*) against the oracle: what a
drop-in caller of Frame::ExtractORB sees -- pageable, pinned and registered frames, one frame per call and pipelined
batches, keypoints, descriptors AND the match table against the previous frame of the stream."""
import ctypes as C

import numpy as np
import pytest

from conftest import frames_for

pytestmark = pytest.mark.gpu


def _oracle_stream(oracle, frames, nf):
    """per frame: (kps, desc, match-vs-previous table, nmatch) of one camera stream"""
    ex = oracle.Extractor(nf, 1.2, 8, 20, 7)
    out, prev = [], None
    for f in frames:
        r = ex(f)
        if prev is None:
            m, n = np.full(len(r["kps"]), -1, dtype=np.int32), 0
        else:
            m, n = oracle.match_bruteforce(r["desc"], r["kps"]["angle"], prev["desc"], prev["kps"]["angle"], 0.7, 50, True)
        out.append((r["kps"], r["desc"], m, n))
        prev = r
    return out


def _check(ref, kps, desc, n, m, nm):
    rk, rd, rm, rn = ref
    assert n == len(rk)
    assert kps[:n].tobytes() == rk.tobytes()
    assert np.array_equal(desc[:n], rd)
    assert nm == rn and np.array_equal(m[:n], rm)


@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (1241, 376, 2000)])
def test_one_frame_per_call(gpu, oracle, w, h, nf):
    """B = 1, pageable frames: Frame::ExtractORB's entry followed by the match against the previous frame"""
    from orbslamm_amd import ORBextractor
    fr = frames_for(w, h, 5, stream=2)
    ref = _oracle_stream(oracle, fr, nf)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
    for t in range(5):
        kps, desc, n, m, nm = ex.extract_match_host(fr[t][None])
        _check(ref[t], kps[0], desc[0], int(n[0]), m[0], int(nm[0]))


@pytest.mark.parametrize("kind", ["pageable", "pinned_device_layout", "registered_tight", "pageable_strided"])
def test_pipelined_batches(gpu, oracle, kind):
    """three tickets in flight, collected in order; batches 5, 5, 5, 3, 2 of one stream (the batch size changes inside
    the pipeline) -- every frame and every match table equals the oracle's, also across batch boundaries"""
    from orbslamm_amd import ORBextractor
    w, h, nf = 640, 480, 1000
    sizes = [5, 5, 5, 3, 2]
    fr = frames_for(w, h, sum(sizes), stream=5)
    ref = _oracle_stream(oracle, fr, nf)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=5, device=0)
    L = ex._L
    pins, keep = [], []
    batches, o = [], 0
    for i, b in enumerate(sizes):
        chunk = fr[o:o + b]
        if kind == "pinned_device_layout":
            p = ex.alloc_pinned_frames(b, w, h)
            p.fill(chunk)
            pins.append(p)
            batches.append(p)
        elif kind == "registered_tight":
            a = np.ascontiguousarray(chunk)
            assert L.orbx_host_register(ex._h, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)) == 0
            keep.append(a)
            batches.append(a)
        elif kind == "pageable_strided":
            big = np.zeros((b, h, w + 37), dtype=np.uint8)
            big[:, :, :w] = chunk
            batches.append([big[f, :, :w] for f in range(b)])  # non-contiguous rows: the mirror makes them contiguous per frame
        else:
            batches.append(np.ascontiguousarray(chunk))
        o += b
    tickets, got = [], []
    for i, b in enumerate(batches):
        tickets.append(ex.submit_host(b))
        if len(tickets) == 3:
            got.append(ex.collect_host(tickets.pop(0), view=False))
    while tickets:
        got.append(ex.collect_host(tickets.pop(0), view=False))
    o = 0
    for i, b in enumerate(sizes):
        kps, desc, n, m, nm = got[i]
        for f in range(b):
            _check(ref[o + f], kps[f], desc[f], int(n[f]), m[f], int(nm[f]))
        o += b
    for a in keep:
        assert L.orbx_host_unregister(ex._h, C.c_void_p(a.ctypes.data)) == 0
    for p in pins:
        p.free()


def test_pipelined_batches_without_matching_read_in_place(gpu, oracle):
    """extraction-only tickets (no match tables) three deep, batch sizes 4, 4, 3, 4, 4 on a handle of four: a batch submitted
    behind others gathers its results in HBM and goes down as ONE copy of the block laid out for its own B -- read in place
    through orbx_collect_view (keypoints, descriptors, counts; match / nmatch are null)"""
    from orbslamm_amd import ORBextractor, _lib
    w, h, nf = 640, 480, 1000
    sizes = [4, 4, 3, 4, 4]
    fr = frames_for(w, h, sum(sizes), stream=11)
    ref = _oracle_stream(oracle, fr, nf)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=4, device=0)
    batches, o = [], 0
    for b in sizes:
        batches.append(np.ascontiguousarray(fr[o:o + b]))
        o += b
    got = []

    def take(t):
        ex._inflight.pop(t, None)
        v = _lib.OrbxBatchView()
        assert ex._L.orbx_collect_view(ex._h, t, C.byref(v)) == 0
        assert not v.match and not v.nmatch
        n = np.ctypeslib.as_array(C.cast(v.n, C.POINTER(C.c_int32)), (v.B,)).copy()
        frames = []
        for f in range(v.B):
            k = int(n[f])
            kps = np.frombuffer((C.c_uint8 * (k * 28)).from_address(v.kps + f * v.cap * 28), dtype=_lib.KP_DTYPE).copy()
            desc = np.frombuffer((C.c_uint8 * (k * 32)).from_address(v.desc + f * v.cap * 32), dtype=np.uint8).reshape(k, 32).copy()
            frames.append((kps, desc))
        assert ex._L.orbx_release(ex._h, t) == 0
        got.append(frames)

    tickets = []
    for b in batches:
        tickets.append(ex.submit_host(b, match=False))
        if len(tickets) == 3:
            take(tickets.pop(0))
    while tickets:
        take(tickets.pop(0))
    o = 0
    for i, b in enumerate(sizes):
        assert len(got[i]) == b
        for f in range(b):
            assert got[i][f][0].tobytes() == ref[o + f][0].tobytes() and np.array_equal(got[i][f][1], ref[o + f][1])
        o += b


def test_view_collect_and_ticket_errors(gpu, oracle):
    from orbslamm_amd import ORBextractor, OrbError, _lib
    w, h, nf = 320, 240, 500
    fr = frames_for(w, h, 4, stream=1)
    ref = _oracle_stream(oracle, fr, nf)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
    t = [ex.submit_host(fr[i][None]) for i in range(3)]
    with pytest.raises(OrbError) as e:   # a fourth batch needs a free slot
        ex.submit_host(fr[3][None])
    assert e.value.code == -1 and "in flight" in str(e.value)
    v = _lib.OrbxBatchView()
    assert ex._L.orbx_collect_view(ex._h, t[0], C.byref(v)) == 0
    n0 = C.cast(v.n, C.POINTER(C.c_int32))[0]
    kps = np.frombuffer((C.c_uint8 * (n0 * 28)).from_address(v.kps), dtype=_lib.KP_DTYPE)
    assert kps.tobytes() == ref[0][0].tobytes() and v.B == 1 and v.cap == ex.max_keypoints
    assert ex._L.orbx_collect_view(ex._h, t[0], C.byref(v)) == -1     # already collected
    assert ex._L.orbx_release(ex._h, t[0]) == 0
    assert ex._L.orbx_release(ex._h, t[0]) == -1                      # already released
    assert ex._L.orbx_release(ex._h, 99) == -1
    t.append(ex.submit_host(fr[3][None]))                             # the freed slot
    for i in (1, 2, 3):
        kps, desc, n, m, nm = ex.collect_host(t[i], view=False)
        _check(ref[i], kps[0], desc[0], int(n[0]), m[0], int(nm[0]))


def test_host_and_device_paths_interleave(gpu, oracle):
    """a stream may switch between the host entry and the device-resident entry: the previous-frame slot is shared"""
    from orbslamm_amd import ORBextractor
    w, h, nf = 640, 480, 1000
    fr = frames_for(w, h, 6, stream=8)
    ref = _oracle_stream(oracle, fr, nf)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2, device=0)
    kps, desc, n, m, nm = ex.extract_match_host(fr[0:2], copy=True)
    for f in range(2):
        _check(ref[f], kps[f], desc[f], int(n[f]), m[f], int(nm[f]))
    d = ex.upload_frames(fr[2:4])
    ex.extract_batch_device(*d)
    ex.match_prev_batch_device(0.7, 50, True)
    for f in range(2):
        k, dd = ex.download(f)
        mm, nmm = ex.download_matches(f)
        assert k.tobytes() == ref[2 + f][0].tobytes() and np.array_equal(dd, ref[2 + f][1])
        assert nmm == ref[2 + f][3] and np.array_equal(mm[:len(k)], ref[2 + f][2])
    kps, desc, n, m, nm = ex.extract_match_host(fr[4:6], copy=True)
    for f in range(2):
        _check(ref[4 + f], kps[f], desc[f], int(n[f]), m[f], int(nm[f]))


def test_soak_random_tickets(gpu, oracle):
    """40 tickets of random sizes (1..8 frames, so latency-mode and throughput-mode calls alternate inside one stream),
    pageable and pinned sources mixed, up to three in flight, collected as views or copies: every frame and every match
    table equals the oracle's -- the hazards of the three-deep pipeline (result sets, match tables, staging slots, the
    previous-frame slot) under an irregular schedule"""
    from orbslamm_amd import ORBextractor
    w, h, nf = 640, 480, 1000
    rng = np.random.default_rng(77)
    sizes = [int(rng.integers(1, 9)) for _ in range(40)]
    fr = frames_for(w, h, sum(sizes), stream=6)
    ref = _oracle_stream(oracle, fr, nf)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=8, device=0)
    pins = [ex.alloc_pinned_frames(8, w, h) for _ in range(3)]
    tickets, got, o = [], [], 0
    for i, b in enumerate(sizes):
        chunk = fr[o:o + b]
        if rng.integers(0, 2):
            p = pins[i % 3]
            p.array[:b, :, :w] = chunk
            src = [p.array[f, :, :w] for f in range(b)] if rng.integers(0, 2) else None
            if src is None:   # the pinned buffer itself (device layout), first b frames
                from orbslamm_amd.extractor import PinnedFrames
                src = PinnedFrames(p.owner, p.ptr, b, w, h, p.stride, p.pitch)
        else:
            src = np.ascontiguousarray(chunk)
        tickets.append((ex.submit_host(src), b))
        o += b
        while len(tickets) == 3 or (tickets and rng.integers(0, 3) == 0):
            t, bb = tickets.pop(0)
            got.append((ex.collect_host(t, view=False), bb))
    while tickets:
        t, bb = tickets.pop(0)
        got.append((ex.collect_host(t, view=False), bb))
    o = 0
    for (kps, desc, n, m, nm), b in got:
        for f in range(b):
            _check(ref[o + f], kps[f], desc[f], int(n[f]), m[f], int(nm[f]))
        o += b
    assert o == sum(sizes)


def test_results_straight_into_caller_arrays(gpu, oracle):
    """orbx_submit_batch_into / orbx_collect: keypoints, descriptors and match tables land in the caller's own pinned
    arrays, exactly n records per frame, pipelined three deep -- equal to the oracle and to the copy-out entry"""
    from orbslamm_amd import ORBextractor, OrbError, synth
    w, h, nf, B = 640, 480, 1000, 4
    fr = synth.make_frames(w, h, 3 * B, stream=6)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=0)
    outs = [ex.alloc_pinned_results(B) for _ in range(3)]
    for o in outs:
        o[0]["x"][:] = -7.0   # stale content must not survive where records are written
    tickets = [ex.submit_host_into(fr[i * B:(i + 1) * B], outs[i]) for i in range(3)]
    for t in tickets:
        ex.collect_into(t)
    oex = oracle.Extractor(nf, 1.2, 8, 20, 7)
    prev = None
    for i in range(3):
        kps, desc, n, m, nm = outs[i]
        for f in range(B):
            r = oex(fr[i * B + f])
            assert n[f] == len(r["kps"]) and kps[f, :n[f]].tobytes() == r["kps"].tobytes() and desc[f, :n[f]].tobytes() == r["desc"].tobytes()
            assert kps[f, n[f]]["x"] == -7.0   # nothing past the n-th record
            if prev is not None:
                wm, wn = oracle.match_bruteforce(r["desc"], r["kps"]["angle"], prev["desc"], prev["kps"]["angle"], 0.7, 50, True)
                assert nm[f] == wn and np.array_equal(m[f, :n[f]], wm)
            prev = r
    # pageable output arrays are refused (the device could not write them), and the ticket kinds do not mix
    ex2 = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=0)
    bad = tuple(np.zeros_like(a) for a in outs[0])
    with pytest.raises(OrbError):
        ex2.submit_host_into(fr[:B], bad)
    t = ex2.submit_host(fr[:B])
    with pytest.raises(OrbError):
        ex2.collect_into(t)
    ex2.collect_host(t)


@pytest.mark.parametrize("kind", ["pageable", "pinned_rings"])
def test_live_handle_chains_of_up_to_eight_frames(gpu, oracle, kind):
    """orbx_create_live: calls of 1..8 frames all run as ONE latency-mode chain (what orbslamm::CameraHub submits for the
    robots that wait together).  Batches of 5, 8, 1, 3, 8 frames of one stream with the brute-force match against the
    previous frame (a batch's frames are consecutive frames of the stream), pageable frames and pinned frames that do NOT
    lie back to back (a ring buffer per camera: k_upload_frames) -- every frame and match table equals the oracle's."""
    from orbslamm_amd import ORBextractor
    w, h, nf = 640, 480, 1000
    sizes = [5, 8, 1, 3, 8]
    fr = frames_for(w, h, sum(sizes), stream=9)
    ref = _oracle_stream(oracle, fr, nf)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=8, device=0, live=True)
    with pytest.raises(Exception):
        ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=9, device=0, live=True)
    L = ex._L
    pins = []
    if kind == "pinned_rings":
        # every frame in a pinned block of its own, in the device layout (stride = width rounded up to 64)
        stride = (w + 63) // 64 * 64
        for f in fr:
            p = C.c_void_p()
            assert L.orbx_host_alloc(ex._h, C.c_size_t(stride * h), C.byref(p)) == 0
            buf = np.frombuffer((C.c_uint8 * (stride * h)).from_address(p.value), dtype=np.uint8).reshape(h, stride)
            buf[:, :w] = f
            pins.append(p)
    t0 = 0
    for B in sizes:
        tk = C.c_int(-1)
        opts = ex._opts(True, 0.7, 50, True)
        if kind == "pinned_rings":
            arr = (C.c_void_p * B)(*[pins[t0 + i].value for i in range(B)])
            stride = (w + 63) // 64 * 64
        else:
            arr = (C.c_void_p * B)(*[fr[t0 + i].ctypes.data for i in range(B)])
            stride = w
        assert L.orbx_submit_batch(ex._h, arr, B, w, h, stride, C.byref(opts), C.byref(tk)) == 0, L.orbx_last_error()
        kps, desc, n, m, nm = ex.collect_host(tk.value, view=False)
        for i in range(B):
            _check(ref[t0 + i], kps[i], desc[i], int(n[i]), m[i], int(nm[i]))
        t0 += B
    for p in pins:
        L.orbx_host_free(ex._h, p)
