"""Host-side mirror of the reference's ORBextractor (include/ORBextractor.h:45-111)
over the C ABI.  Same constructor arguments, same getters, `__call__` = operator().
All compute happens in HIP kernels; nothing here touches pixels."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbxParams, check, lib, ptr


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1  # ORBextractor.h:49 (unused enum in the reference too)

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST,
                 max_width=1280, max_height=720, max_batch=1, device=0, live=False):
        """ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        -- src/ORBextractor.cc:410.  device=-1 gives a host-only handle (tables only).  live=True: orbx_create_live -- every
        call of 1..max_batch (<= 8) frames runs as one latency-mode chain (the handle a camera hub creates)."""
        self._L = lib()
        self._h = C.c_void_p()
        prm = OrbxParams(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST))
        create = self._L.orbx_create_live if live else self._L.orbx_create
        check(create(C.byref(prm), int(max_width), int(max_height), int(max_batch), int(device), C.byref(self._h)))
        self.nfeatures, self.nlevels, self.max_batch, self.device = nfeatures, nlevels, max_batch, device
        self._cap = self._L.orbx_max_keypoints(self._h)
        self._last_shape = None
        self._dev_bufs = []

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            for d in getattr(self, "_dev_bufs", []):
                self._L.orbx_device_free(self._h, d)
            self._dev_bufs = []
            for p in getattr(self, "_pinned_results", []):
                self._L.orbx_host_free(self._h, p)
            self._pinned_results = []
            self._L.orbx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (ORBextractor.h:60-83)
    def GetLevels(self):
        return self._L.orbx_levels(self._h)

    def GetScaleFactor(self):
        return self._L.orbx_scale_factor(self._h)

    def _tables(self):
        n = self.GetLevels()
        a = [np.zeros(n, dtype=np.float32) for _ in range(4)]
        check(self._L.orbx_scale_tables(self._h, *[ptr(x) for x in a]))
        return a

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        out = np.zeros(self.GetLevels(), dtype=np.int32)
        check(self._L.orbx_features_per_level(self._h, ptr(out)))
        return out

    def umax(self):
        out = np.zeros(16, dtype=np.int32)
        check(self._L.orbx_umax(self._h, ptr(out)))
        return out

    @property
    def max_keypoints(self):
        return self._cap

    # ---- operator() (src/ORBextractor.cc:1043-1105); mask is ignored like in the reference
    def __call__(self, image, mask=None):
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, dtype=KP_DTYPE), np.zeros((0, 32), dtype=np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "image.type() == CV_8UC1"
        kps, desc = self.extract_batch(image[None])
        return kps[0], desc[0]

    def extract_batch(self, images):
        """images: [B,H,W] uint8 host array (or list of [H,W]); returns lists of (kps, desc)."""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        B = len(imgs)
        h, w = imgs[0].shape
        cap = self._cap
        kps = np.zeros((B, cap), dtype=KP_DTYPE)
        desc = np.zeros((B, cap, 32), dtype=np.uint8)
        n = np.zeros(B, dtype=np.int32)
        arr = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
        check(self._L.orbx_extract_batch(self._h, arr, B, w, h, w, ptr(kps), ptr(desc), cap, ptr(n)))
        self._last_shape = (w, h)
        return [kps[f, :n[f]].copy() for f in range(B)], [desc[f, :n[f]].copy() for f in range(B)]

    # ---- host-buffer entries with matching (orbx_extract_match_batch / orbx_submit_batch .. orbx_collect_*)
    def _opts(self, match, nnratio, th_low, check_ori):
        return _lib.OrbxStreamOpts(int(bool(match)), float(nnratio), int(th_low), int(bool(check_ori)))

    def _frame_ptrs(self, images):
        """(keep-alive list, pointer array, B, w, h, stride) for [B,H,W] uint8 arrays, lists of [H,W], or PinnedFrames"""
        if isinstance(images, PinnedFrames):
            B = images.B
            arr = (C.c_void_p * B)(*[images.ptr + f * images.pitch for f in range(B)])
            return images, arr, B, images.w, images.h, images.stride
        if isinstance(images, np.ndarray) and images.ndim == 3 and images.dtype == np.uint8 and images.flags.c_contiguous:
            B, h, w = images.shape
            base = images.ctypes.data
            arr = (C.c_void_p * B)(*[base + f * h * w for f in range(B)])
            return images, arr, B, w, h, w
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        h, w = imgs[0].shape
        return imgs, (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs]), len(imgs), w, h, w

    def _out_arrays(self, B):
        """reused output arrays of the synchronous host entries (no allocation per call on the latency path)"""
        o = getattr(self, "_out", None)
        if o is None or o[0].shape[0] < B:
            cap = self._cap
            o = (np.zeros((B, cap), dtype=KP_DTYPE), np.zeros((B, cap, 32), dtype=np.uint8), np.zeros(B, dtype=np.int32),
                 np.full((B, cap), -1, dtype=np.int32), np.zeros(B, dtype=np.int32))
            self._out = o
        return o

    def extract_match_host(self, images, match=True, nnratio=0.7, th_low=50, check_ori=True, copy=False):
        """orbx_extract_match_batch: host frames in, (kps, desc, n, match, nmatch) on the host; synchronous.
        Returns views into arrays this object reuses (copy=True for private copies)."""
        keep, arr, B, w, h, stride = self._frame_ptrs(images)
        kps, desc, n, m, nm = self._out_arrays(B)
        opts = self._opts(match, nnratio, th_low, check_ori)
        check(self._L.orbx_extract_match_batch(self._h, arr, B, w, h, stride, C.byref(opts), ptr(kps), ptr(desc), self._cap,
                                               ptr(n), ptr(m), ptr(nm)))
        self._last_shape = (w, h)
        res = (kps[:B], desc[:B], n[:B], m[:B], nm[:B])
        return tuple(a.copy() for a in res) if copy else res

    def submit_host(self, images, match=True, nnratio=0.7, th_low=50, check_ori=True):
        """orbx_submit_batch: returns a ticket; the frames must stay untouched until it is collected"""
        keep, arr, B, w, h, stride = self._frame_ptrs(images)
        t = C.c_int(-1)
        opts = self._opts(match, nnratio, th_low, check_ori)
        check(self._L.orbx_submit_batch(self._h, arr, B, w, h, stride, C.byref(opts), C.byref(t)))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[t.value] = keep
        return t.value

    def collect_host(self, ticket, view=True):
        """view=True: orbx_collect_view + copies of the per-frame counts only + orbx_release -- what a consumer that reads
        the pinned results in place costs; view=False: orbx_collect_batch into fresh arrays."""
        getattr(self, "_inflight", {}).pop(ticket, None)
        if view:
            v = _lib.OrbxBatchView()
            check(self._L.orbx_collect_view(self._h, int(ticket), C.byref(v)))
            n = np.ctypeslib.as_array(C.cast(v.n, C.POINTER(C.c_int32)), (v.B,)).copy()
            nm = np.ctypeslib.as_array(C.cast(v.nmatch, C.POINTER(C.c_int32)), (v.B,)).copy() if v.nmatch else None
            check(self._L.orbx_release(self._h, int(ticket)))
            return n, nm
        B, cap = self.max_batch, self._cap
        kps = np.zeros((B, cap), dtype=KP_DTYPE)
        desc = np.zeros((B, cap, 32), dtype=np.uint8)
        n = np.zeros(B, dtype=np.int32)
        m = np.full((B, cap), -1, dtype=np.int32)
        nm = np.zeros(B, dtype=np.int32)
        check(self._L.orbx_collect_batch(self._h, int(ticket), ptr(kps), ptr(desc), cap, ptr(n), ptr(m), ptr(nm)))
        return kps, desc, n, m, nm

    def alloc_pinned_results(self, B):
        """caller-owned PINNED result arrays (orbx_host_alloc) for submit_host_into: (kps, desc, n, match, nmatch) as numpy
        views; they live until close()"""
        cap = self._cap
        sizes = [B * cap * 28, B * cap * 32, B * 4, B * cap * 4, B * 4]
        offs = np.cumsum([0] + [(x + 63) // 64 * 64 for x in sizes])
        p = C.c_void_p()
        check(self._L.orbx_host_alloc(self._h, C.c_size_t(int(offs[-1])), C.byref(p)))
        self._pinned_results = getattr(self, "_pinned_results", []) + [p]

        def view(off, dtype, shape):
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            buf = (C.c_uint8 * n).from_address(p.value + int(off))
            return np.frombuffer(buf, dtype=dtype).reshape(shape)

        return (view(offs[0], KP_DTYPE, (B, cap)), view(offs[1], np.uint8, (B, cap, 32)), view(offs[2], np.int32, (B,)),
                view(offs[3], np.int32, (B, cap)), view(offs[4], np.int32, (B,)))

    def submit_host_into(self, images, out, match=True, nnratio=0.7, th_low=50, check_ori=True):
        """orbx_submit_batch_into: the results of the batch land in `out` (from alloc_pinned_results) -- no copy on collect"""
        keep, arr, B, w, h, stride = self._frame_ptrs(images)
        kps, desc, n, m, nm = out
        o = _lib.OrbxBatchOut(kps.ctypes.data, desc.ctypes.data, n.ctypes.data, m.ctypes.data, nm.ctypes.data, kps.shape[1])
        t = C.c_int(-1)
        opts = self._opts(match, nnratio, th_low, check_ori)
        check(self._L.orbx_submit_batch_into(self._h, arr, B, w, h, stride, C.byref(opts), C.byref(o), C.byref(t)))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[t.value] = (keep, out)
        return t.value

    def collect_into(self, ticket):
        """orbx_collect: waits; the arrays given to submit_host_into hold the results"""
        self._inflight.pop(ticket, None)
        check(self._L.orbx_collect(self._h, int(ticket)))

    def alloc_pinned_frames(self, B, w, h):
        """orbx_host_alloc_frames: pinned host frames in the device layout, as a PinnedFrames (numpy view in .array)"""
        p, st, pitch = C.c_void_p(), C.c_int(), C.c_size_t()
        check(self._L.orbx_host_alloc_frames(self._h, int(B), int(w), int(h), C.byref(p), C.byref(st), C.byref(pitch)))
        return PinnedFrames(self, p.value, B, w, h, st.value, pitch.value)

    def link_rate(self, up_bytes, down_bytes, reps=40):
        """orbx_debug_link_rate: GB/s of plain pinned copies -- (h2d alone, d2h alone, h2d and d2h while both run)"""
        a, b, c, d = C.c_float(0), C.c_float(0), C.c_float(0), C.c_float(0)
        check(self._L.orbx_debug_link_rate(self._h, C.c_size_t(int(up_bytes)), C.c_size_t(int(down_bytes)), int(reps),
                                           C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    # ---- device-resident path (frames already in HBM)
    def upload_frames(self, frames, stride=None):
        """copy [B,H,W] uint8 host frames into a device buffer with 64-byte aligned rows;
        returns (device_ptr, B, w, h, stride, frame_pitch).  The buffer lives until close()."""
        frames = np.asarray(frames, dtype=np.uint8)
        B, h, w = frames.shape
        stride = stride or (w + 63) // 64 * 64
        pad = np.zeros((B, h, stride), dtype=np.uint8)
        pad[:, :, :w] = frames
        d = C.c_void_p()
        check(self._L.orbx_device_alloc(self._h, C.c_size_t(pad.nbytes), C.byref(d)))
        self._dev_bufs.append(d)
        check(self._L.orbx_upload(self._h, d, ptr(pad), C.c_size_t(pad.nbytes)))
        return d.value, B, w, h, stride, stride * h

    def extract_batch_device(self, d_ptr, B, w, h, stride, frame_pitch):
        check(self._L.orbx_extract_batch_device(self._h, C.c_void_p(int(d_ptr)), int(B), int(w), int(h), int(stride),
                                                C.c_size_t(int(frame_pitch))))
        self._last_shape = (w, h)

    def match_prev_batch_device(self, nnratio=0.7, th_low=50, check_ori=True):
        check(self._L.orbx_match_prev_batch_device(self._h, C.c_float(nnratio), int(th_low), int(bool(check_ori))))

    def reset_stream(self):
        check(self._L.orbx_reset_stream(self._h))

    def sync(self):
        check(self._L.orbx_sync(self._h))

    def download(self, frame):
        cap = self._cap
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        check(self._L.orbx_download(self._h, int(frame), ptr(kps), ptr(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def download_matches(self, frame):
        cap = self._cap
        m = np.full(cap, -1, dtype=np.int32)
        nm = C.c_int(0)
        check(self._L.orbx_download_matches(self._h, int(frame), ptr(m), cap, C.byref(nm)))
        return m, nm.value

    def device_results(self):
        k, d, c, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        check(self._L.orbx_device_results(self._h, C.byref(k), C.byref(d), C.byref(c), C.byref(cap)))
        return k.value, d.value, c.value, cap.value

    def device_matches(self):
        m, n = C.c_void_p(), C.c_void_p()
        check(self._L.orbx_device_matches(self._h, C.byref(m), C.byref(n)))
        return m.value, n.value

    # ---- mvImagePyramid (ORBextractor.h:85), lazy download
    def pyramid_level(self, frame, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        check(self._L.orbx_pyramid_level(self._h, int(frame), int(level), int(blurred), None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), dtype=np.uint8)
        check(self._L.orbx_pyramid_level(self._h, int(frame), int(level), int(blurred), ptr(out), C.byref(w), C.byref(h)))
        return out

    def compute_stereo_matches(self, right, mb, mbf, frame=0):
        """Frame::ComputeStereoMatches (Frame.cc:466-638) with self as the left extractor: (mvuRight, mvDepth)"""
        cap = self._cap
        ur = np.zeros(cap, dtype=np.float32)
        dp = np.zeros(cap, dtype=np.float32)
        n = C.c_int(0)
        check(self._L.orbx_compute_stereo_matches(self._h, right._h, int(frame), C.c_float(mb), C.c_float(mbf), ptr(ur), ptr(dp), cap, C.byref(n)))
        return ur[:n.value].copy(), dp[:n.value].copy()

    def level_sizes(self):
        """(width, height) of every pyramid level of the last extracted shape (ORBextractor.cc:1111-1112)"""
        out = []
        for l in range(self.GetLevels()):
            w, h = C.c_int(), C.c_int()
            check(self._L.orbx_pyramid_level(self._h, 0, l, 0, None, C.byref(w), C.byref(h)))
            out.append((w.value, h.value))
        return out

    def level_candidates(self, frame, level):
        n = C.c_int()
        check(self._L.orbx_level_candidates(self._h, int(frame), int(level), None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint64)
        check(self._L.orbx_level_candidates(self._h, int(frame), int(level), ptr(out), out.shape[0], C.byref(n)))
        return out[:n.value]

    # ---- profiling
    def set_serial(self, on=True):
        check(self._L.orbx_set_serial(self._h, int(bool(on))))

    def profile_enable(self, on=True):
        check(self._L.orbx_profile_enable(self._h, int(bool(on))))

    def profile_select(self, kernel=None):
        """bracket only `kernel`'s launches (None: all kernels again)"""
        check(self._L.orbx_profile_select(self._h, kernel.encode() if kernel else None))

    def profile_read(self, reset=True):
        p = _lib.OrbxProfile()
        check(self._L.orbx_profile_read(self._h, C.byref(p), int(bool(reset))))
        return {p.name[i].decode(): (p.ms[i], p.launches[i]) for i in range(p.n)}


class PinnedFrames:
    """B pinned host frames in the device layout (rows `stride` bytes apart); .array is a [B, h, stride] numpy view,
    .array[:, :, :w] the pixels"""
    def __init__(self, owner, ptr_, B, w, h, stride, pitch):
        self.owner, self.ptr, self.B, self.w, self.h, self.stride, self.pitch = owner, ptr_, B, w, h, stride, pitch
        buf = (C.c_uint8 * (pitch * B)).from_address(ptr_)
        self.array = np.frombuffer(buf, dtype=np.uint8).reshape(B, h, stride)

    def fill(self, frames):
        self.array[:, :, :self.w] = frames

    def free(self):
        if self.ptr:
            self.array = None
            check(self.owner._L.orbx_host_free(self.owner._h, C.c_void_p(self.ptr)))
            self.ptr = 0


def unpack_candidates(rec):
    """u64 candidate records -> (x, y, response, order) arrays (orbx_common.hpp: pack_cand)"""
    rec = np.asarray(rec, dtype=np.uint64)
    x = (rec & np.uint64(0x1FFF)).astype(np.int32)
    y = ((rec >> np.uint64(13)) & np.uint64(0x1FFF)).astype(np.int32)
    resp = (rec >> np.uint64(56)).astype(np.int32)
    order = ((~(rec >> np.uint64(26))) & np.uint64(0x3FFFFFFF)).astype(np.int64)
    return x, y, resp, order
