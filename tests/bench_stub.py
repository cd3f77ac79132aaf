"""A stand-in with the ORBextractor mirror's methods and no GPU behind it, for the CPU plumbing tests of bench.py's
N > 1 path (launcher, rank -> stream / device map, one JSON line, MAX over ranks).  Selected with
ORBX_BENCH_EXTRACTOR=tests.bench_stub:StubExtractor; it is test infrastructure and computes nothing."""
import json
import os
import time

import numpy as np


class StubExtractor:
    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width=0, max_height=0, max_batch=1, device=0):
        self.nfeatures, self.nlevels, self.max_batch, self.device = nfeatures, nlevels, max_batch, device
        self.w, self.h = max_width, max_height
        self.rank = int(os.environ.get("RANK", "0"))
        self.steps = 0
        self.uploads = []

    def upload_frames(self, frames, stride=None):
        self.uploads.append(int(np.asarray(frames, dtype=np.uint8).sum() % 1000003))  # a fingerprint of this rank's stream
        return (len(self.uploads) - 1, frames.shape[0], frames.shape[2], frames.shape[1], stride or frames.shape[2], 0)

    def extract_batch_device(self, *a):
        self.steps += 1
        if os.environ.get("ORBX_BENCH_STUB_FAIL_RANK") == str(self.rank) and self.steps == 2:
            raise RuntimeError("stub: rank %d fails on purpose" % self.rank)
        time.sleep(0.002 * (self.rank + 1))  # rank r is (r + 1) x slower: the record must carry the slowest rank's time

    def match_prev_batch_device(self, *a):
        pass

    def sync(self):
        pass

    def reset_stream(self):
        pass

    def profile_enable(self, on=True):
        self.prof_on = bool(on)

    def profile_select(self, k=None):
        self.prof_kernel = k

    def profile_read(self, reset=True):
        # {kernel: (total ms, launches)} like the mirror's; ORBX_BENCH_STUB_PROFILE=1: a made-up span for the kernel bench.py
        # selected, so that the record's `roofline` block is assembled on the N > 1 path too (numbers mean nothing)
        if os.environ.get("ORBX_BENCH_STUB_PROFILE") == "1" and getattr(self, "prof_kernel", None):
            return {self.prof_kernel: (0.05 * max(self.steps, 1), 4 * max(self.steps, 1))}
        return {}

    def set_serial(self, on=True):
        pass

    def download(self, frame):
        return np.zeros(1500 + self.rank, dtype=np.uint8), None

    def download_matches(self, frame):
        dump = os.environ.get("ORBX_BENCH_STUB_DUMP")
        if dump:  # what this rank did, for the test to read
            with open(os.path.join(dump, "rank%d.json" % self.rank), "w") as f:
                json.dump({"rank": self.rank, "device": self.device, "steps": self.steps, "uploads": self.uploads,
                           "world": int(os.environ.get("WORLD_SIZE", "1"))}, f)
        return None, 700 + self.rank

    def level_sizes(self):
        out, w, h = [], self.w, self.h
        for _ in range(self.nlevels):
            out.append((w, h))
            w, h = int(round(w / 1.2)), int(round(h / 1.2))
        return out
