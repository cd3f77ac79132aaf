"""sha256[:16] over the kernel sources (orbslamm_amd/csrc/*_kernels.hip, orbx_common.hpp): stamped into the PMC tables under
profiles/ when they are collected, compared by bench.py when it replays them (`roofline.traffic`, `roofline.issue`)"""
import glob
import hashlib
import os


def kernels_sha():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orbslamm_amd", "csrc")
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(root, "*_kernels.hip")) + [os.path.join(root, "orbx_common.hpp")]):
        h.update(open(p, "rb").read())
    # the one run-time switch that selects other KERNELS (the popcount scan instead of the matrix-core scan): a table taken
    # under it is not a table of the default build.  (ORBX_SERIAL, under which the counter passes run, only removes overlap.)
    if os.environ.get("ORBX_MATCH_POPCOUNT") == "1":
        h.update(b"ORBX_MATCH_POPCOUNT=1")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernels_sha())
