#!/usr/bin/env python3
"""Compare the HIP-API call counts of two rocprofv3 --hip-runtime-trace --stats runs of the same program at two loop
lengths: calls per extra iteration, per API.  usage: hip_api_diff.py <dir_short> <dir_long> <extra_iterations>"""
import csv
import glob
import sys

CREATORS = ("hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipStreamCreate", "hipStreamDestroy", "hipEventCreate",
            "hipEventDestroy", "hipMallocAsync", "hipFreeAsync", "hipExtMallocWithFlags", "hipHostRegister", "hipHostUnregister",
            "hipModuleLoad", "hipFuncSetAttribute", "hipMemcpy", "hipMemset", "hipStreamSynchronize", "hipDeviceSynchronize")


def counts(d):
    out = {}
    files = glob.glob(d + "/**/*hip_api_stats.csv", recursive=True)
    if not files:
        files = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)
        for f in files:
            for row in csv.DictReader(open(f)):
                out[row["Function"]] = out.get(row["Function"], 0) + 1
        return out
    for f in files:
        for row in csv.DictReader(open(f)):
            out[row["Name"]] = out.get(row["Name"], 0) + int(row["Calls"])
    return out


def main():
    a, b, extra = counts(sys.argv[1]), counts(sys.argv[2]), int(sys.argv[3])
    print("# rocprofv3 --hip-runtime-trace --stats of examples/tracking_loop at two lengths (%d more frames in the second)" % extra)
    print("%-36s %10s %10s %14s" % ("HIP API", "short", "long", "per frame"))
    bad = []
    for name in sorted(set(a) | set(b), key=lambda n: -(b.get(n, 0))):
        per = (b.get(name, 0) - a.get(name, 0)) / float(extra)
        print("%-36s %10d %10d %14.3f" % (name, a.get(name, 0), b.get(name, 0), per))
        if any(name.startswith(c) for c in CREATORS[:13]) and b.get(name, 0) != a.get(name, 0):
            bad.append(name)
    print("allocation / stream / event APIs whose count grew with the loop: %s" % (", ".join(bad) if bad else "none"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
