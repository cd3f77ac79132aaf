#!/usr/bin/env python3
"""profiles/<tag>_pmc_sq.json from the SQ_* rocprofv3 --pmc passes (ORBX_SERIAL=1: every kernel alone on the GPU).
Per kernel and 64-frame step (summed over the kernel's dispatches of a step): wave-instruction counts by class, LDS
bank-conflict cycles, waves; plus the shader clock of the run (GRBM_GUI_ACTIVE per XCD over the kernel's duration) and the
measured SIMD cycles a wave64 VALU instruction occupies (tools/ubench/valu_rate2.hip: 4.3 VOP2 .. 5.2, 4.8 for these kernels'
mix).  bench.py turns it into `roofline.issue`: VALU wave-instructions x cycles / (1024 SIMDs x clock x time).
usage: pmc_sq_json.py <out.json> <db> [<db> ...]"""
import json
import sqlite3
import sys

WANT = {"SQ_INSTS_VALU": "valu", "SQ_INSTS_SALU": "salu", "SQ_INSTS_LDS": "lds", "SQ_INSTS_VMEM_RD": "vmem_rd", "SQ_INSTS_VMEM_WR": "vmem_wr",
        "SQ_INSTS_SMEM": "smem", "SQ_LDS_BANK_CONFLICT": "lds_bank_conflict", "SQ_WAVES": "waves", "SQ_INSTS_MFMA": "mfma",
        "SQ_WAIT_INST_ANY": "wait_inst_any", "SQ_BUSY_CYCLES": "sq_busy_cycles", "SQ_VALU_MFMA_BUSY_CYCLES": "mfma_busy_cycles",
        "SQ_ACTIVE_INST_VALU": "active_inst_valu", "SQ_ACTIVE_INST_LDS": "active_inst_lds", "SQ_WAVE_CYCLES": "wave_cycles"}


def short(name):
    return name.split("(")[0].split("::")[-1].split("<")[0]


out = {"frames_per_step": 64, "cycles_per_valu_instr": 4.8, "simds": 1024, "kernels": {}, "source": "rocprofv3 --pmc, ORBX_SERIAL=1, bench.py --steps 5 --warmup 2"}
clock = []
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    nsteps = max([n for k, cn, n, v, d in rows if short(k) == "k_pyramid"] or [1])  # one launch per step when ORBX_SERIAL=1
    for k, cn, n, v, d in rows:
        k = short(k)
        if k.startswith("__amd"):
            continue
        e = out["kernels"].setdefault(k, {})
        if cn in WANT:
            e[WANT[cn]] = e.get(WANT[cn], 0.0) + v * n / nsteps
        if cn == "GRBM_GUI_ACTIVE" and d > 50000:   # kernels of >= 50 us only: the short ones are mostly fixed cost
            clock.append(v / 8.0 / d)   # cycles summed over the 8 XCDs / ns
        e["dispatches_per_step"] = n / nsteps
        e.setdefault("dur_us_per_step", 0.0)
    durs = {}
    for k, n, d in c.execute("select kernel_name, count(*), avg(duration) from counters_collection group by kernel_name"):
        pass
out["clock_ghz"] = round(sum(clock) / len(clock), 3) if clock else 2.4
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernels_sha import kernels_sha  # noqa: E402
out["kernels_sha16"] = kernels_sha()   # bench.py replays this table: it says so when the kernels have changed since
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps({k: {a: round(b) for a, b in v.items()} for k, v in out["kernels"].items()}, indent=1))
print("clock_ghz", out["clock_ghz"])
