#!/bin/bash
# on the GPU box: the pipelined host entries (three tickets of 64 pinned frames in flight) on one time base -- engine copies
# (frames up in two parts, results down) and, per ticket, the first and last kernel -- to see which resource is busy when
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hpprof
AB_SECONDS=0.15 timeout 250 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/hpprof -o t -- python $R/tools/down_engine_ab.py > /dev/null 2>&1
python - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/hpprof/t_results.db")
cols = [r[1] for r in c.execute("pragma table_info(memory_copies)")]
print("# memory_copies columns:", cols)
size = "size" if "size" in cols else None
cp = c.execute("select start, end%s from memory_copies order by start" % (", " + size if size else ", 0")).fetchall()
ks = c.execute("select start, end, name from kernels order by start").fetchall()
# the pinned loop is the first half of the run; take a window of ~4 ms from its middle
t_lo = cp[len(cp) // 4][0]
t_hi = t_lo + 4.2e6
ev = [(s, e, "COPY %8d B" % b) for s, e, b in cp if t_lo <= s < t_hi]
for s, e, n in ks:
    if t_lo <= s < t_hi:
        n = n.split("(")[0].split("::")[-1][:24]
        if n in ("k_pyramid", "k_pack_host", "k_match_mfma", "k_roll_prev", "k_match_accept_prune") or not n.startswith("k_"):
            ev.append((s, e, n))
ev.sort()
t0 = ev[0][0]
busy_up = sum(e - s for s, e, b in cp if t_lo <= s < t_hi and b > 4e6 and b != 8519680)
busy_dn = sum(e - s for s, e, b in cp if t_lo <= s < t_hi and b == 8519680)
last = None
for s, e, n in ev:
    if n == "k_pyramid" and last == "k_pyramid":
        continue
    last = n
    print("%9.1f %9.1f %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
import collections
print("# kernel names in the run:", dict(collections.Counter(n.split("(")[0].split("::")[-1][:24] for _, _, n in ks)))
print("# copies by size:", dict(collections.Counter(b for _, _, b in cp)))
print("# window %.1f ms: upload engine busy %.2f ms, download engine busy %.2f ms" % ((t_hi - t_lo) / 1e6, busy_up / 1e6, busy_dn / 1e6))
PY
