"""Per hardware queue: kernel busy time and the idle gaps between consecutive kernels, from a rocprofv3 --kernel-trace database."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
byq = defaultdict(list)
for name, s, e, q in rows:
    byq[q].append((s, e, name))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    if len(ks) < 50:
        continue
    ks.sort()
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    gaps_us = sorted(g * 1e-3 for g in gaps)
    busy = sum(e - s for s, e, _ in ks) * 1e-3
    span = (ks[-1][1] - ks[0][0]) * 1e-3
    n = len(gaps_us)
    print("queue %s: %d kernels, span %.0f us, busy %.0f us (%.0f%%), gap median %.1f us, p90 %.1f, p99 %.1f, mean %.1f"
          % (q, len(ks), span, busy, 100 * busy / span, gaps_us[n // 2], gaps_us[int(n * 0.9)], gaps_us[int(n * 0.99)], sum(gaps_us) / n))
    # gaps by what follows
    after = defaultdict(list)
    for i in range(len(ks) - 1):
        nm = ks[i + 1][2]
        key = "extend" if "k_extend" in nm else "shade" if "k_shade" in nm else "raygen" if "k_raygen" in nm else "finish" if "k_finish" in nm else "other"
        after[key].append((ks[i + 1][0] - ks[i][1]) * 1e-3)
    print("   mean gap before:", {k: round(sum(v) / len(v), 1) for k, v in after.items()})
    dur = defaultdict(list)
    for s_, e_, nm in ks:
        key = "extend" if "k_extend" in nm else "shade" if "k_shade" in nm else "raygen" if "k_raygen" in nm else "other"
        dur[key].append((e_ - s_) * 1e-3)
    print("   mean duration us:", {k: round(sum(v) / len(v), 1) for k, v in dur.items()}, " max extend:", round(max(dur["extend"]) if dur["extend"] else 0, 1))

print("longest kernels:")
t0 = rows[0][1]
for name, s_, e_, q in sorted(rows, key=lambda r: r[1] - r[2])[:25]:
    print("   %8.1f us  queue %s  start %9.1f ms  %s" % ((e_ - s_) * 1e-3, q, (s_ - t0) * 1e-6, name[:60]))
