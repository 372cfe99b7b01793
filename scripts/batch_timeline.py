"""Per hardware queue and per batch of frames (a batch starts with k_raygen): start, end and summed kernel time of the
traversal and shading launches, from a rocprofv3 --kernel-trace database.  Shows how the chains' batches drift apart during a run.

usage: python scripts/batch_timeline.py <trace_results.db>"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
rows = [r for r in rows if "ptd::" in r[0]]
byq = defaultdict(list)
for n, s, e, q in rows:
    byq[q].append((s, e, n))
t0 = min(r[1] for r in rows)
batches = defaultdict(list)      # batch number -> list of (queue, start, end, ext, shade)
for q, ks in byq.items():
    k = -1
    cur = None
    for s, e, n in ks:
        if "k_raygen" in n:
            if cur:
                batches[k].append(cur)
            k += 1
            cur = [q, s, e, 0, 0, 0]
        if cur is None:
            continue
        cur[2] = max(cur[2], e)
        if "k_extend" in n:
            cur[3] += e - s
            cur[5] += 1
        elif "k_shade" in n:
            cur[4] += e - s
    if cur:
        batches[k].append(cur)
print("batch: chains, first start ms, spread of starts ms, last end ms, spread of ends ms, mean chain span ms, mean sum extend ms, mean sum shade ms")
for k in sorted(batches):
    b = batches[k]
    st = [x[1] for x in b]
    en = [x[2] for x in b]
    print("%3d: %2d  %9.2f %6.2f  %9.2f %6.2f  span %6.2f  ext %6.2f  shade %6.2f" % (
        k, len(b), (min(st) - t0) * 1e-6, (max(st) - min(st)) * 1e-6, (max(en) - t0) * 1e-6, (max(en) - min(en)) * 1e-6,
        sum(e - s for s, e in zip(st, en)) / len(b) * 1e-6, sum(x[3] for x in b) / len(b) * 1e-6, sum(x[4] for x in b) / len(b) * 1e-6))
