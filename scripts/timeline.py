"""How full the chip is over time, from a rocprofv3 --kernel-trace database: per time bin the number of kernels running
side by side (busy time of all kernels in the bin / bin width), traversal and shading apart.  Shows the ramp at the start of a
timed region and the drain at its end.

usage: python scripts/timeline.py <trace_results.db> [bin_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
bin_ns = int(float(sys.argv[2]) * 1e3) if len(sys.argv) > 2 else 500_000
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = [(n, s, e) for n, s, e in rows if "ptd::" in n]
t0 = rows[0][1]
t1 = max(e for _, _, e in rows)
nb = (t1 - t0) // bin_ns + 1
ext = [0.0] * nb
shd = [0.0] * nb
oth = [0.0] * nb
for n, s, e in rows:
    arr = ext if "k_extend" in n else shd if "k_shade" in n else oth
    b = (s - t0) // bin_ns
    while s < e:
        edge = t0 + (b + 1) * bin_ns
        seg = min(e, edge) - s
        arr[b] += seg
        s += seg
        b += 1
print("bin %.0f us; columns: time ms, traversal launches running, shading launches running, other" % (bin_ns * 1e-3))
last_idle = True
for i in range(nb):
    a, b, c = ext[i] / bin_ns, shd[i] / bin_ns, oth[i] / bin_ns
    idle = a + b + c < 0.01
    if idle and last_idle:
        continue
    last_idle = idle
    print("%8.2f  %5.2f %5.2f %5.2f  %s" % (i * bin_ns * 1e-6, a, b, c, "#" * int(round(a + b + c))))
