#!/bin/bash
# L2 hits and misses of every k_debug_intersect dispatch of scripts/ray_order_probe.py, in order
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ray_order
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rop2
timeout -s KILL 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/rop2 -o t -- python3 $ROOT/scripts/ray_order_probe.py "$@" > $OUT/pmc.log 2>$OUT/pmc.err || { tail -5 $OUT/pmc.err; exit 1; }
python3 - <<P >> $OUT/pmc.log
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/rop2/**/*.db', recursive=True)[0])
rows = db.execute("select dispatch_id, counter_name, value from counters_collection where kernel_name like '%k_debug_intersect%' order by dispatch_id").fetchall()
by = {}
for d, c, v in rows:
    by.setdefault(d, {})[c] = by.setdefault(d, {}).get(c, 0.0) + v
print("L2 hit rate of the k_debug_intersect dispatches in order:")
print(" ".join("%.3f" % (v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum'])) for k, v in sorted(by.items())))
print("L2 misses per dispatch, millions:")
print(" ".join("%.2f" % (v['TCC_MISS_sum'] * 1e-6) for k, v in sorted(by.items())))
P
tail -5 $OUT/pmc.log
