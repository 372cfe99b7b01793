#!/bin/bash
# runs scripts/ray_order_probe.py under rocprofv3 --kernel-trace and lists the duration of every k_debug_intersect dispatch in order
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ray_order
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rop
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/rop -o t -- python3 $ROOT/scripts/ray_order_probe.py "$@" > $OUT/probe.log 2>$OUT/probe.err || { tail -5 $OUT/probe.err; exit 1; }
python3 - <<P >> $OUT/probe.log
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/rop/**/*.db', recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
print("k_debug_intersect dispatches in order, ms:")
print(" ".join("%.3f" % ((e - s) * 1e-6) for n, s, e in rows if "k_debug_intersect" in n))
P
cat $OUT/probe.log
