#!/bin/bash
# short PMC passes over bench.py; prints per-kernel averages for the extend and shade kernels
# usage: pmc_quick.sh "<counters of pass 1>" "<counters of pass 2>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass -d "$OUT/p$i" -o pmc -- python3 $ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 ${BENCH_ARGS:-} > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; exit 1; }
done
python3 - <<PY
import sqlite3, glob
for path in sorted(glob.glob("$OUT/*/pmc_results.db")):
    d = sqlite3.connect(path)
    rows = d.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_extend<false,%' or kernel_name like '%k_shade%' group by kernel_name, counter_name").fetchall()
    for k, c, n, a in rows:
        print(("extend" if "extend" in k else "shade"), c, n, "%.5g" % a)
PY
