#!/bin/bash
# one SQ counter pass over a short bench run; prints per-kernel averages
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d "$OUT/sq" -o pmc -- python3 $ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 > "$OUT/sq.log" 2>&1 || { tail -5 "$OUT/sq.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS -d "$OUT/sq2" -o pmc -- python3 $ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 > "$OUT/sq2.log" 2>&1 || { tail -5 "$OUT/sq2.log"; exit 1; }
python3 - <<PY
import sqlite3, glob
for path in sorted(glob.glob("$OUT/*/pmc_results.db")):
    d = sqlite3.connect(path)
    rows = d.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_extend<false>%' or kernel_name like '%k_shade%' group by kernel_name, counter_name").fetchall()
    for k, c, n, a in rows:
        print(("extend" if "extend" in k else "shade"), c, n, "%.4g" % a)
PY
