#!/bin/bash
# vector instructions per kernel of one bench.py run (one counter pass): scripts/pmc_valu.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmc_valu_$$
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $OUT -o pmc -- python3 $ROOT/bench.py --steps 16 --warmup 2 --cpu-seconds 0 "$@" > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - $OUT <<'PY'
import sqlite3, sys, glob, re
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, n, v in rows:
    if "ptd::k_shade" in k or "k_extend<false" in k:
        m = re.search(r"ptd::k_\w+(<[\w, ]+>)?", k)
        print(m.group(0), c, n, "%.4g" % v)
PY
