#!/bin/bash
# Profiles bench.py on the GPU box: one kernel-trace/stats run and separate PMC passes
# (counters are never combined with other trace domains).  Results land in gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${PROF_OUT:-$ROOT/gpurun_out/prof_$TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 16 --warmup 2 --cpu-seconds 0 --others 0 ${BENCH_ARGS:-}"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 $ARGS > "$OUT/trace.log" 2>&1 || { echo "kernel-trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $pass -d "$OUT/pmc_$name" -o pmc -- python3 $ARGS > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $pass failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
find "$OUT" -name "*.csv" | head -40
