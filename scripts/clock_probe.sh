#!/bin/bash
# samples the GPU's shader clock (rocm-smi) every ~30 ms while bench.py runs a long headline pass: does the clock ramp after the idle
# moment before the timed region?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/clock_probe.txt
: > $OUT
( for i in $(seq 1 400); do echo "$(date +%s.%N) $(cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*' | tr '\n' ' ')" >> $OUT; sleep 0.03; done ) &
SAMPLER=$!
timeout -k 10 100 python3 $ROOT/bench.py --steps 256 --warmup 5 --cpu-seconds 0 --others 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'])"
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
echo samples $(wc -l < $OUT)
awk '{print $2,$3,$4}' $OUT | sort | uniq -c | sort -rn | head -12
