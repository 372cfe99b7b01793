# A/B of the captured-graph replay (PT_GRAPH=1) against plain launches at several run lengths
for s in 20 64; do for g in 0 1 0 1; do PT_GRAPH=$g timeout -k 10 100 python3 bench.py --steps $s --warmup 5 --cpu-seconds 0 --others 0 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph $g steps', $s, d['value'], d['ms_per_step'])"; done; done
