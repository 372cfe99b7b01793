#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for w in 4 8 12 16 20 24 32; do
  PT_EXT_WAVES_PER_CU=$w timeout -k 10 200 python3 -u bench.py --cpu-seconds 0 --steps 16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves/CU $w', d['value'], d['roofline']['frame_ms'])"
done
export PATHTRACER_AMD_LIB=$PWD/variants/ovf32.so
for w in 20 24; do
  PT_EXT_WAVES_PER_CU=$w timeout -k 10 200 python3 -u bench.py --cpu-seconds 0 --steps 16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ovf32 waves/CU $w', d['value'], d['roofline']['frame_ms'])"
done
