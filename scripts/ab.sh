#!/bin/bash
# A/B: runs bench.py once per variant library in variants/ (and the default build); prints value and frame times
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for so in default variants/*.so; do
  if [ "$so" = default ]; then unset PATHTRACER_AMD_LIB; else export PATHTRACER_AMD_LIB=$PWD/$so; fi
  timeout -k 10 200 python3 -u bench.py --cpu-seconds 0 --steps ${STEPS:-256} "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so', d['value'], d['roofline']['frame_ms'], d['roofline']['nodes_per_ray'], d['roofline']['tris_per_ray'])" || echo "$so failed"
done
