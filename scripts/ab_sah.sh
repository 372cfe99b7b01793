#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for so in default variants/*.so; do
  if [ "$so" = default ]; then unset PATHTRACER_AMD_LIB; else export PATHTRACER_AMD_LIB=$PWD/$so; fi
  timeout -k 10 200 python3 -u bench.py --cpu-seconds 0 --steps 128 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so', d['value'], d['config']['hierarchy']['sah_cost'], d['config']['hierarchy']['nodes'], d['config']['hierarchy'].get('build_ms'), d['roofline']['nodes_per_ray'], d['roofline']['tris_per_ray'])" || echo "$so failed"
done
