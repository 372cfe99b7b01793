#!/bin/bash
# bench.py over build flags: scripts/ab_flags.sh "0 16 17 24" --workload bistro_class
cd "${GRAFT_REPO_ROOT:-.}"
flags="$1"; shift
for f in $flags; do
  timeout -k 10 200 python3 -u bench.py --cpu-seconds 0 --steps 128 --build-flags $f "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['config']['hierarchy']; print('flags $f', d['value'], h['sah_cost'], h['nodes'], h.get('build_ms'), d['roofline'].get('nodes_per_ray'), d['roofline'].get('tris_per_ray'))" || echo "flags $f failed"
done
