#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for d in 0.1 0.25 0.5 1.0 2.0 4.0; do
  PT_WAVE_STATS=1 timeout -k 10 300 python3 -u bench.py --cpu-seconds 0 --steps 8 --detail $d 2>gpurun_out/ws.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('detail $d tris', d['config']['triangles'], 'nodes', d['config']['hierarchy']['nodes'], 'Mrays/s', d['value'], d['roofline']['frame_ms'], 'n/ray', d['roofline']['nodes_per_ray'], 't/ray', d['roofline']['tris_per_ray'])"
  tail -1 gpurun_out/ws.log
done
