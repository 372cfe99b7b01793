#!/bin/bash
# On the GPU box: profiles bench.py (scripts/profile.sh), summarises the databases there (they are too large to travel back) and
# leaves the summaries under gpurun_out/profiles_<tag>/ for copying into profiles/.
# usage: scripts/profile_and_summarize.sh <tag> <commit> [bench.py arguments, e.g. --workload bistro_class]
set -u
TAG=$1; COMMIT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
RAW=/tmp/prof_$TAG
rm -rf "$RAW"
BENCH_ARGS="$*" PROF_OUT=$RAW bash "$ROOT/scripts/profile.sh" "$TAG" > "$ROOT/gpurun_out/profile_$TAG.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/profile_$TAG.log"; exit 1; }
cd "$ROOT" && python3 scripts/summarize_profile.py "$RAW" "$TAG" "$COMMIT" > "gpurun_out/summary_$TAG.log" 2>&1 || { tail -5 "gpurun_out/summary_$TAG.log"; exit 1; }
mkdir -p "gpurun_out/profiles_$TAG"
cp profiles/${TAG}_* profiles/counters_per_ray_*.json "gpurun_out/profiles_$TAG/" 2>/dev/null
grep '^{"metric"' "$RAW/trace.log" | tail -1 > "gpurun_out/profiles_$TAG/${TAG}_bench_line_profiled.json"
ls "gpurun_out/profiles_$TAG"
