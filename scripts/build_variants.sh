#!/bin/bash
# builds tuning variants of the backend into build/variants/<name>.so: name=DEFINES pairs on the command line
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result $defs -shared -o ../../variants/$name.so api.hip trace.hip bvh_build.hip passes.hip &
done
wait
ls -la ../../variants
