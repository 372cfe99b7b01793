for rep in 1 2; do
for lib in variants/head.so variants/capin.so variants/flagonly.so variants/r02tree/pathtracer_amd/libpathtracer_amd.so; do
export PATHTRACER_AMD_LIB=$PWD/$lib
PT_DEFER=2 timeout -k 10 100 python3 variants/r02tree/bench.py --workload sponza_class --steps 64 --warmup 5 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('defer2 sponza_class white, $lib', d['value'])"
done
for lib in variants/head.so variants/capin.so variants/flagonly.so; do
export PATHTRACER_AMD_LIB=$PWD/$lib
timeout -k 10 100 python3 bench.py --steps 64 --warmup 5 --cpu-seconds 0 --others 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline, $lib', d['value'])"
done
done
