#!/bin/bash
# A/B of device-layer builds: tools/ab.sh <name>... (names under pathtracer_amd/_variants, or "base")
# Each run is its own process under its own time limit; stops at the first failure.
set -e
mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" = base ]; then unset PTAMD_LIB; else export PTAMD_LIB=$PWD/pathtracer_amd/_variants/libptamd_$v.so; fi
  timeout -k 10 200 python3 bench.py $AB_ARGS --scene bunny --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab/${v}_bunny.json 2> gpurun_out/ab/${v}_bunny.err
  timeout -k 10 300 python3 bench.py $AB_ARGS --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab/${v}_dragon.json 2> gpurun_out/ab/${v}_dragon.err
  echo "== $v"
  python3 tools/bench_brief.py gpurun_out/ab/${v}_bunny.json | sed -n 1,2p
  python3 tools/bench_brief.py gpurun_out/ab/${v}_dragon.json | sed -n 1,2p
done
