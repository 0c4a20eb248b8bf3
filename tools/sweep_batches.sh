#!/bin/bash
# Sweep of the batches-per-round policy: tools/sweep_batches.sh <scene> <stride> "<target> <max>" ...
set -e
SCENE=$1; STRIDE=$2; shift 2
mkdir -p gpurun_out/sweep
for tm in "$@"; do
  set -- $tm
  f=gpurun_out/sweep/${SCENE}_s${STRIDE}_t$1_m$2
  timeout -k 10 200 python3 bench.py --scene $SCENE --pixel-stride $STRIDE --target-paths $1 --max-batches $2 --steps 3 --warmup 1 --no-cpu-baseline > $f.json 2> $f.err
  echo "target $1 max $2: $(python3 tools/bench_brief.py $f.json | sed -n 1p)"
done
