#!/bin/bash
# tools/sweep2.sh <scene> <stride> "<max_paths> <target> <max_batches>" ...
set -e
SCENE=$1; STRIDE=$2; shift 2
mkdir -p gpurun_out/sweep
for tm in "$@"; do
  set -- $tm
  f=gpurun_out/sweep/${SCENE}_s${STRIDE}_p$1_t$2_m$3
  timeout -k 10 200 python3 bench.py --scene $SCENE --pixel-stride $STRIDE --max-paths $1 --target-paths $2 --max-batches $3 --steps 3 --warmup 1 --no-cpu-baseline > $f.json 2> $f.err
  echo "$SCENE stride $STRIDE max_paths $1 target $2 max $3: $(python3 tools/bench_brief.py $f.json | sed -n 1p | cut -c1-40)"
done
