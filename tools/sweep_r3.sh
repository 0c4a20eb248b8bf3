#!/bin/bash
# Rank-of-8 rehearsal (every 8th pixel) and the whole frame under a launch-policy knob given in the environment.
# usage: tools/sweep_r3.sh VAR value...
set -e
mkdir -p gpurun_out/sw
VAR=$1; shift
for v in "$@"; do
  for st in 8 1; do
    name=${VAR}_${v}_s${st}
    env $VAR=$v timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --pixel-stride $st > gpurun_out/sw/$name.json 2> gpurun_out/sw/$name.err
    python3 - $name <<'PY'
import json, sys
d = json.load(open("gpurun_out/sw/%s.json" % sys.argv[1]))
print("%-40s %7.2f ms  %.4g samples/s  launches %d" % (sys.argv[1], d["ms_per_step"], d["value"], sum(d["roofline"]["launches"].values()) // d["steps"]))
PY
  done
done
