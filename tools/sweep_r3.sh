#!/bin/bash
# Rank-of-8 rehearsal (every 8th pixel) under the look-ahead margin and the read-back threshold.
set -e
mkdir -p gpurun_out/sw
run() {  # name, env margin, bench args...
  local name=$1 m=$2; shift 2
  PTAMD_PLAN_MARGIN=$m timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/sw/$name.json 2> gpurun_out/sw/$name.err
  python3 - $name <<'PY'
import json, sys
d = json.load(open("gpurun_out/sw/%s.json" % sys.argv[1]))
print("%-24s %7.2f ms  %.4g samples/s" % (sys.argv[1], d["ms_per_step"], d["value"]))
PY
}
for m in 1.5 2 3; do
  for sb in 0 2048 16384; do
    run s8_m${m}_sb${sb} $m --pixel-stride 8 --sync-below $sb
  done
done
run s1_m1.5 1.5
run s1_m2 2
run s1_m1.5_sb0 1.5 --sync-below 0
run s1_m1.5_sb2048 1.5 --sync-below 2048
