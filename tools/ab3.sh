#!/bin/bash
# A/B of device-layer builds on the headline workload, per-class kernel times of the warm-up frame beside
# the frame time: tools/ab3.sh <name>... (names under pathtracer_amd/_variants, or "base"); AB_ARGS: extra bench arguments
set -e
mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" = base ]; then unset PTAMD_LIB; else export PTAMD_LIB=$PWD/pathtracer_amd/_variants/libptamd_$v.so; fi
  timeout -k 10 300 python3 bench.py $AB_ARGS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab/${v}.json 2> gpurun_out/ab/${v}.err
  python3 - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab/%s.json" % sys.argv[1])); r = d["roofline"]["kernel_ms_all_classes_warmup_frame"]
print("%-12s %7.2f ms  %.4g samples/s " % (sys.argv[1], d["ms_per_step"], d["value"]), {k: round(v, 1) for k, v in r.items()})
PY
done
