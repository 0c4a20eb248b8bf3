#!/bin/bash
# The numbers quoted in the documents, in one run on one box: every workload of the benchmark table,
# the irregular stand-in, the earlier point-light headline, and the rehearsal of a rank of 2 / 4 / 8.
set -e
mkdir -p gpurun_out/final
run() {
  local name=$1; shift
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/final/$name.json 2> gpurun_out/final/$name.err
  python3 - $name <<'PY'
import json, sys
d = json.load(open("gpurun_out/final/%s.json" % sys.argv[1])); r = d["roofline"]
print("%-14s %7.2f ms  %.4g samples/s  frac %s stale %s  %s" % (sys.argv[1], d["ms_per_step"], d["value"], r.get("frac") and round(r["frac"], 3),
      r.get("profile_stale"), {k: round(v, 1) for k, v in r["kernel_ms_all_classes_warmup_frame"].items()}))
PY
}
run dragon
run dragon_s2 --pixel-stride 2
run dragon_s4 --pixel-stride 4
run dragon_s8 --pixel-stride 8
run bunny --scene bunny
run bunny_s8 --scene bunny --pixel-stride 8
run dragon_point --scene dragon-point
run folded --scene folded
run lucy --scene lucy --width 1920 --height 1080 --depth 8
run spheres --scene spheres --depth 16
run spheres_spec --scene spheres --depth 16 --specular
run cbsl --scene cbsl --width 480 --height 360 --light-samples 16
