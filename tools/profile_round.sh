#!/bin/bash
# One workload's full rocprofv3 evidence, condensed on the box: runs tools/pmc_passes.sh,
# summarises, keeps the summary + the kernel-trace stats and drops the raw counter dumps
# (hundreds of MB). usage: tools/profile_round.sh <outdir> <bench args...>
set -e
OUT=$1; shift
tools/pmc_passes.sh "$OUT" "$@"
python3 tools/isa_mix.py --json "$OUT/isa_mix.json" > "$OUT/isa_mix.txt"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt"
python3 tools/trace_hist.py "$OUT/trace" > "$OUT/duration_hist.txt"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
for d in trace pmc_fetch pmc_write pmc_l2 pmc_sq pmc_grbm pmc_sq2 pmc_l1; do rm -rf "$OUT/$d"; done
rm -f "$OUT"/*.err
ls "$OUT"
