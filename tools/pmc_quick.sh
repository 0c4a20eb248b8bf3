#!/bin/bash
# One or more PMC passes on a short bench run; prints per-kernel sums for the traversal kernels.
# usage: tools/pmc_quick.sh <outdir> "<counters pass 1>" ["<counters pass 2>" ...] -- <bench args>
set -u
OUT=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python3 bench.py "$@" > "$OUT/p$i.json" 2> "$OUT/p$i.err" || { echo "pass $i failed"; tail -3 "$OUT/p$i.err"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    if k.startswith(("k_extend", "k_shadow", "k_shade", "k_resolve", "k_light", "k_trace2")):
        print(k, {c: "%.4g" % v for c, v in sorted(acc[k].items())})
PY
