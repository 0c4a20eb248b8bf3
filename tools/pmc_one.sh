#!/bin/bash
# One rocprofv3 counter pass over a bench configuration, summed per kernel.
# usage: tools/pmc_one.sh <outdir> "<counters>" <bench args...>
set -u
OUT=$1; CTRS=$2; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/raw" -- python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob(os.path.join(d, "raw", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
with open(os.path.join(d, "counters.txt"), "a") as out:
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        out.write("%-28s n=%d %s\n" % (k, max(n[(k, c)] for c in v), " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items()))))
PY
rm -rf "$OUT/raw"
