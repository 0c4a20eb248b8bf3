#!/bin/bash
# Per-launch timeline of the last frame of one bench configuration (rocprofv3 kernel trace).
# usage: tools/timeline_run.sh <outdir> <bench args...>   -> <outdir>/timeline.txt
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 bench.py "$@" > "$OUT/trace.json" 2> "$OUT/trace.err" || exit 1
python3 tools/trace_timeline.py "$OUT/trace" -2 > "$OUT/timeline.txt"  # the last timed frame
rm -rf "$OUT/trace"
head -1 "$OUT/timeline.txt"
