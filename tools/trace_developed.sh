#!/bin/bash
# Development aid: kernel-trace statistics of 600 steps of the collapsing column (where the time of a
# list rebuild goes).  usage: tools/trace_developed.sh <tag>
set -u
tag=$1
out=gpurun_out/dev_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 tools/time_developed.py 0 600 > "$out/run.log" 2>&1 || { echo "run failed"; tail -5 "$out/run.log"; exit 1; }
f=$(find "$out/stats" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out/kernel_stats.csv"
head -16 "$f" | cut -c1-60,300-
