#!/bin/bash
# Development aid: kernel-trace statistics of DensityReinit at the benchmark's size.
set -u
out=gpurun_out/reinit_trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 tools/time_reinit.py > "$out/run.log" 2>&1 || { echo "run failed"; tail -5 "$out/run.log"; exit 1; }
f=$(find "$out/stats" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out/kernel_stats.csv"
grep reinit "$f" | cut -c1-40,200-
cat "$out/run.log" | grep "^n "
