#!/bin/bash
# Profiles of one round: kernel-trace stats of the default bench, HBM byte counters of the pair
# kernel, and the counter calibration.  usage: tools/profile_round.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python bench.py --steps 10 --warmup 2 --no-cpu > "$out/stats.log" 2>&1 || { echo "stats run failed"; tail -5 "$out/stats.log"; exit 1; }
echo "stats ok"; tail -1 "$out/stats.log" > "$out/bench_under_trace.json"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --kernel-include-regex "k_pair" --output-format csv -d "$out/pair_$c" -- python bench.py --steps 2 --warmup 1 --no-cpu > "$out/pair_$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$out/pair_$c.log"; exit 1; }
  echo "pair $c ok"
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out/calib_$c" -- tools/bin/fetch_calib > "$out/calib_$c.log" 2>&1 || { echo "calib $c failed"; tail -5 "$out/calib_$c.log"; exit 1; }
  echo "calib $c ok"
done
python tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
