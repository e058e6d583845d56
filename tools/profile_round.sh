#!/bin/bash
# Profiles of one round: the default bench line, kernel-trace stats of the same command, and the
# HBM byte counters of the pair kernel over it (one rocprofv3 run per counter, counters only).
# usage: tools/profile_round.sh <tag>      (run on the GPU box; results under gpurun_out/<tag>)
set -u
tag=$1
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -k 10 400 python bench.py > "$out/bench_default.log" 2>&1 || { echo "bench failed"; tail -5 "$out/bench_default.log"; exit 1; }
tail -1 "$out/bench_default.log" > "$out/bench_default.json"; echo "bench ok"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python bench.py --no-cpu > "$out/stats.log" 2>&1 || { echo "stats run failed"; tail -5 "$out/stats.log"; exit 1; }
echo "stats ok"; tail -1 "$out/stats.log" > "$out/bench_under_trace.json"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-include-regex "k_pair" --output-format csv -d "$out/pair_$c" -- python bench.py --steps 5 --warmup 2 --no-cpu > "$out/pair_$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$out/pair_$c.log"; exit 1; }
  echo "pair $c ok"
done
# what the launch keeps busy: vector-instruction issue, waits, the address units (one run per pass)
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-include-regex "k_pair" --output-format csv -d "$out/pair_SQ" -- python bench.py --steps 5 --warmup 2 --no-cpu > "$out/pair_SQ.log" 2>&1 || { echo "pass SQ failed"; tail -5 "$out/pair_SQ.log"; exit 1; }
echo "pair SQ ok"
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE --kernel-include-regex "k_pair" --output-format csv -d "$out/pair_TA" -- python bench.py --steps 5 --warmup 2 --no-cpu > "$out/pair_TA.log" 2>&1 || { echo "pass TA failed"; tail -5 "$out/pair_TA.log"; exit 1; }
echo "pair TA ok"
python tools/pmc_phases.py "$out" > "$out/pair_counters.txt" 2>&1
python tools/traffic_json.py "$out" > "$out/pair_traffic.json" && cat "$out/pair_traffic.json"
cp $(ls "$out"/stats/*/*kernel_stats.csv | head -1) "$out/kernel_stats.csv"
