#!/bin/bash
# Counter passes of the step-list draw (k_draw_lists) over the default bench run: one rocprofv3
# run per pass, counters only; the summary takes the last five dispatches (developed flow).
# usage: tools/pmc_draw.sh <outdir> [kernel-regex]
set -u
out=$1; shift
rx=${1:-k_draw_lists}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
run() {
  name=$1; ctrs=$2
  timeout -s KILL 280 rocprofv3 --pmc $ctrs --kernel-include-regex "$rx" --output-format csv -d "$out/$name" -- python bench.py --steps 5 --warmup 2 --no-cpu > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
  echo "pass $name ok"
}
run fetch "FETCH_SIZE" &&
run write "WRITE_SIZE" &&
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" &&
run sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" &&
run sq2 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU" &&
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" &&
run ta "TA_BUSY_avr TA_BUSY_max" &&
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
python - "$out" "$rx" <<'P' | tee "$out/summary.txt"
import csv, glob, sys
from collections import defaultdict
root, rx = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
    by = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in sorted(by.items()):
        v.sort()
        last = [x for _, x in v[-5:]]
        print("%-34s %16.6g   (last %d of %d dispatches)" % (c, sum(last) / len(last), len(last), len(v)))
P
