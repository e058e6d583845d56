#!/bin/bash
# Counter passes over a short bench run (one rocprofv3 run per pass; counters only).
# usage: [KERNEL=regex] [PASSES="sq1 ta"] tools/pmc.sh <outdir> [bench args...]     (KERNEL defaults to k_pair)
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
# PASSES="sq1 sq2 ta" limits the run to the named passes
run() {
  name=$1; ctrs=$2; shift 2
  if [ -n "${PASSES:-}" ] && ! [[ " $PASSES " == *" $name "* ]]; then return 0; fi
  timeout -s KILL 180 rocprofv3 --pmc $ctrs --kernel-include-regex "${KERNEL:-k_pair}" --output-format csv -d "$out/$name" -- python bench.py --steps 2 --warmup 1 --no-cpu "$@" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "$@" &&
run sq2 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS" "$@" &&
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "$@" &&
run fetch "FETCH_SIZE" "$@" &&
run write "WRITE_SIZE" "$@" &&
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "$@" &&
run ta "TA_BUSY_avr TA_BUSY_max" "$@" &&
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT" "$@"
