#!/bin/bash
# Bench line (no CPU leg) for several settings of the automatic skin's constant and cap.
# usage (GPU box): tools/sweep_skin.sh "K2:MAX ..."      e.g. "16:0.5 24:0.5 36:0.6"
cd "$GRAFT_REPO_ROOT" || exit 1
for kv in $1; do
  k=${kv%%:*}; m=${kv##*:}
  SPH_SKIN_K2=$k SPH_SKIN_MAX=$m timeout -k 10 200 python bench.py --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('K2 $k max $m: %.3f ms/step  %.1f M  rebuilds %d x %.2f ms  draw %.2f ms  pair %.3f ms' % (d['ms_per_step'], d['value']/1e6, c['list_rebuilds'], c['rebuild_ms_each'], c['draw_ms_each'], d['roofline']['kernel_ms']))" || exit 1
done
