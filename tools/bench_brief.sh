#!/bin/bash
# bench.py on C3 without the CPU leg; prints the few numbers that matter while tuning
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.4g  ms/step %.3f  pair_ms %.3f  hbm_frac %.3f  fp64_frac %.3f  rebuilds %s x %.2f ms  redos %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['fp64']['frac'], d['config']['list_rebuilds'], d['config']['rebuild_ms_each'], d['config']['step_redos']))"
