#!/bin/bash
# vp_sweep.sh: bench window + fits for a few settings of an environment knob.  usage: vp_sweep.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v python bench.py --no-cpu --joint-fit-cap 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernel_ms']
print('bench %.2f it/s %.3f ms/step fields %.3f ms %.2f passes' % (d['value'], d['ms_per_step'], k['fields'], k['field_passes_per_evaluation']))
for n,v in d['fit'].items(): print('  ', n, round(v.get('seconds_total', v.get('seconds_wall', v.get('seconds_optimize'))),3), v.get('iterations'))
"
  env $VAR=$v python scripts/config_table.py c2 c3 c4 c3_g 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    n,_,j=l.partition(' ')
    try: d=json.loads(j)
    except Exception: continue
    print('  ', n, d['fit_seconds'], d['iterations'], d['field_solver'])
"
done
