#!/bin/bash
# (on the GPU box) alternate the two builds on one bench configuration, twice
CFG=${1:-C2}
for k in 1 2; do
  for L in A B; do
    COBEL_LIB=$PWD/cobel-rl_amd/lib/libcobel_$L.so timeout -k 10 200 python bench.py --full --config $CFG --also "" --no-cpu-baseline --min-seconds 0 --steps 8 --full-out "" 2>/dev/null | python -c "
import json,sys
r=json.load(sys.stdin)
print('$L $CFG %.4g env-steps/s' % r['value'], [round(x,4) for x in r['roofline']['launch_ms_all']])"
  done
done
