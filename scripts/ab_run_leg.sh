#!/bin/bash
# (on the GPU box) alternate the two builds on the extra legs of bench.py, twice:
#   bash scripts/ab_run_leg.sh grid_search,general_dynaq_b100
LEGS=${1:-grid_search}
for k in 1 2; do
  for L in A B; do
    COBEL_LIB=$PWD/cobel-rl_amd/lib/libcobel_$L.so timeout -k 10 300 python bench.py --full --also "" --legs $LEGS --no-cpu-baseline --min-seconds 0 --steps 4 --full-out "" 2>/dev/null | python -c "
import json,sys
r=json.load(sys.stdin)
for k,v in r.get('other_configs',{}).items():
    print('$L', k, '%.4g %s' % (v['value'], v['unit']), v.get('ms_per_step'))"
  done
done
