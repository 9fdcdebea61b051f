#!/bin/bash
# A/B of the network legs on a GPU box: the library in the tree against other builds, alternating.
#   bash scripts/ab_net.sh "C5_f32,C5_f64" 2 gpurun_ab/libcobel_net0.so [more .so ...]
cd "$GRAFT_REPO_ROOT"
export COBEL_DEBUG=1
LEGS=$1; REP=$2; shift 2
for r in $(seq 1 $REP); do
  for lib in HEAD "$@"; do
    if [ "$lib" = HEAD ]; then unset COBEL_LIB; else export COBEL_LIB=$PWD/$lib; fi
    python3 bench.py --full --no-cpu-baseline --min-seconds 0 --max-pretrain 2 --also= --legs $LEGS 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', ' '.join('%s %.4f ms (%.3f)' % (k, v.get('ms_per_step') or 0, (v.get('roofline') or {}).get('frac') or 0) for k, v in d['other_configs'].items()))"
  done
done
