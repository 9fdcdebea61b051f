# (with staggered_split.patch applied: COBEL_DEBUG_PWG_SPLIT exists only there)
# (on the GPU box) split launches against whole instances on C3 and its shards
export COBEL_DEBUG=1
E=scripts/experiments/exp_pwg.py
for n in 65536 32768 16384 8192; do
  for sp in ${SPLITS:-0 auto}; do
    if [ "$sp" = "auto" ]; then
      PWG_N=$n timeout -k 10 120 python $E 60 2>&1 | grep -v amdgpu.ids | sed "s/^/split=auto /"
    else
      COBEL_DEBUG_PWG_SPLIT=$sp PWG_N=$n timeout -k 10 120 python $E 60 2>&1 | grep -v amdgpu.ids | sed "s/^/split=$sp /"
    fi
  done
done
