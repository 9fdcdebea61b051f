#!/bin/bash
# (on the GPU box) HEAD's build (libcobel_A.so) against the working tree's (libcobel_B.so) on C3 at
# full size, trained agents, for a list of LDS / global-memory wave mixes of B:
#   bash scripts/ab_pwg_mixes.sh "10,3 10,4 9,4"
export COBEL_DEBUG=1
E=scripts/experiments/exp_pwg.py
for k in 1 2; do
  COBEL_LIB=libcobel_A.so timeout -k 10 120 python $E 60 2>&1 | grep -v amdgpu.ids
  for M in ${1:-"10,3"}; do
    COBEL_DEBUG_PWG=$M COBEL_LIB=libcobel_B.so timeout -k 10 120 python $E 60 2>&1 | grep -v amdgpu.ids
  done
done
