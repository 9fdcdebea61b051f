#!/bin/bash
# (on the GPU box) C3 at full size, trained agents: a list of "LIB:nl,ng" builds / wave mixes, twice
#   bash scripts/ab_libs.sh "A:- B:10,3 C:9,4"
export COBEL_DEBUG=1
E=scripts/experiments/exp_pwg.py
for k in 1 2; do
  for X in $1; do
    L=${X%%:*}; M=${X#*:}
    if [ "$M" = "-" ]; then
      COBEL_LIB=libcobel_$L.so timeout -k 10 120 python $E ${PRE:-60} 2>&1 | grep -v amdgpu.ids
    else
      COBEL_DEBUG_PWG=$M COBEL_LIB=libcobel_$L.so timeout -k 10 120 python $E ${PRE:-60} 2>&1 | grep -v amdgpu.ids
    fi
  done
done
