#!/bin/bash
# (on the GPU box) alternate the two builds, twice: C3 at full size, trained agents
export COBEL_DEBUG=1
E=scripts/experiments/exp_pwg.py
for k in 1 2; do
  for L in libcobel_A.so libcobel_B.so; do
    COBEL_LIB=$L timeout -k 10 120 python $E 60 2>&1 | grep -v amdgpu.ids
  done
done
