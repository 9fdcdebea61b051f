#!/bin/bash
# Same-box A/B of k_tab_pwg variants: the working tree's tabular_pwg.hip as libcobel_B.so, HEAD's as
# libcobel_A.so (both next to libcobel_hip.so; they travel to the GPU box with the snapshot).
#   bash scripts/ab_pwg_build.sh && gpurun -- 'bash scripts/ab_pwg_run.sh'
set -e
cd "$(dirname "$0")/../cobel-rl_amd/csrc"
mkdir -p ../lib/obj_x
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function"
O="../lib/obj/world.o ../lib/obj/tabular.o ../lib/obj/tabular_nact.o ../lib/obj/sr.o ../lib/obj/sr_wave.o ../lib/obj/general.o ../lib/obj/sfma.o ../lib/obj/adam.o ../lib/obj/mlp.o ../lib/obj/mlp_fit.o ../lib/obj/dsr_targets.o ../lib/obj/dqn_act.o"
git show HEAD:cobel-rl_amd/csrc/tabular_pwg.hip > ab_a.hip
cp tabular_pwg.hip ab_b.hip
for v in a b; do
  /opt/rocm/bin/hipcc $F -c ab_$v.hip -o ../lib/obj_x/ab_$v.o
  V=$(echo $v | tr a-z A-Z)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcobel_$V.so $O ../lib/obj_x/ab_$v.o
done
rm -f ab_a.hip ab_b.hip
ls -la ../lib/*.so
