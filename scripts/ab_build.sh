#!/bin/bash
# Same-box A/B of ONE kernel source: HEAD's version as libcobel_A.so, the working tree's as
# libcobel_B.so (next to libcobel_hip.so: they travel to the GPU box with the snapshot).
#   bash scripts/ab_build.sh tabular.hip && gpurun -- 'bash scripts/ab_run_bench.sh C2'
set -e
SRC=${1:-tabular_pwg.hip}
BASE=${SRC%.hip}
cd "$(dirname "$0")/../cobel-rl_amd/csrc"
mkdir -p ../lib/obj_x
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function"
O=""
for f in world tabular tabular_pwg tabular_nact sr sr_wave general sfma adam mlp mlp_fit dsr_targets dqn_act; do
  [ "$f" = "$BASE" ] || O="$O ../lib/obj/$f.o"
done
git show HEAD:cobel-rl_amd/csrc/$SRC > ab_a.hip
cp $SRC ab_b.hip
for v in a b; do
  /opt/rocm/bin/hipcc $F -c ab_$v.hip -o ../lib/obj_x/ab_$v.o
  V=$(echo $v | tr a-z A-Z)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcobel_$V.so $O ../lib/obj_x/ab_$v.o
done
rm -f ab_a.hip ab_b.hip
ls -la ../lib/*.so
