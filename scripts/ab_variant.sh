#!/bin/bash
# One more build of the library next to libcobel_hip.so for a same-box comparison: the working
# tree's (or, with REV=<commit>, that commit's) version of ONE kernel source compiled with extra
# flags, linked with the other objects of the regular build.
#   bash scripts/ab_variant.sh C tabular_pwg.hip -DCOBEL_PWG_GLB_TABLES     -> lib/libcobel_C.so
#   REV=HEAD bash scripts/ab_variant.sh A tabular_pwg.hip
set -e
NAME=$1; SRC=$2; shift 2
BASE=${SRC%.hip}
cd "$(dirname "$0")/../cobel-rl_amd/csrc"
mkdir -p ../lib/obj_x
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function"
O=""
for f in world tabular tabular_pwg tabular_nact sr sr_wave general sfma adam mlp mlp_fit dsr_targets dqn_act; do
  [ "$f" = "$BASE" ] || O="$O ../lib/obj/$f.o"
done
if [ -n "$REV" ]; then git show $REV:cobel-rl_amd/csrc/$SRC > ab_$NAME.hip; else cp $SRC ab_$NAME.hip; fi
/opt/rocm/bin/hipcc $F "$@" -c ab_$NAME.hip -o ../lib/obj_x/ab_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcobel_$NAME.so $O ../lib/obj_x/ab_$NAME.o
rm -f ab_$NAME.hip
ls -la ../lib/libcobel_$NAME.so
