# (on the GPU box) slice plans of the shards of C3 (ten LDS waves per CU)
export COBEL_DEBUG=1
E=scripts/experiments/exp_pwg.py
run() { n=$1; sl=$2
  if [ "$sl" = "auto" ]; then PWG_N=$n timeout -k 10 120 python $E 60 2>&1 | grep -v amdgpu.ids | sed "s/^/slices=auto /"
  else COBEL_DEBUG_PWG_SLICES=$sl PWG_N=$n timeout -k 10 120 python $E 60 2>&1 | grep -v amdgpu.ids | sed "s/^/slices=$sl /"; fi; }
for sl in auto 384,128 352,96,64 256,128,64,64 288,128,64,32 320,128,48,16; do run 8192 $sl; done
for sl in auto 448,64 384,96,32 320,128,64; do run 16384 $sl; done
for sl in auto 512 480,32 384,128 416,64,32; do run 32768 $sl; done
for sl in auto 480,32 448,64; do run 65536 $sl; done
