source scripts/ab_pwg.sh
ARGS=""
run head A=1
run prev COBEL_LIB=$PWD/gpurun_ab/libcobel_pre_direct.so
run k0 COBEL_LIB=$PWD/gpurun_ab/libcobel_k0.so
ARGS="--instances 8192"
run head_8k A=1
run prev_8k COBEL_LIB=$PWD/gpurun_ab/libcobel_pre_direct.so
